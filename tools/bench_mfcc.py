#!/usr/bin/env python3
"""Config-4-shaped throughput: 40-dim MFCC (40 filters, 40 ceps, lifter 22) on LibriSpeech-like lengths
(log-normal, 1-35 s, seeded), one GPU, device-resident; cuts/s and audio-seconds/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd
from lhotse_amd import _lib
rs = np.random.RandomState(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dur = np.clip(np.exp(rs.randn(B) * 0.45 + 2.42), 1.0, 35.0)
lens = (np.round(dur * 16000).astype(np.int64) + 3) & ~3
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
ex = lhotse_amd.HipMfcc(lhotse_amd.HipMfccConfig(num_filters=40, num_ceps=40))
plan = ex.plan; L = plan.lib
wave = torch.rand(int(lens.sum()), device="cuda") - 0.5
frames = (lens + 80) // 160
out = torch.empty(int(frames.sum()), 40, device="cuda")
h = np.zeros(1, dtype=np.uint64)
L.check("hipfeat_layout_create", plan.handle, B, _lib.addr(offs), _lib.addr(lens), None, None, 40, None, _lib.addr(h))
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), out.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 10
for _ in range(K): L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), out.data_ptr(), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"{plan.kernel_name}: {B} cuts, mean {dur.mean():.1f} s: {B/dt:,.0f} cuts/s, {lens.sum()/16000/dt:,.0f} audio-s/s, "
      f"{(lens.sum()*4 + frames.sum()*160)/dt/1e9:.0f} GB/s algorithmic")
