#!/usr/bin/env python3
"""Per-phase shader clocks per wave-round of fft1024c_kernel (experiment build -DHIPFEAT_PHASE_TIMERS), 24 kHz fbank-80.
usage (GPU box): HIPFEAT_LIB=lhotse_amd/_lib/var_<name>.so python tools/phase_timers_w.py [cuts] [sampling_rate]"""
import ctypes, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd as LA
from lhotse_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sr = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ex = LA.HipFbank(LA.HipFbankConfig(sampling_rate=sr))
plan = ex.plan; L = plan.lib
dll = L.backend.dll
dll.hipfeat_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
S = 10 * sr
wave = torch.empty(C * S, device="cuda").uniform_(-0.5, 0.5)
offs = np.arange(C, dtype=np.int64) * S; lens = np.full(C, S, dtype=np.int64)
buf = torch.zeros(C * 8 * 16 * 8, dtype=torch.int64, device="cuda")
assert dll.hipfeat_debug_set_phase_buffer(buf.data_ptr()) == 0
for it in range(2):
    plan.run(wave, offs, lens, None); torch.cuda.synchronize()
v = buf.view(-1, 8).double().sum(0).cpu().numpy()
rounds = v[7]
names = ["sample / neighbour / window reads", "span request (LDS-DMA issue)", "mean, prolog, pass 1, twiddles", "exchange", "pass 2, split, power rows", "wait for the next span", "mel phase (reads, MFMA, log, stores)"]
print(plan.kernel_name, "wave-rounds:", int(rounds))
tot = v[:7].sum()
for n, x in zip(names, v[:7]):
    print(f"  {n:38s} {x / rounds:9.0f} clk per wave-round  ({100 * x / tot:5.1f} %)")
print(f"  total {tot / rounds:9.0f} clk per wave-round")
