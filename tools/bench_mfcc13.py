#!/usr/bin/env python3
"""Kaldi-default MFCC (23 filters, 13 cepstra) on 10 000 x 10 s cuts at 16 and 8 kHz, device resident."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
for sr in (16000, 8000):
    ex = LA.HipMfcc(LA.HipMfccConfig(sampling_rate=sr))
    plan = ex.plan
    C, S = 10000, 10 * sr
    wave = torch.empty(C * S, device="cuda").uniform_(-0.5, 0.5)
    offs = np.arange(C, dtype=np.int64) * S
    lens = np.full(C, S, dtype=np.int64)
    plan.run(wave, offs, lens, None); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, e in evs:
        s.record(); plan.run(wave, offs, lens, None); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
    print(json.dumps({"sampling_rate": sr, "kernel": plan.kernel_name, "ms_per_launch": round(ms, 3), "cuts_per_s": round(C / ms * 1e3, 1)}))
    del wave
