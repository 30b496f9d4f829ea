#!/usr/bin/env python3
"""8 kHz (telephone) throughput, device resident: C x 10 s cuts @ 8 kHz -> (1000, M) fbank / mfcc.  JSON lines."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=10000)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
S = 80000
wave = torch.empty(a.cuts * S, device="cuda").uniform_(-0.5, 0.5)
offs = np.arange(a.cuts, dtype=np.int64) * S
lens = np.full(a.cuts, S, dtype=np.int64)
for name, ex, F in [("fbank-40", LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, num_filters=40)), 40),
                    ("fbank-80", LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, num_filters=80)), 80),
                    ("mfcc-13", LA.HipMfcc(LA.HipMfccConfig(sampling_rate=8000)), 13)]:
    plan = ex.plan
    plan.run(wave, offs, lens, None); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in evs:
        s.record(); out, fr = plan.run(wave, offs, lens, None); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
    bytes_per_cut = S * 4 + 1000 * F * 4
    print(json.dumps({"workload": f"{a.cuts} x 10 s @ 8 kHz -> {name}", "kernel": plan.kernel_name, "ms_per_launch": round(ms, 3),
                      "cuts_per_s": round(a.cuts / ms * 1e3, 1), "algorithmic_GBps": round(a.cuts * bytes_per_cut / ms / 1e6, 1),
                      "frac_of_8TBps": round(a.cuts * bytes_per_cut / ms / 1e6 / 8000, 3)}))
