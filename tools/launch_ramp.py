#!/usr/bin/env python3
"""Per-launch time of the headline kernel over the first launches after an idle phase (GPU box): how long does the device take to reach
its steady rate?  python tools/launch_ramp.py [launches]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
args = argparse.Namespace(cuts=10000, total_cuts=0, input="uniform", no_host_fed=True)
w = bench.Fbank16k(torch.device("cuda", 0), 0, args)
torch.cuda.synchronize()
for idle in (0.0, 2.0):
    time.sleep(idle)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); w.step(); b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(f"after {idle:.0f} s idle: launches 1-5 {t[:5].mean():.3f} ms, 6-25 {t[5:25].mean():.3f}, 26-50 {t[25:50].mean():.3f}, 51-100 {t[50:100].mean():.3f}, last 20 {t[-20:].mean():.3f}")
    print("   every 10th:", " ".join(f"{x:.3f}" for x in t[::10]))
