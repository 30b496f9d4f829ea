#!/usr/bin/env python3
"""What makes the first launches after the bench's barrier slow?  30 launches, then (a) a bare synchronize, (b) synchronize + zeroing the
3.2 GB output, (c) synchronize + zeroing 64 cuts' rows, (d) 20 ms sleep -- then 30 more launches, per-launch times.  (GPU box)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

args = argparse.Namespace(cuts=10000, total_cuts=0, input="uniform", no_host_fed=True)
w = bench.Fbank16k(torch.device("cuda", 0), 0, args)
def burst(n=30):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); w.step(); b.record()
    return evs
def report(tag, evs):
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(f"{tag:44s} launches 1-5 {t[:5].mean():.3f} ms   6-20 {t[5:20].mean():.3f}   21-30 {t[20:].mean():.3f}   first three: {t[0]:.3f} {t[1]:.3f} {t[2]:.3f}")
burst(60); torch.cuda.synchronize()
for tag, gap in (("bare synchronize", lambda: None), ("synchronize + zero the whole output (3.2 GB)", lambda: w.out.zero_()),
                 ("synchronize + zero 64 cuts' rows (20 MB)", lambda: w.out[: 64 * 1000].zero_()), ("synchronize + 20 ms sleep", lambda: time.sleep(0.02)),
                 ("synchronize + 200 ms sleep", lambda: time.sleep(0.2))):
    burst(40); torch.cuda.synchronize(); gap(); torch.cuda.synchronize()
    report(tag, burst(30))
