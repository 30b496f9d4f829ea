#!/usr/bin/env python3
"""Host work per on-the-fly mini-batch (GPU box): wall time of enqueueing bursts of mini-batches on a drained device -- Python + ctypes +
the HIP launch calls of hipfeat_minibatch_plan / _run -- and a cProfile of the same loop.  python tools/host_profile_minibatch.py [prefetch]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = argparse.Namespace(cuts=64, prefetch=K, streams=3, route="pair", no_host_fed=True)
w = bench.OnTheFly(torch.device("cuda", 0), 0, args)
for route, ns in (("pair", 3), ("pair", 1), ("per_factor", 1)):
    if route == "per_factor" and K > 1:
        continue
    w.route, w.nstreams = route, ns
    w.streams = [torch.cuda.Stream() for _ in range(ns)] if ns > 1 else []
    r, h = w._rate(0.5)
    print(f"route={route:10s} streams={ns} prefetch={K}: {r:10.0f} cuts/s device-resident, host {h:6.2f} us per mini-batch")
w.route, w.nstreams, w.streams = "pair", 1, []
keep = w.batches
w.batches = keep[: max(1, 8 // K)]
pr = cProfile.Profile()
for _ in range(200):
    torch.cuda.synchronize()
    pr.enable(); w.step(); pr.disable()
n = 200 * len(w.batches) * K
print(f"\ncProfile of {n} mini-batches enqueued in bursts of {len(w.batches) * K} on a drained device (1 stream; profiler overhead included):")
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)
