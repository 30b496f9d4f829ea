#!/usr/bin/env python3
"""SpecAugment on a device-resident (B, T, F) batch: HipSpecAugment (host draws + two launches) vs the reference's
per-sequence torch implementation restated in oracle/specaug_torch.py running on the same GPU.  One JSON line."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
from oracle.specaug_torch import TorchSpecAugment

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=60)
ap.add_argument("--frames", type=int, default=1500)
ap.add_argument("--dim", type=int, default=80)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
x = torch.randn(a.batch, a.frames, a.dim, device="cuda") * 3 - 8


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


hip = LA.HipSpecAugment()
ref = TorchSpecAugment()
ms_hip = timed(lambda: hip(x))
ms_ref = timed(lambda: ref(x))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
seg_rounds, masks = hip.draw(a.batch, a.frames, a.dim, None)
from lhotse_amd.signal_transforms import apply_specaug
apply_specaug(x, seg_rounds, masks); torch.cuda.synchronize()
ev0.record()
for _ in range(a.steps):
    apply_specaug(x, seg_rounds, masks)
ev1.record(); torch.cuda.synchronize()
dev_ms = ev0.elapsed_time(ev1) / a.steps
nbytes = x.numel() * 4 * 2
print(json.dumps({"workload": f"SpecAugment defaults on ({a.batch}, {a.frames}, {a.dim}) float32, device resident", "hip_ms_per_batch": round(ms_hip, 3),
                  "hip_device_ms": round(dev_ms, 4), "hip_device_GBps": round(nbytes / dev_ms / 1e6, 1), "torch_gpu_ms_per_batch": round(ms_ref, 3),
                  "speedup": round(ms_ref / ms_hip, 1)}))
