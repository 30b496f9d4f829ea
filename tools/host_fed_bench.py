#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in API (never reported as bench `value`): HipFbank.extract_batch on
host tensors, i.e. pack -> H2D -> kernel -> D2H, for the default 600 s batch (60 x 10 s) and a large one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd
ex = lhotse_amd.HipFbank()
for B in (60, 1024):
    x = (torch.rand(B, 160000) - 0.5)
    xs = [x[i] for i in range(B)]
    xn = [x[i].numpy() for i in range(B)]
    for name, arg, kw in [("padded tensor + lengths (collate=True)", x, dict(lengths=torch.full((B,), 160000, dtype=torch.int32))),
                          ("list of torch tensors", xs, {}), ("list of numpy arrays", xn, {})]:
        ex.extract_batch(arg, 16000, **kw); torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            r = ex.extract_batch(arg, 16000, **kw)
            if isinstance(r, torch.Tensor): r = r.cpu()
            elif isinstance(r, list) and isinstance(r[0], torch.Tensor): r = [t.cpu() for t in r]
            n += B
        torch.cuda.synchronize()
        print(f"B={B:5d} {name:42s} {n / (time.perf_counter() - t0):10.0f} cuts/s (host in, host out)")

# fused collation (features stay on the GPU, padded (B, Tmax, F)) and int16 PCM input
for B in (60, 1024):
    lens = np.random.RandomState(0).randint(8 * 16000, 12 * 16000, size=B)
    xf = [(np.random.RandomState(i).rand(n).astype(np.float32) - 0.5) for i, n in enumerate(lens)]
    xi = [np.round(a * 32767).astype(np.int16) for a in xf]
    for name, arg in [("float32 list -> collated on device", xf), ("int16 PCM list -> collated on device", xi)]:
        ex.extract_collated(arg, 16000); torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            f, l = ex.extract_collated(arg, 16000)
            n += B
        torch.cuda.synchronize()
        print(f"B={B:5d} {name:42s} {n / (time.perf_counter() - t0):10.0f} cuts/s (host in, device out)")
