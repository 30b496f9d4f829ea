#!/usr/bin/env python3
"""Where do the largest element-wise differences of the bench workload sit?  (GPU box)  python tools/parity_probe.py [cuts]
For 64 cuts of the bench's synthetic input: HIP vs float32 oracle vs float64 oracle; prints the worst elements (cut, frame, mel), the
three values, the mel energy they correspond to, and per-mel-filter maxima."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd as LA
from oracle.kaldi_ref import RefConfig, RefExtractor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(1234)
wave = torch.empty((n, 160000), device="cuda").uniform_(-0.5, 0.5, generator=g)
ex = LA.HipFbank()
out = ex.extract_batch(list(wave), 16000)
out = out.cpu().numpy() if isinstance(out, torch.Tensor) else np.stack([o.cpu().numpy() for o in out])
o32, o64 = RefExtractor(RefConfig(kind="fbank"), np.float32), RefExtractor(RefConfig(kind="fbank"), np.float64)
W = wave.cpu().numpy()
want = np.stack([o32.extract(w) for w in W]); truth = np.stack([o64.extract(w) for w in W])
d_hw, d_ht, d_wt = np.abs(out - want), np.abs(out - truth), np.abs(want - truth)
print(f"max |hip-ref32| {d_hw.max():.3e}  max |hip-f64| {d_ht.max():.3e}  max |ref32-f64| {d_wt.max():.3e}")
print(f"rms |hip-f64| {np.sqrt((d_ht**2).mean()):.3e}  rms |ref32-f64| {np.sqrt((d_wt**2).mean()):.3e}")
for name, d in (("hip-f64", d_ht), ("ref32-f64", d_wt)):
    print(name, "per-mel max (first 8, then every 8th):", np.round(d.max(axis=(0, 1))[:8], 5), np.round(d.max(axis=(0, 1))[8::8], 5))
    print(name, "quantiles 0.999 / 0.99999:", np.quantile(d, 0.999), np.quantile(d, 0.99999))
idx = np.argsort(d_ht.ravel())[::-1][:12]
for i in idx:
    c, t, m = np.unravel_index(i, d_ht.shape)
    print(f"cut {c} frame {t} mel {m}: hip {out[c,t,m]:.5f} ref32 {want[c,t,m]:.5f} f64 {truth[c,t,m]:.5f}  (mel energy {np.exp(truth[c,t,m]):.3e}; row median energy {np.exp(np.median(truth[c,t])):.3e})")
