#!/usr/bin/env python3
"""Impulse probe of the wave kernel's pre-emphasis: for every position p of an impulse inside a frame, where does the -c tap land?"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
sr = 22050
x = np.zeros(sr * 60, dtype=np.float32)
pos = np.arange(2000, len(x) - 2000, 1409)
x[pos] = 1.0
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ex = LA.HipSpectrogram(LA.HipSpectrogramConfig(sampling_rate=sr, remove_dc_offset=False, window_type="rectangular", snip_edges=True))
print(ex.kernel_name)
y = ex.extract(x, sr)  # (T, 513) power
n, shift = 551, 220
res = {}
for t in range(y.shape[0]):
    lo = t * shift
    inside = pos[(pos >= lo) & (pos < lo + n)]
    if len(inside) != 1:
        continue
    p = int(inside[0] - lo)
    full = np.concatenate([y[t], y[t][-2:0:-1]])
    r = np.fft.ifft(full).real  # autocorrelation: r[0] = 1 + a^2, r[k] = a at the lag where the tap landed
    k = int(np.argmax(np.abs(r[1:200]))) + 1
    res.setdefault(p, (k, round(float(r[k]), 3), round(float(r[0]), 3)))
bad = {p: v for p, v in res.items() if not (v[0] == 1 and abs(v[1] + 0.97) < 1e-3) and p != n - 1}
print("positions probed", len(res), "bad", len(bad))
print(sorted(bad.items())[:40])
