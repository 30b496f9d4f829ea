#!/usr/bin/env python3
"""Localise a wave_kernel discrepancy: wave vs generic kernel on variants of a 22.05 kHz fbank config."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch, warnings
    import lhotse_amd as LA
    rng = np.random.RandomState(0)
    x = (rng.rand(22050 * 2).astype(np.float32) - 0.5)
    out = {}
    variants = {
        "default": {},
        "nopre": {"preemph_coeff": 0.0},
        "nodc": {"remove_dc_offset": False},
        "nopre_nodc": {"preemph_coeff": 0.0, "remove_dc_offset": False},
        "rect_nopre_nodc": {"preemph_coeff": 0.0, "remove_dc_offset": False, "window_type": "rectangular"},
    }
    for name, kw in variants.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ex = LA.HipSpectrogram(LA.HipSpectrogramConfig(sampling_rate=22050, **kw))
        y = ex.extract(x, 22050)
        np.save(f"gpurun_out/dbg_{sys.argv[1]}_{name}.npy", y)
        print(sys.argv[1], name, ex.kernel_name, y.shape)
else:
    for mode, env in (("wave", {}), ("generic", {"HIPFEAT_NO_WAVE_KERNEL": "1"})):
        subprocess.run([sys.executable, __file__, mode], env={**os.environ, **env}, check=True)
    for name in ("default", "nopre", "nodc", "nopre_nodc", "rect_nopre_nodc"):
        a, b = np.load(f"gpurun_out/dbg_wave_{name}.npy"), np.load(f"gpurun_out/dbg_generic_{name}.npy")
        d = np.abs(a - b) / (np.abs(b).max(axis=1, keepdims=True) + 1e-30)
        bad = np.argwhere(d > 1e-4)
        print(name, "max rel-to-row-peak err", d.max(), "bad count", len(bad), "first bad", bad[:6].tolist(), "rows bad", np.unique(bad[:, 0])[:8].tolist(), "cols bad", np.unique(bad[:, 1])[:12].tolist())
