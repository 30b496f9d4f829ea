#!/usr/bin/env python3
"""Throughput of every registered extractor with its DEFAULT config, device resident, 10 s cuts.  JSON lines."""
import argparse, json, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=2000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--only", default="", help="comma-separated extractor names (default: all)")
a = ap.parse_args()
for cls in (LA.HipFbank, LA.HipMfcc, LA.HipSpectrogram, LA.HipLogSpectrogram, LA.HipKaldifeatFbank, LA.HipKaldifeatMfcc, LA.HipWhisperFbank, LA.HipLibrosaFbank):
    if a.only and cls.name not in a.only.split(","):
        continue
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = cls()
    sr = getattr(ex.config, "sampling_rate", None) or getattr(getattr(ex.config, "frame_opts", None), "sampling_rate", None) or 16000
    sr = int(sr)
    plan = ex.plan if hasattr(ex, "plan") else ex.inner.plan
    S = 10 * sr
    wave = torch.empty(a.cuts * S, device="cuda").uniform_(-0.5, 0.5)
    offs = np.arange(a.cuts, dtype=np.int64) * S
    lens = np.full(a.cuts, S, dtype=np.int64)
    for _ in range(8):  # two output blocks alternate in the caching allocator: warm both; the shader clock ramps up over the first launches
        out, fr = plan.run(wave, offs, lens, None)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in evs:
        s.record(); out, fr = plan.run(wave, offs, lens, None); e.record()
    torch.cuda.synchronize()
    ms = float(np.median([s.elapsed_time(e) for s, e in evs]))
    bpc = S * 4 + int(fr[0]) * plan.feature_dim * 4
    print(json.dumps({"extractor": ex.name, "sampling_rate": sr, "feature_dim": plan.feature_dim, "kernel": plan.kernel_name.split(" ")[0], "ms_per_launch": round(ms, 3),
                      "cuts_per_s": round(a.cuts / ms * 1e3, 1), "frac_of_8TBps": round(a.cuts * bpc / ms / 1e6 / 8000, 3)}))
    del wave, out
