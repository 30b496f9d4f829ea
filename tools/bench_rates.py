#!/usr/bin/env python3
"""Throughput of 80-dim fbank at several sampling rates (25/10 ms frames), device resident, 10 s cuts.  JSON lines."""
import argparse, json, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=2000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=8, help="untimed launches first: the shader clock ramps up over the first ~6 launches after an idle phase")
ap.add_argument("--rates", default="8000,16000,22050,24000,32000,44100,48000")
a = ap.parse_args()
for sr in [int(r) for r in a.rates.split(",")]:
    S = 10 * sr
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = LA.HipFbank(LA.HipFbankConfig(sampling_rate=sr))
    plan = ex.plan
    wave = torch.empty(a.cuts * S, device="cuda").uniform_(-0.5, 0.5)
    offs = np.arange(a.cuts, dtype=np.int64) * S
    lens = np.full(a.cuts, S, dtype=np.int64)
    for _ in range(max(1, a.warmup)):
        plan.run(wave, offs, lens, None)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in evs:
        s.record(); out, fr = plan.run(wave, offs, lens, None); e.record()
    torch.cuda.synchronize()
    ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
    bpc = S * 4 + int(fr[0]) * 80 * 4
    print(json.dumps({"sampling_rate": sr, "fft": plan.fft, "kernel": plan.kernel_name.split(" ")[0] + (" fixed-schedule" if "fixed-schedule" in plan.kernel_name else ""), "ms_per_launch": round(ms, 3),
                      "cuts_per_s": round(a.cuts / ms * 1e3, 1), "audio_seconds_per_s": round(a.cuts * 10 / ms * 1e3, 1),
                      "frac_of_8TBps": round(a.cuts * bpc / ms / 1e6 / 8000, 3)}))
    del wave, out
