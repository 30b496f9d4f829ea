#!/usr/bin/env python3
"""Whisper log-mel throughput, device resident: C x 10 s cuts @ 16 kHz -> (1000, 80).  One JSON line."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=2000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--quiet-tail", type=float, default=0.0, help="fraction of every cut (at its end) scaled by 1e-6: more than 80 dB under the rest, so the clamp of the normalisation binds there")
a = ap.parse_args()
ex = LA.HipWhisperFbank()
plan = ex.plan
S = 160000
wave = torch.empty(a.cuts * S, device="cuda").uniform_(-0.5, 0.5)
offs = np.arange(a.cuts, dtype=np.int64) * S
if a.quiet_tail >= 2:  # "comb": a quiet 0.1 s in every second -> every row block of every cut needs the clamp sweep (worst case)
    wave.view(a.cuts, 10, 16000)[:, :, :1600] *= 1e-6
elif a.quiet_tail > 0:
    wave.view(a.cuts, S)[:, int(S * (1 - a.quiet_tail)):] *= 1e-6
lens = np.full(a.cuts, S, dtype=np.int64)
plan.run(wave, offs, lens, None); torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
for s, e in evs:
    s.record(); out, fr = plan.run(wave, offs, lens, None); e.record()
torch.cuda.synchronize()
ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
print(json.dumps({"workload": f"{a.cuts} x 10 s cuts -> Whisper log-mel (1000, 80), device resident, quiet tail {a.quiet_tail}", "kernel": plan.kernel_name,
                  "ms_per_launch": round(ms, 3), "cuts_per_s": round(a.cuts / ms * 1e3, 1), "audio_seconds_per_s": round(a.cuts * 10 / ms * 1e3, 1),
                  "algorithmic_GBps": round(a.cuts * 960000 / ms / 1e6, 1)}))
