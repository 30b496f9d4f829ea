#!/usr/bin/env python3
"""Speed perturbation + fbank, device resident (BASELINE configs[4]-shaped: LibriSpeech-like cut lengths,
3x speed perturbation 0.9 / 1.0 / 1.1 -> 80-dim log-mel).  Prints one JSON line with the throughput of
(a) the resample kernel alone and (b) resample + feature extraction, in perturbed cuts/s and audio-seconds/s.

    python tools/bench_speed_fbank.py [--cuts 6000] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA  # noqa: E402
from lhotse_amd import _lib, augmentation as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cuts", type=int, default=6000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=6, help="untimed passes per factor (at least 6)")
    ap.add_argument("--unaligned", action="store_true", help="resampled cuts back to back (not 16-byte aligned)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(0)
    # LibriSpeech train-clean-100 utterances: 1.4 .. 24.5 s, mean ~12.7 s
    lens = np.clip(rng.normal(12.7, 3.6, size=args.cuts), 1.4, 24.5)
    lens = (lens * 16000).astype(np.int64)
    offs = np.zeros(len(lens), dtype=np.int64)
    np.cumsum(lens[:-1], out=offs[1:])
    total = int(lens.sum())
    wave = torch.empty(total, device=dev).uniform_(-0.5, 0.5)
    ex = LA.HipFbank()
    plan = ex.plan
    res = {}
    for factor in (0.9, 1.1):
        r = A.get_or_create_resampler(round(16000 * factor), 16000)
        # >= 6 untimed passes per factor: the first launches after an idle phase (plan creation, the previous factor's host work) run at a
        # ramping shader clock and through cold caches -- with ONE warm-up the first factor used to be timed cold (5.97 ms
        # recorded next to 1.9-2.0 ms in every other run).  tools/bench_rates.py documents the same ramp.
        for _ in range(max(6, args.warmup)):
            out, ooffs, olens = r.run(wave, offs, lens, align=not args.unaligned)
            feats, frames = plan.run(out, ooffs, olens, None)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for a, b, c in evs:
            a.record()
            out, ooffs, olens = r.run(wave, offs, lens, align=not args.unaligned)
            b.record()
            feats, frames = plan.run(out, ooffs, olens, None)
            c.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps
        rs_ms = float(np.mean([a.elapsed_time(b) for a, b, c in evs]))
        fb_ms = float(np.mean([b.elapsed_time(c) for a, b, c in evs]))
        rs_bytes = 4 * (total + int(olens.sum()))
        res[f"speed{factor}"] = {
            "resample_ms": round(rs_ms, 3),
            "resample_GBps": round(rs_bytes / rs_ms / 1e6, 1),
            "resample_frac_of_8TBps": round(rs_bytes / rs_ms / 1e6 / 8000, 3),
            "fbank_ms": round(fb_ms, 3),
            "wall_ms_per_step": round(wall * 1e3, 3),
            "cuts_per_s": round(args.cuts / wall, 1),
            "audio_seconds_per_s": round(total / 16000 / wall, 1),
        }
    print(json.dumps({"workload": f"{args.cuts} LibriSpeech-like cuts ({total / 16000 / 3600:.1f} h), speed perturb -> 80-dim fbank, device resident", **res}))


if __name__ == "__main__":
    main()
