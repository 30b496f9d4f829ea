#!/usr/bin/env python3
"""VERDICT r5 task 7: non-default geometries at 32-48 kHz run the GENERIC wave-autonomous instances (fft2048c<32,...>, fft1024c<26/32,...>);
same-call A/B against the route they would take otherwise (HIPFEAT_NO_WAVE_AUTONOMOUS=1: wave_kernel), bit-compared.  JSON lines.
    python tools/bench_geometry_ab.py [--cuts 2000]"""
import argparse, json, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=2000)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
CASES = [dict(sampling_rate=48000, frame_length=0.02, frame_shift=0.01, num_filters=64),
         dict(sampling_rate=48000, frame_length=0.032, frame_shift=0.01, num_filters=80),
         dict(sampling_rate=44100, frame_length=0.04, frame_shift=0.01, num_filters=80),
         dict(sampling_rate=32000, frame_length=0.032, frame_shift=0.016, num_filters=64),
         dict(sampling_rate=24000, frame_length=0.04, frame_shift=0.01, num_filters=80)]
for cfg in CASES:
    sr = cfg["sampling_rate"]
    S = 10 * sr
    wave = torch.empty(a.cuts * S, device="cuda").uniform_(-0.5, 0.5)
    offs = np.arange(a.cuts, dtype=np.int64) * S
    lens = np.full(a.cuts, S, dtype=np.int64)
    res, outs = {}, {}
    for route in ("wave-autonomous", "wave_kernel"):
        if route == "wave_kernel":
            os.environ["HIPFEAT_NO_WAVE_AUTONOMOUS"] = "1"
        else:
            os.environ.pop("HIPFEAT_NO_WAVE_AUTONOMOUS", None)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ex = LA.HipFbank(LA.HipFbankConfig(**cfg))
        plan = ex.plan
        for _ in range(8):
            plan.run(wave, offs, lens, None)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        for s, e in evs:
            s.record(); out, fr = plan.run(wave, offs, lens, None); e.record()
        torch.cuda.synchronize()
        ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
        res[route] = {"kernel": plan.kernel_name.split(" ")[0], "ms_per_launch": round(ms, 3), "cuts_per_s": round(a.cuts / ms * 1e3, 1)}
        outs[route] = out[: int(fr[0]) * 4].clone()
        ex._drop_plan()
    os.environ.pop("HIPFEAT_NO_WAVE_AUTONOMOUS", None)
    d = (outs["wave-autonomous"] - outs["wave_kernel"]).abs().max().item()
    print(json.dumps({"config": cfg, **res, "wave_autonomous_over_wave_kernel": round(res["wave-autonomous"]["cuts_per_s"] / res["wave_kernel"]["cuts_per_s"], 3),
                      "max_abs_difference_between_the_routes": d}), flush=True)
    del wave
