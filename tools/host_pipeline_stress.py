#!/usr/bin/env python3
"""GPU: the steps of tests/test_gpu_bulk_save.py::test_native_host_pipeline_equals_the_python_pipeline_bit_for_bit with progress lines and a
watchdog that dumps every thread's Python stack if a step stalls (a hang on the box otherwise costs the whole time limit).

    timeout 300 python tools/host_pipeline_stress.py"""
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
faulthandler.dump_traceback_later(90, exit=True)

import numpy as np
import torch

import lhotse_amd as LA
from lhotse_amd import storage as S


def say(*a):
    print(f"[{time.perf_counter():9.3f}]", *a, flush=True)


def batches(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        lens = rs.randint(2000, 90000, size=rs.randint(1, 40))
        out.append([((rs.rand(int(k)) - 0.5) * rs.choice([1.0, 0.1, 1e-3])).astype(np.float32) for k in lens])
    return out


for kind, pcm, half, zero_pad in [("fbank", False, False, False), ("fbank", True, True, False), ("mfcc", False, True, False), ("fbank", False, False, True)]:
    cfg = {"edge_rule": "batch_zero_pad"} if zero_pad else {}
    ex = LA.HipFbank(LA.HipFbankConfig(**cfg)) if kind == "fbank" else LA.HipMfcc(LA.HipMfccConfig(**cfg))
    bs = batches(11, 9) + [[(np.random.RandomState(1).rand(160000).astype(np.float32) - 0.5) for _ in range(60)]]
    if pcm:
        bs = [[(w * 32767).astype(np.int16) for w in waves] for waves in bs]
    say("case", kind, pcm, half, zero_pad, "plan", ex.kernel_name.split(" ")[0])
    pend = []
    for k, waves in enumerate(bs):
        p, frames = S._batch_features_pending(ex, [torch.from_numpy(w) for w in waves], 16000, None, half=half)
        say("  submitted", k, len(waves), "ticket", p.ticket)
        pend.append((p, frames, waves))
    for p, frames, waves in reversed(pend):
        t = p.ticket
        got = p.wait().copy()
        p.release()
        want, wf = S._batch_features_on_host(ex, [torch.from_numpy(w) for w in waves], 16000, None, half=half)
        say("  ticket", t, "equal", bool(np.array_equal(got, want)), got.shape)
        assert np.array_equal(got, want) and list(frames) == list(wf)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:  # steady state, two in flight
        a, _ = S._batch_features_pending(ex, [torch.from_numpy(w) for w in bs[-1]], 16000, None, half=half)
        b, _ = S._batch_features_pending(ex, [torch.from_numpy(w) for w in bs[-1]], 16000, None, half=half)
        a.wait(), a.release(), b.wait(), b.release()
        n += 120
    say("  steady", round(n / (time.perf_counter() - t0)), "cuts/s (two 60-cut batches in flight, this thread only)")
    try:
        S._batch_features_pending(ex, [torch.zeros(100)], 16000, None)
        raise SystemExit("too-short cut was accepted")
    except ValueError as e:
        say("  too short:", str(e)[:80])
    ex.to("cuda:0")
    say("  moved (pipeline destroyed)")
faulthandler.cancel_dump_traceback_later()
say("ok")
