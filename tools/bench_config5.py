#!/usr/bin/env python3
"""BASELINE configs[4] / SURVEY 8d config 5, emulated without lhotse: mini-batches of 600 s of audio drawn from cut
lengths U(1, 30) s (seed 0), every cut speed-perturbed by a factor from {0.9, 1.0, 1.1}, then 80-dim fbank collated into
a padded (B, Tmax, 80) tensor with LOG_EPSILON -- what K2SpeechRecognitionDataset's OnTheFlyFeatures does per batch.
Audio starts as float32 numpy arrays in HOST memory (as decoded audio would), so the numbers are PCIe-inclusive:
pack + H2D -> resample on the device (one launch per factor) -> fbank + collation (one launch).  One JSON line.

    python tools/bench_config5.py [--batches 40]
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
from lhotse_amd import augmentation as A
from lhotse_amd.extractors import pack_to_device

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=40)
a = ap.parse_args()
rng = np.random.RandomState(0)
batches = []
for b in range(a.batches):
    lens, tot = [], 0.0
    while True:
        d = rng.uniform(1.0, 30.0)
        if tot + d > 600.0:
            break
        lens.append(int(d * 16000)); tot += d
    batches.append([(rng.rand(n).astype(np.float32) - 0.5, float(rng.choice([0.9, 1.0, 1.1]))) for n in lens])
ex = LA.HipFbank()
res = {f: A.get_or_create_resampler(round(16000 * f), 16000) for f in (0.9, 1.1)}

def run(batch, timers=None):
    t0 = time.perf_counter()
    packed, offs, lens_ = pack_to_device([x for x, _ in batch], torch.device("cuda", 0))   # pinned staging, one H2D
    dev = [packed[o : o + n] for o, n in zip(offs.tolist(), lens_.tolist())]
    if timers is not None: torch.cuda.synchronize(); t1 = time.perf_counter()
    out = list(dev)
    for f, r in res.items():
        idx = [i for i, (_, ff) in enumerate(batch) if ff == f]
        if idx:
            for i, y in zip(idx, r.resample_batch([dev[i] for i in idx])):
                out[i] = y
    if timers is not None: torch.cuda.synchronize(); t2 = time.perf_counter()
    feats, lens = ex.extract_collated(out, 16000)
    if timers is not None:
        torch.cuda.synchronize(); t3 = time.perf_counter()
        timers["h2d"] += t1 - t0; timers["resample"] += t2 - t1; timers["fbank_collate"] += t3 - t2
    return feats, lens

run(batches[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches:
    feats, lens = run(b)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
timers = {"h2d": 0.0, "resample": 0.0, "fbank_collate": 0.0}
for b in batches:
    run(b, timers)
ncuts = sum(len(b) for b in batches); secs = sum(len(x) for b in batches for x, _ in b) / 16000
print(json.dumps({"workload": f"{a.batches} batches x 600 s (cuts U(1,30) s, speed 0.9/1.0/1.1) -> 80-dim fbank, padded (B,Tmax,80), host float32 in, device out",
                  "batches_per_s": round(a.batches / wall, 1), "cuts_per_s": round(ncuts / wall, 1), "audio_seconds_per_s": round(secs / wall, 1),
                  "ms_per_batch": round(wall / a.batches * 1e3, 3),
                  "split_ms_per_batch_synchronised": {k: round(v / a.batches * 1e3, 3) for k, v in timers.items()}}))
