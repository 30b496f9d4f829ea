#!/usr/bin/env python3
"""Plain collated extraction of 600 s mini-batches (no speed perturbation), device resident: plan.run_collated through the launch pair of
round 4 (default) or through hipfeat_extract_collated (HIPFEAT_COLLATED_NO_PAIR=1).  One JSON line.  (GPU box)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd as LA

ex = LA.HipFbank()
plan = ex.plan
rng = np.random.RandomState(0)
batches = []
for b in range(64):
    lens, tot = [], 0.0
    while True:
        d = rng.uniform(1.0, 30.0)
        if tot + d > 600.0:
            break
        lens.append(int(d * 16000)); tot += d
    lens = np.asarray(lens, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum((lens + 3) & ~3)[:-1]]).astype(np.int64)
    wave = torch.empty(int(offs[-1] + lens[-1]), device="cuda").uniform_(-0.5, 0.5)
    batches.append((wave, offs, lens))
cuts = sum(len(b[2]) for b in batches)
def step():
    for w, o, l in batches:
        plan.run_collated(w, o, l, None, -23.025850929940457)
for _ in range(5): step()
torch.cuda.synchronize()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    step(); n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
host = []
for i in range(0, 64, 8):
    torch.cuda.synchronize(); h0 = time.perf_counter()
    for w, o, l in batches[i:i + 8]:
        plan.run_collated(w, o, l, None, -23.025850929940457)
    host.append((time.perf_counter() - h0) / 8)
torch.cuda.synchronize()
print(json.dumps({"route": "hipfeat_extract_collated" if os.environ.get("HIPFEAT_COLLATED_NO_PAIR") else "launch pair (hipfeat_minibatch_*)",
                  "cuts_per_s": round(cuts * n / dt, 1), "us_per_minibatch": round(dt / (n * 64) * 1e6, 2), "host_us_per_minibatch": round(sorted(host)[len(host) // 2] * 1e6, 2)}))
