#!/usr/bin/env python3
"""Follow-up of tools/loader_transport_probe.py (GPU box): a batch that arrives from a DataLoader worker as ONE shared-memory tensor goes
into the library's host pipeline (pack into page-locked staging by a pool of copy threads -> H2D -> kernel -> D2H).  Per batch: submit +
wait, (a) as delivered, (b) after the calling thread has touched every page once (single-threaded first touch), (c) from a private copy
of the tensor (ordinary anonymous memory), each with HIPFEAT_COPY_THREADS = 1 / default.
    python tools/loader_pipeline_probe.py [passes]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import plumbing as P


def run(cuts, workers, mode, threads, context=None, fork_first=False):
    import lhotse_amd

    if threads:
        os.environ["HIPFEAT_COPY_THREADS"] = str(threads)
    else:
        os.environ.pop("HIPFEAT_COPY_THREADS", None)
    batches = P.batches_of(cuts)
    ex = lhotse_amd.HipFbank()
    it = None
    if fork_first:  # the worker processes exist BEFORE this process has a plan / page-locked memory (lhotse's driver: the plan is created
        it = iter(P._loader(P.DecodeDataset(cuts, packed=True), batches, workers, context))  # lazily, at the first extract_batch)
    ex.extract_batch([torch.rand(160000) - 0.5 for _ in range(60)], 16000)
    pipe = ex._native_pipe()
    t_wait = t_prep = t_submit = t_done = 0.0
    n = 0
    first = None
    t0 = time.perf_counter()
    if it is None:
        it = iter(P._loader(P.DecodeDataset(cuts, packed=True), batches, workers, context))
    for b in it:
        a = time.perf_counter()
        if first is None:
            first = a - t0
        buf, offs, lens = b["audio"], b["offs"].tolist(), b["lens"].tolist()
        if mode == "touched":
            float(buf[::1024].sum())  # one read per page: first touch on this thread
        elif mode == "private copy":
            buf = buf.clone()
        c = time.perf_counter()
        p = ex.submit_host_items([buf[o : o + k] for o, k in zip(offs, lens)], 16000)
        d = time.perf_counter()
        p.wait()
        p.release()
        e = time.perf_counter()
        t_prep += c - a
        t_submit += d - c
        t_done += e - d
        n += len(lens)
    wall = time.perf_counter() - t0
    nb = len(batches)
    st = pipe.stats()
    ex._drop_plan()
    return {"mode": mode, "start_method": context or "fork (default)", "workers_started": "before the plan" if fork_first else "after the plan",
            "copy_threads": pipe.threads, "workers": workers, "cuts_per_s_behind_first_batch": round((n - 60) / (wall - first), 1),
            "ms_per_batch": {"prepare": round(t_prep / nb * 1e3, 2), "submit": round(t_submit / nb * 1e3, 2), "wait (pack + H2D + kernel + D2H)": round(t_done / nb * 1e3, 2),
                             "pipeline thread packing": round(st["pack_s"] / nb * 1e3, 2), "pipeline thread busy": round(st["busy_s"] / nb * 1e3, 2)}}


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        paths = P.write_corpus(os.path.join(td, "wav"), 64)
        cuts = P.make_cuts(paths, passes)
        if os.environ.get("HIPFEAT_WAIT"):
            r = run(cuts, 8, "as delivered", 0)
            r["HIPFEAT_WAIT"] = os.environ["HIPFEAT_WAIT"]
            print(json.dumps(r), flush=True)
            return
        print(json.dumps(run(cuts, 8, "as delivered", 0)), flush=True)
        print(json.dumps(run(cuts, 8, "as delivered", 0, fork_first=True)), flush=True)
        print(json.dumps(run(cuts, 8, "as delivered", 0, context="spawn")), flush=True)
        print(json.dumps(run(cuts, 8, "as delivered", 0, context="forkserver")), flush=True)
        print(json.dumps(run(cuts, 8, "private copy", 0, context="forkserver")), flush=True)
        print(json.dumps(run(cuts, 0, "as delivered", 0)), flush=True)


if __name__ == "__main__":
    main()
