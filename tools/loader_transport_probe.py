#!/usr/bin/env python3
"""Where does a batch of decoded audio lose its time between a DataLoader worker and the extractor's staging buffer?  (GPU box.)
Measured per batch of 60 x 10 s float32 cuts (38 MB): waiting for the loader, touching the delivered memory (one memcpy pass into a
preallocated buffer), freeing it -- for lhotse's transport (one array per cut) and the packed one (one tensor per batch), with 4 / 16
workers, BEFORE the process has a HIP context and AFTER (plan + page-locked staging exist: KFD's MMU notifiers then see every mmap / munmap
of this process).    python tools/loader_transport_probe.py [passes]"""
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


def run(cuts, workers, packed, strategy):
    import torch.multiprocessing as mp

    mp.set_sharing_strategy(strategy)
    batches = P.batches_of(cuts)
    dst = np.empty(60 * P.SAMPLES + 1024, dtype=np.float32)
    t_wait = t_touch = t_free = t_dec = 0.0
    n = 0
    t0 = time.perf_counter()
    first = None
    it = iter(P._loader(P.DecodeDataset(cuts, packed=packed), batches, workers))
    while True:
        a = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        c = time.perf_counter()
        if first is None:
            first = c - t0
        t_dec += float(b["worker_decode_s"])
        if packed:
            x = b["audio"].numpy()
            dst[: x.size] = x
            n += len(b["lens"])
        else:
            o = 0
            for w in b["audio"]:
                x = w.numpy().reshape(-1)
                dst[o : o + x.size] = x
                o += x.size
            n += len(b["audio"])
        d = time.perf_counter()
        del b, x
        e = time.perf_counter()
        t_wait += c - a
        t_touch += d - c
        t_free += e - d
    wall = time.perf_counter() - t0
    nb = len(batches)
    return {"workers": workers, "transport": "packed" if packed else "per cut", "sharing": strategy, "cuts_per_s_behind_first_batch": round((n - 60) / (wall - first), 1),
            "first_batch_s": round(first, 3), "ms_per_batch": {"wait_for_loader": round(t_wait / nb * 1e3, 2), "touch (memcpy 38 MB)": round(t_touch / nb * 1e3, 2),
                                                               "free (munmap)": round(t_free / nb * 1e3, 2),
                                                               "decode inside a worker": round(t_dec / nb * 1e3, 2)}}


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        paths = P.write_corpus(os.path.join(td, "wav"), 64)
        cuts = P.make_cuts(paths, passes)
        out = {"cuts_per_pass": len(cuts), "cpus": len(os.sched_getaffinity(0))}
        for phase in ("no HIP context", "HIP context + plan + page-locked staging"):
            if phase.startswith("HIP"):
                if not torch.cuda.is_available():
                    break
                import lhotse_amd

                ex = lhotse_amd.HipFbank()
                ex.extract_batch([torch.rand(160000) - 0.5 for _ in range(60)], 16000)
                pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()  # noqa: F841
            rows = []
            for workers in (4, 16):
                for packed in (False, True):
                    rows.append(run(cuts, workers, packed, "file_descriptor"))
            rows.append(run(cuts, 16, True, "file_system"))
            out[phase] = rows
            for r in rows:
                print(phase, "|", json.dumps(r), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
