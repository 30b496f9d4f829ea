"""What one loader worker of the ring loader spends per cut (no GPU): WAV decode as tools/plumbing.py::read_wav does it (the stand-in for
lhotse's audio backend), the copy into the ring slot, the manifest-line halves -- wall / user / system per cut, page faults -- next to a
variant that converts int16 -> float32 straight into the slot (no temporaries; informative only: lhotse's audio I/O is out of scope).
Usage: python tools/loader_worker_probe.py [--batches 40]"""
import argparse
import os
import resource
import shutil
import sys
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)


def main():
    import plumbing as P
    from lhotse_amd.ring_loader import pack_into
    from lhotse_amd.storage import manifest_fragments

    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    a = ap.parse_args()
    d = "/dev/shm/worker_probe_wav"
    P.write_corpus(d, 64)
    paths = sorted(os.path.join(d, f) for f in os.listdir(d))
    cuts = P.make_cuts(paths, (a.batches * 60 + 63) // 64 + 1)
    template = {"type": "hip-fbank", "num_features": 80, "frame_shift": 0.01, "sampling_rate": 16000, "storage_type": "hip_archive", "storage_path": ""}
    out = np.zeros(60 * (160000 + 8) * 4, dtype=np.uint8)  # (touched: the slot's first-touch faults are not this probe's subject)
    flat = out.view(np.float32)
    rc = {}

    def ru():
        r = resource.getrusage(resource.RUSAGE_SELF)
        return r.ru_utime, r.ru_stime, r.ru_minflt

    from lhotse_amd.ring_loader import SlotWriter

    for variant in ("read_wav x 60, then pack_into (60 decoded arrays alive)", "read_wav -> SlotWriter.add per cut (leg D's worker)", "int16 -> float32 straight into the slot"):
        T = {"decode": 0.0, "pack": 0.0, "line_halves": 0.0}
        u0, t00 = ru(), time.perf_counter()
        for b in range(a.batches):
            idx = list(range(b * 60, b * 60 + 60))
            t0 = time.perf_counter()
            if variant.startswith("read_wav x 60"):
                audio = [P.read_wav(cuts[i].path, False)[0] for i in idx]
                t1 = time.perf_counter()
                pack_into(out, audio)
                del audio
            elif variant.startswith("read_wav ->"):
                w = SlotWriter(out)
                for i in idx:
                    assert w.add(P.read_wav(cuts[i].path, False)[0])
                t1 = time.perf_counter()
            else:
                o = 0
                for i in idx:
                    with wave.open(cuts[i].path, "rb") as f:
                        raw = f.readframes(f.getnframes())
                    x = np.frombuffer(raw, dtype=np.int16)
                    np.multiply(x, np.float32(1.0 / 32768.0), out=flat[o : o + x.shape[0]], dtype=np.float32)
                    o += (x.shape[0] + 3) & ~3
                t1 = time.perf_counter()
            t2 = time.perf_counter()
            [manifest_fragments(cuts[i], template, 0.01, rc) for i in idx]
            t3 = time.perf_counter()
            T["decode"] += t1 - t0
            T["pack"] += t2 - t1
            T["line_halves"] += t3 - t2
        u1 = ru()
        n = a.batches * 60
        print(f"{variant}: " + ", ".join(f"{k} {v / n * 1e3:.4f}" for k, v in T.items()) + f" ms per cut; wall {(time.perf_counter() - t00) / n * 1e3:.4f}, "
              f"user {(u1[0] - u0[0]) / n * 1e3:.4f}, system {(u1[1] - u0[1]) / n * 1e3:.4f} ms per cut; {(u1[2] - u0[2]) / n:.1f} page faults per cut", flush=True)
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
