#!/usr/bin/env python3
"""
BASELINE configs[0] as written -- "CPU Fbank (n_mels=80, 16 kHz, 25/10 ms) via compute_and_store_features (plumbing, no GPU)" -- and
SURVEY 8d baseline C, under the REAL lhotse, in the authoring container (needs /root/reference; no GPU here, none needed):

    python tools/plumbing_reference.py > profiles/r06_plumbing_container.json

64 int16 WAV files (10 s, 16 kHz; the corpus of tools/plumbing.py) on tmpfs -> RecordingSet / CutSet ->

  R1  CutSet.compute_and_store_features(Fbank(), NumpyFilesWriter, num_jobs=1)              lhotse/cut/set.py:1981-2195
  R2  ... num_jobs = ncores (worker processes forked: the three stub modules this container needs to import lhotse do not survive a spawn)
  R3  CutSet.compute_and_store_features_batch(Fbank(), NumpyFilesWriter, num_workers=4)     lhotse/cut/set.py:2197-2408
  A1 / A2  tools/plumbing.py::cpu_per_cut with num_jobs = 1 / ncores -- the restated loop that `bench.py --config plumbing` times on the
           GPU box, here on the same files and cores as R1 / R2, so that "port" can be read against "reference"
  D   the drop-in itself: the same three drivers with HipFbank over the oracle-backed CPU stand-in of the device plan -- proves the plumbing
      (registry, pickling into the jobs, manifests, validate_features); its RATE is meaningless (numpy stand-in) and is not reported

torch.set_num_threads(1) throughout, as the CLI does (lhotse/bin/modes/features.py:25-32).
"""
from __future__ import annotations

import json
import multiprocessing
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch

    torch.set_num_threads(1)
    import plumbing as P
    from _dropin_support import import_lhotse, install_wave_backend, make_cpu_plan

    import_lhotse()
    install_wave_backend()
    from lhotse import CutSet, MonoCut, Recording, SupervisionSegment
    from lhotse.audio import AudioSource
    from lhotse.features.io import NumpyFilesWriter
    from lhotse.features.kaldi.extractors import Fbank

    ncpu = len(os.sched_getaffinity(0))
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    out = {"host": {"logical_cores": ncpu, "cpu_model": next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?"),
                    "torch_threads": torch.get_num_threads(), "storage": "tmpfs (/dev/shm)" if base else "tmp"},
           "workload": "BASELINE configs[0] / SURVEY 8d Config 1: 64 x 10 s 16 kHz mono int16 WAV files (seeded noise) -> CutSet -> Fbank() (80 mel, 25/10 ms) -> NumpyFilesWriter"}
    with tempfile.TemporaryDirectory(prefix="plumb_ref_", dir=base) as td:
        paths = P.write_corpus(os.path.join(td, "wav"), 64, seed=0)

        def cutset(repeat):
            cuts = []
            for r in range(repeat):
                for i, p in enumerate(paths):
                    rec = Recording(id=f"rec{i:03d}", sources=[AudioSource(type="file", channels=[0], source=p)], sampling_rate=P.SR, num_samples=P.SAMPLES,
                                    duration=P.SAMPLES / P.SR)
                    cid = f"cut-{r * 64 + i:07d}"
                    sup = SupervisionSegment(id=cid, recording_id=rec.id, start=0.0, duration=rec.duration, channel=0, text="SYNTHETIC UTTERANCE " * 4,
                                             language="English", speaker=f"spk{(r * 64 + i) % 251}")
                    cuts.append(MonoCut(id=cid, start=0, duration=rec.duration, channel=0, recording=rec, supervisions=[sup]))
            return CutSet.from_cuts(cuts)

        def timed(tag, fn, n_cuts):
            d = os.path.join(td, tag)
            t0 = time.perf_counter()
            res = fn(d)
            n = sum(1 for _ in res)
            dt = time.perf_counter() - t0
            shutil.rmtree(d, ignore_errors=True)
            assert n == n_cuts, (tag, n, n_cuts)
            return {"cuts_per_s": round(n / dt, 1), "cuts": n, "seconds": round(dt, 2)}

        ref = {}
        cs = cutset(4)  # 256 cuts
        cs.subset(first=8).compute_and_store_features(extractor=Fbank(), storage_path=os.path.join(td, "warm"), num_jobs=1, storage_type=NumpyFilesWriter)
        ref["R1 compute_and_store_features(Fbank(), NumpyFilesWriter, num_jobs=1)"] = timed(
            "r1", lambda d: cs.compute_and_store_features(extractor=Fbank(), storage_path=d, num_jobs=1, storage_type=NumpyFilesWriter), len(cs))
        big = cutset(4 * max(1, ncpu // 2))
        ex = ProcessPoolExecutor(ncpu, mp_context=multiprocessing.get_context("fork"))
        list(ex.map(abs, range(ncpu * 4)))  # workers started before the clock, as in leg A (a corpus-sized run amortises process start-up)
        ref[f"R2 compute_and_store_features(Fbank(), NumpyFilesWriter, num_jobs={ncpu}) [forked workers, started before the clock]"] = timed(
            "r2", lambda d: big.compute_and_store_features(extractor=Fbank(), storage_path=d, num_jobs=ncpu, executor=ex, storage_type=NumpyFilesWriter), len(big))
        ex.shutdown()
        ref["R3 compute_and_store_features_batch(Fbank(), NumpyFilesWriter, num_workers=4, batch_duration=600, collate=False)"] = timed(
            "r3", lambda d: cs.compute_and_store_features_batch(extractor=Fbank(), storage_path=d, manifest_path=d + ".jsonl.gz", batch_duration=600.0, num_workers=4,
                                                                collate=False, storage_type=NumpyFilesWriter), len(cs))
        out["reference_drivers (kind: reference)"] = ref

        port = {}
        port["A1 tools/plumbing.py::cpu_per_cut(num_jobs=1)"] = P.cpu_per_cut(P.make_cuts(paths, 4), os.path.join(td, "a1"), 1)
        port[f"A2 tools/plumbing.py::cpu_per_cut(num_jobs={ncpu})"] = P.cpu_per_cut(P.make_cuts(paths, 4 * max(1, ncpu // 2)), os.path.join(td, "a2"), ncpu)
        out["restated_loops (kind: port)"] = port
        r1 = ref[next(k for k in ref if k.startswith("R1"))]["cuts_per_s"]
        r2 = ref[next(k for k in ref if k.startswith("R2"))]["cuts_per_s"]
        out["port_over_reference"] = {"num_jobs=1": round(port[next(k for k in port if k.startswith("A1"))]["cuts_per_s"] / r1, 3),
                                      f"num_jobs={ncpu}": round(port[next(k for k in port if k.startswith("A2"))]["cuts_per_s"] / r2, 3),
                                      "what": "the restated per-cut loop against lhotse's own driver, same files, same cores: the loop carries no Recording / "
                                              "MonoCut objects, no fastcopy, no validate_features -- it is the LIGHTER of the two, so its rate is an upper bound "
                                              "on what the reference driver would show on the GPU box's host"}

        # D: the drop-in under the real drivers (oracle-backed stand-in for the device: plumbing proof, not a rate)
        import lhotse_amd as LA
        import lhotse_amd.extractors as E

        E._Plan = make_cpu_plan()
        small = cutset(1).subset(first=12)
        hip = LA.HipFbank(LA.HipFbankConfig(device="cpu"))
        d1 = small.compute_and_store_features(extractor=hip, storage_path=os.path.join(td, "d1"), num_jobs=1, storage_type=NumpyFilesWriter)
        ex = ProcessPoolExecutor(2, mp_context=multiprocessing.get_context("fork"))
        d2 = small.compute_and_store_features(extractor=hip, storage_path=os.path.join(td, "d2"), num_jobs=2, executor=ex, storage_type=NumpyFilesWriter)
        d3 = small.compute_and_store_features_batch(extractor=hip, storage_path=os.path.join(td, "d3"), manifest_path=os.path.join(td, "d3.jsonl.gz"),
                                                    batch_duration=60.0, num_workers=2, storage_type=NumpyFilesWriter)
        want = {c.id: Fbank().extract(c.load_audio(), 16000) for c in small}
        import numpy as np

        worst = 0.0
        for res in (d1, d2, d3):
            for c in res:
                f = c.load_features()
                assert c.features.type == "hip-fbank" and f.shape == want[c.id].shape == (1000, 80)
                worst = max(worst, float(np.linalg.norm(f - want[c.id]) / np.linalg.norm(want[c.id])))
        out["drop_in_under_the_real_drivers"] = {"drivers": ["compute_and_store_features num_jobs=1", "num_jobs=2 (pickled into worker processes)",
                                                             "compute_and_store_features_batch num_workers=2"],
                                                 "cuts_each": len(small), "features_type": "hip-fbank", "worst_rel_l2_vs_reference_Fbank": worst,
                                                 "device": "oracle-backed CPU stand-in of the plan (no GPU in this container): a plumbing proof, its rate is not reported"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
