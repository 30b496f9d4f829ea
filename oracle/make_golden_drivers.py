#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE -- generates tests/golden/drivers.npz + drivers.json by running LHOTSE'S OWN DRIVERS with the reference's CPU Fbank.

Run in the authoring container only (needs /root/reference):

    python oracle/make_golden_drivers.py

What is recorded (VERDICT r5 Missing #1: the product has never met lhotse's callers on a GPU -- the reference cannot travel to the GPU
box, its OUTPUTS can):

  batch      CutSet.compute_and_store_features_batch(Fbank(), NumpyFilesWriter, num_workers=2, batch_duration=10, collate=False | True)
             [batches of 5 + 3 cuts.  NB with collate=True the reference driver CRASHES on a batch of ONE cut: _extract_batch returns the bare
             (T, F) matrix for a single item (kaldi/extractors.py:542-546) and _save_worker then iterates its rows (cut/set.py:2322-2330,
             IndexError) -- found while generating these fixtures with batch_duration=4 ... 9 (the sampler's budget counts the PADDED batch, so 6 s gives 3 + 2 + 2 + 1 cuts); the product returns (1, T, F) there]
             (lhotse/cut/set.py:2197-2408: DataLoader workers -> extract_batch on the main thread -> save thread)
  per_cut    CutSet.compute_and_store_features(Fbank(), NumpyFilesWriter, num_jobs=2)
             (lhotse/cut/set.py:1981-2195: the extractor pickled into worker processes, one `extract` per cut)
  k2         K2SpeechRecognitionDataset(input_strategy=OnTheFlyFeatures(Fbank()), cut_transforms=[PerturbSpeed([0.9, 1.1], p=1, Random(0))])
             (lhotse/dataset/speech_recognition.py:94-134, dataset/input_strategies.py:410-476, cut_transforms/perturb_speed.py:8-47)
             and the same without the perturbation
  registry   the keys of FEATURE_EXTRACTORS after `import lhotse_amd` (lhotse/features/base.py:391-405; bin/modes/features.py:40)

per cut: the stored matrix, and the manifest fields a reader relies on (id, num_frames, num_features, frame_shift, sampling_rate, start,
duration, type).  tests/test_gpu_reference_drivers.py regenerates the corpus on the GPU box and holds the real HIP plan, driven through
the same call shapes, to these outputs.
"""
from __future__ import annotations

import json
import os
import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.driver_corpus import CORPUS, SAMPLING_RATE, write_corpus  # noqa: E402


BATCH_DURATION = 10.0


def build_cutset(files):
    from lhotse import CutSet, MonoCut, Recording, SupervisionSegment
    from lhotse.audio import AudioSource

    cuts = []
    for f in files:
        dur = f["num_samples"] / SAMPLING_RATE
        rec = Recording(id=f"rec-{f['id']}", sources=[AudioSource(type="file", channels=[0], source=f["path"])], sampling_rate=SAMPLING_RATE,
                        num_samples=f["num_samples"], duration=dur)
        sup = SupervisionSegment(id=f"sup-{f['id']}", recording_id=rec.id, start=0.0, duration=dur, channel=0, text=f"text of {f['id']}")
        cuts.append(MonoCut(id=f["id"], start=0, duration=dur, channel=0, recording=rec, supervisions=[sup]))
    return CutSet.from_cuts(cuts)


def manifest_fields(cut):
    f = cut.features
    return {"id": cut.id, "num_frames": int(f.num_frames), "num_features": int(f.num_features), "frame_shift": float(f.frame_shift),
            "sampling_rate": int(f.sampling_rate), "start": float(f.start), "duration": float(f.duration), "type": f.type,
            "num_samples": int(cut.num_samples)}


def compact(arrays, meta):
    """The drivers differ from `per_cut` (every cut framed on its own) only in the last rows of the shorter items of a zero-padded batch
    (SURVEY Q1), so every other matrix is stored as (rows that differ, their values) against its per_cut base; `expand` restores them."""
    out = {}
    for k, v in arrays.items():
        group, _, name = k.partition("/")
        if group in ("batch_collate0", "batch_collate1"):
            base = arrays[f"per_cut/{name}"]
            assert v.shape == base.shape
            rows = np.nonzero((v != base).any(axis=1))[0]
            out[f"{k}@rows"], out[f"{k}@vals"] = rows.astype(np.int32), v[rows]
        elif k == "k2_plain/inputs":
            for i, cid in enumerate(meta["k2_plain"]["cut_ids"]):
                base = arrays[f"per_cut/{cid}"]
                got = v[i, : len(base)]
                rows = np.nonzero((got != base).any(axis=1))[0]
                out[f"k2_plain/{cid}@rows"], out[f"k2_plain/{cid}@vals"] = rows.astype(np.int32), got[rows]
                assert (v[i, len(base):] == v[i, -1, -1]).all() or len(base) == v.shape[1]
            out["k2_plain/shape"] = np.array(v.shape, dtype=np.int32)
        elif k == "k2_speed/inputs":
            nf = arrays["k2_speed/num_frames"]
            for i, cid in enumerate(meta["k2_speed"]["cut_ids"]):
                out[f"k2_speed/{cid}"] = v[i, : int(nf[i])]
            out["k2_speed/shape"] = np.array(v.shape, dtype=np.int32)
        else:
            out[k] = v
    return out


def expand(z, meta):
    """Inverse of `compact` (used by tests/_golden.load_driver_goldens)."""
    LOG_EPS = np.float32(-23.025850929940457)
    out = {k: z[k] for k in z if "@" not in k and not k.endswith("/shape") and not (k.startswith("k2_speed/utt"))}
    for k in z:
        if k.endswith("@rows") and not k.startswith("k2_plain/"):
            name = k[: -len("@rows")]
            m = z[f"per_cut/{name.partition('/')[2]}"].copy()
            m[z[k]] = z[f"{name}@vals"]
            out[name] = m
    for tag in ("k2_plain", "k2_speed"):
        shape = tuple(int(x) for x in z[f"{tag}/shape"])
        full = np.full(shape, LOG_EPS, dtype=np.float32)
        for i, cid in enumerate(meta[tag]["cut_ids"]):
            if tag == "k2_plain":
                m = z[f"per_cut/{cid}"].copy()
                m[z[f"k2_plain/{cid}@rows"]] = z[f"k2_plain/{cid}@vals"]
            else:
                m = z[f"k2_speed/{cid}"]
            full[i, : len(m)] = m
        out[f"{tag}/inputs"] = full
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    from _dropin_support import import_lhotse, install_wave_backend

    import_lhotse()
    install_wave_backend()
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor

    from lhotse.dataset import K2SpeechRecognitionDataset
    from lhotse.dataset.cut_transforms import PerturbSpeed
    from lhotse.dataset.input_strategies import OnTheFlyFeatures
    from lhotse.features.base import FEATURE_EXTRACTORS
    from lhotse.features.io import NumpyFilesWriter
    from lhotse.features.kaldi.extractors import Fbank

    arrays, meta = {}, {"corpus": [list(c) for c in CORPUS], "sampling_rate": SAMPLING_RATE}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        files = write_corpus(td / "wav")
        meta["files"] = [{k: v for k, v in f.items() if k != "path"} for f in files]
        cuts = build_cutset(files)

        from lhotse.dataset import SimpleCutSampler

        meta["batch_duration"] = BATCH_DURATION
        meta["batches"] = [[c.id for c in b] for b in SimpleCutSampler(cuts, max_duration=BATCH_DURATION)]  # the sampler the driver builds (cut/set.py:2296-2300)
        assert min(len(b) for b in meta["batches"]) >= 2, meta["batches"]
        for collate in (False, True):
            tag = f"batch_collate{int(collate)}"
            out = cuts.compute_and_store_features_batch(extractor=Fbank(), storage_path=td / tag, manifest_path=td / f"{tag}.jsonl.gz",
                                                        batch_duration=BATCH_DURATION, num_workers=2, collate=collate, storage_type=NumpyFilesWriter)
            out = sorted(out, key=lambda c: c.id)
            meta[tag] = [manifest_fields(c) for c in out]
            for c in out:
                arrays[f"{tag}/{c.id}"] = c.load_features()

        ex = ProcessPoolExecutor(2, mp_context=multiprocessing.get_context("fork"))  # (the stub modules of this container do not survive a spawn)
        out = cuts.compute_and_store_features(extractor=Fbank(), storage_path=td / "per_cut", num_jobs=2, executor=ex, storage_type=NumpyFilesWriter)
        out = sorted(out, key=lambda c: c.id)
        meta["per_cut"] = [manifest_fields(c) for c in out]
        for c in out:
            arrays[f"per_cut/{c.id}"] = c.load_features()

        for tag, tf in (("k2_plain", []), ("k2_speed", [PerturbSpeed(factors=[0.9, 1.1], p=1.0, randgen=random.Random(0))])):
            ds = K2SpeechRecognitionDataset(input_strategy=OnTheFlyFeatures(Fbank()), cut_transforms=tf, return_cuts=True)
            batch = ds[cuts]
            sup = batch["supervisions"]
            bc = sup["cut"]
            meta[tag] = {
                "cut_ids": [c.id for c in bc],
                "source_ids": [c.id.split("_sp")[0] for c in bc],
                "speed_factors": [float(getattr(c.recording.transforms[0], "factor", 1.0)) if c.recording.transforms else 1.0 for c in bc]
                if tf else [1.0] * len(bc),
                "num_samples": [int(c.num_samples) for c in bc],
                "durations": [float(c.duration) for c in bc],
                "text": list(sup["text"]),
            }
            arrays[f"{tag}/inputs"] = batch["inputs"].numpy()
            for k in ("sequence_idx", "start_frame", "num_frames"):
                arrays[f"{tag}/{k}"] = sup[k].numpy()

        import lhotse_amd  # noqa: F401

        meta["registry_hip_names"] = sorted(k for k in FEATURE_EXTRACTORS if k.startswith("hip-"))
        meta["registry_reference_names_sample"] = sorted(k for k in FEATURE_EXTRACTORS if k.startswith("kaldi-"))

    out_dir = os.path.join(ROOT, "tests", "golden")
    full_arrays = arrays
    arrays = compact(arrays, meta)
    np.savez_compressed(os.path.join(out_dir, "drivers.npz"), **arrays)
    with open(os.path.join(out_dir, "drivers.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    z = dict(np.load(os.path.join(out_dir, "drivers.npz")))
    back = expand(z, meta)
    for k, v in full_arrays.items():
        assert np.array_equal(back[k], v), k
    print(f"wrote {len(arrays)} arrays, {os.path.getsize(os.path.join(out_dir, 'drivers.npz')) / 1e3:.0f} KB")
    for tag in ("k2_plain", "k2_speed"):
        print(tag, meta[tag]["cut_ids"], meta[tag]["speed_factors"], full_arrays[f"{tag}/inputs"].shape)


if __name__ == "__main__":
    main()
