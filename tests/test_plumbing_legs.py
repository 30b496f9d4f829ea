"""CPU: the three legs of tools/plumbing.py (BASELINE configs[0]) on a tiny corpus with the oracle-backed stand-in for the device plan --
the loops, the decoding workers, the per-cut .npy path, the native archive + spliced manifest lines, and the read-back used by bench.py's
parity leg.  (Rates are measured on the GPU box by `bench.py --config plumbing` and, for the real lhotse drivers, by
tools/plumbing_reference.py in the authoring container.)"""
import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture()
def cpu_plan(monkeypatch):
    import lhotse_amd.extractors as E
    from _dropin_support import make_cpu_plan

    monkeypatch.setattr(E, "_Plan", make_cpu_plan())


def test_legs_on_a_tiny_corpus(tmp_path, cpu_plan):
    import plumbing as P

    import lhotse_amd as LA
    from oracle.kaldi_torch import TorchFbank

    paths = P.write_corpus(str(tmp_path / "wav"), n_files=3, seed=5)
    cuts = P.make_cuts(paths, 2)
    assert len(cuts) == 6 and P.read_wav(paths[0]).shape == (1, P.SAMPLES) and P.read_wav(paths[0], pcm16=True).dtype == np.int16
    want = [TorchFbank().extract(P.read_wav(c.path)[0]) for c in cuts]
    # A: two forked single-threaded jobs, .npy per cut + one manifest per job
    a = P.cpu_per_cut(cuts, str(tmp_path / "a"), num_jobs=2)
    assert a["cuts"] == 6 and a["errors"] is None
    for j in range(2):
        with gzip.open(tmp_path / "a" / f"cuts-{j}.jsonl.gz", "rt") as f:
            for ln, k in zip(f, range(j, 6, 2)):
                d = json.loads(ln)
                assert d["id"] == cuts[k].id and d["features"]["num_frames"] == 1000 and d["features"]["type"] == "kaldi-fbank"
                assert np.array_equal(np.load(os.path.join(d["features"]["storage_path"], d["features"]["storage_key"])), want[k])
    ex = LA.HipFbank(LA.HipFbankConfig(device="cpu"))
    # B: loader workers -> extract_batch on the main thread -> one save thread
    keep = {0: None, 5: None}
    b = P.hip_batch_numpy_files(ex, cuts, str(tmp_path / "b"), num_workers=2, keep=keep)
    assert b["cuts"] == 6 and 0.0 <= b["save_thread_busy_share"] <= 1.5
    with gzip.open(tmp_path / "b" / "cuts.jsonl.gz", "rt") as f:
        lines = [json.loads(ln) for ln in f]
    assert [d["id"] for d in lines] == [c.id for c in cuts] and lines[0]["features"]["type"] == "hip-fbank"
    for k in (0, 5):
        got = np.load(os.path.join(lines[k]["features"]["storage_path"], lines[k]["features"]["storage_key"]))
        assert np.array_equal(got, keep[k]) and np.linalg.norm(got - want[k]) / np.linalg.norm(want[k]) <= 1e-4
    # C: the product's bulk driver; what the manifest lines point at is what extract computes
    for pcm16, half in ((False, False), (False, True)):  # (int16 input is converted on the DEVICE: GPU box only, bench.py --config plumbing)
        c = P.hip_bulk(ex, cuts, str(tmp_path / f"c{int(half)}"), num_workers=2, pcm16=pcm16, half=half, stripes=2)
        assert c["cuts"] == 6 and len(c["archive_paths"]) == 2
        for k in (0, 3, 5):
            got = P.read_back(c, k)
            assert got.shape == (1000, 80)
            assert np.linalg.norm(got - want[k]) / np.linalg.norm(want[k]) <= (2e-3 if half else 1e-4)


def test_ring_loader_leg_and_the_loader_itself(tmp_path, cpu_plan):
    """Leg D: the shared-memory ring loader (lhotse_amd/ring_loader.py) feeding the bulk driver -- same stored features as leg C; and the
    loader on its own: submission order, slot recycling under a slow consumer, an error inside a worker reaching the consumer."""
    import plumbing as P

    import lhotse_amd as LA
    from lhotse_amd.ring_loader import RingLoader, pack_into
    from oracle.kaldi_torch import TorchFbank

    paths = P.write_corpus(str(tmp_path / "wav"), n_files=3, seed=5)
    cuts = P.make_cuts(paths, 2)
    ex = LA.HipFbank(LA.HipFbankConfig(device="cpu"))
    d = P.hip_ring(ex, cuts, str(tmp_path / "d"), num_workers=2, stripes=2)
    assert d["cuts"] == 6 and "shared ring" in d["transport"]
    for k in (0, 5):
        want = TorchFbank().extract(P.read_wav(cuts[k].path)[0])
        got = P.read_back(d, k)
        assert got.shape == (1000, 80) and np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4

    def load(spec, out):  # spec = (seed, n): n floats of RandomState(seed), or an error
        if spec[0] < 0:
            raise ValueError("boom")
        a = np.random.RandomState(spec[0]).rand(spec[1]).astype(np.float32)
        used, offs, lens = pack_into(out, [a, a[:7]])
        return used, {"offs": offs, "lens": lens, "seed": spec[0]}

    with RingLoader(load, num_workers=3, slot_bytes=1 << 16, num_slots=4, start_method="fork") as rl:
        held = []
        for i, rb in enumerate(rl.batches([(s, 1000 + s) for s in range(40)])):
            assert rb.index == i and rb.meta["seed"] == i
            flat = rb.data.view(np.float32)
            o, n = rb.meta["offs"].tolist(), rb.meta["lens"].tolist()
            assert n == [1000 + i, 7] and o[1] % 4 == 0
            assert np.array_equal(flat[o[0] : o[0] + n[0]], np.random.RandomState(i).rand(1000 + i).astype(np.float32))
            held.append(rb)  # a consumer that keeps three batches (of four slots) before giving them back
            if len(held) == 3:
                held.pop(0).release()
        del held
        with pytest.raises(RuntimeError, match="boom"):
            list(rl.batches([(1, 10), (-1, 0), (2, 10)]))
