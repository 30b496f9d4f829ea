"""CPU: the three legs of tools/plumbing.py (BASELINE configs[0]) on a tiny corpus with the oracle-backed stand-in for the device plan --
the loops, the decoding workers, the per-cut .npy path, the native archive + spliced manifest lines, and the read-back used by bench.py's
parity leg.  (Rates are measured on the GPU box by `bench.py --config plumbing` and, for the real lhotse drivers, by
tools/plumbing_reference.py in the authoring container.)"""
import gzip
import json
import os
import sys
import time

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


def test_slot_writer_and_page_locking_bookkeeping_of_the_ring():
    """SlotWriter: cuts land on 16-byte boundaries as they are added, a cut of another dtype or one that does not fit is refused without
    a trace, `arrays()` gives back what was added.  RingLoader.pin_for: every slot is handed to libhipfeat's hipfeat_host_register ONCE,
    the first time a batch is delivered in it, on a background thread; refused registrations are remembered and not retried; close()
    unregisters exactly the slots that were registered (a stand-in library records the calls: no GPU here)."""
    from lhotse_amd.ring_loader import ALIGN, RingLoader, SlotWriter

    out = np.zeros(4096, dtype=np.uint8)
    w = SlotWriter(out)
    a, b = np.arange(5, dtype=np.float32), np.arange(7, dtype=np.float32) + 100
    assert w.add(a) and w.add(b)
    assert not w.add(np.zeros(3, dtype=np.int16))  # another dtype
    assert not w.add(np.zeros(2000, dtype=np.float32))  # does not fit
    used, offs, lens = w.finish()
    assert offs.tolist() == [0, 8] and lens.tolist() == [5, 7] and used == 64 and used % ALIGN == 0
    flat = out.view(np.float32)
    assert np.array_equal(flat[0:5], a) and np.array_equal(flat[8:15], b)
    back = w.arrays()
    assert np.array_equal(back[0], a) and np.array_equal(back[1], b) and back[0].base is None

    class Lib:
        def __init__(self):
            self.calls, self.refuse = [], set()

        def raw(self, name, *args):
            self.calls.append((name, *args))
            if name == "hipfeat_host_register" and args[1] in self.refuse:
                return 3
            return 0

    def load(spec, o):
        s = SlotWriter(o)
        s.add(np.full(16, spec, dtype=np.float32))
        used, offs, lens = s.finish()
        return used, {"offs": offs, "lens": lens}

    lib = Lib()
    rl = RingLoader(load, num_workers=2, slot_bytes=4096, num_slots=4, start_method="fork")
    base = rl._ring.ctypes.data
    lib.refuse.add(base + 1 * rl.slot_bytes)  # slot 1 is refused (e.g. a locked-memory limit)
    rl.pin_for(lib, 3)
    rl.pin_for(lib, 3)  # (idempotent)
    seen = set()
    for rb in rl.batches(range(30)):
        assert rb.data.view(np.float32)[0] == rb.index
        seen.add(rb.slot)
        rb.release()
    for _ in range(200):  # the pin thread works behind the deliveries
        if len(rl._pin[4]) == len(seen) and len([c for c in lib.calls if c[0] == "hipfeat_host_register"]) == len(seen):
            break
        time.sleep(0.01)
    regs = [c for c in lib.calls if c[0] == "hipfeat_host_register"]
    assert sorted(c[2] for c in regs) == sorted(base + s * rl.slot_bytes for s in seen) and all(c[1] == 3 and c[3] == rl.slot_bytes for c in regs)
    assert len(regs) == len(set(regs))  # once per slot
    assert rl.pinned_slots() == len(seen - {1})
    rl.close()
    unreg = [c[1] for c in lib.calls if c[0] == "hipfeat_host_unregister"]
    assert sorted(unreg) == sorted(base + s * rl.slot_bytes for s in seen - {1})
    rl.close()  # (idempotent)


def test_per_cut_driver_leg_with_the_hip_extractor(tmp_path, cpu_plan):
    """Leg E: the per-cut driver of leg A (CutSet.compute_and_store_features, lhotse/cut/set.py:2141-2195) with HipFbank in place of the
    reference's Fbank -- forked job processes, each with its own extractor; same files, same manifest fields, the extractor's name."""
    import plumbing as P

    from oracle.kaldi_torch import TorchFbank

    paths = P.write_corpus(str(tmp_path / "wav"), n_files=3, seed=5)
    cuts = P.make_cuts(paths, 2)
    e = P.cpu_per_cut(cuts, str(tmp_path / "e"), num_jobs=2, extractor="hip")
    assert e["cuts"] == 6 and e["errors"] is None and e["extractor"] == "hip"
    for j in range(2):
        with gzip.open(tmp_path / "e" / f"cuts-{j}.jsonl.gz", "rt") as f:
            for ln, k in zip(f, range(j, 6, 2)):
                d = json.loads(ln)
                assert d["id"] == cuts[k].id and d["features"]["type"] == "hip-fbank" and d["features"]["num_frames"] == 1000
                got = np.load(os.path.join(d["features"]["storage_path"], d["features"]["storage_key"]))
                want = TorchFbank().extract(P.read_wav(cuts[k].path)[0])
                assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4


def test_ring_leg_with_lhotse_s_numpy_files_storage(tmp_path, cpu_plan):
    """Leg F: the ring's transport in front of lhotse's own per-cut storage (one .npy + one manifest line per cut)."""
    import plumbing as P

    import lhotse_amd as LA
    from oracle.kaldi_torch import TorchFbank

    paths = P.write_corpus(str(tmp_path / "wav"), n_files=3, seed=5)
    cuts = P.make_cuts(paths, 2)
    f = P.hip_ring_numpy_files(LA.HipFbank(LA.HipFbankConfig(device="cpu")), cuts, str(tmp_path / "f"), num_workers=2)
    assert f["cuts"] == 6 and f["storage"] == "numpy_files"
    with gzip.open(f["manifest"], "rt") as fh:
        lines = [json.loads(ln) for ln in fh]
    assert [d["id"] for d in lines] == [c.id for c in cuts]
    for k in (0, 5):
        got = np.load(os.path.join(lines[k]["features"]["storage_path"], lines[k]["features"]["storage_key"]))
        want = TorchFbank().extract(P.read_wav(cuts[k].path)[0])
        assert got.shape == (1000, 80) and np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4
