"""GPU: the plumbing legs of tools/plumbing.py (BASELINE configs[0] with the GPU in it) on the REAL plan -- the product's bulk driver behind a
torch DataLoader (leg C) and behind the shared-memory ring whose slots are page-locked for the GPU (leg D, the default loader of
lhotse_amd.compute_and_store_features_batch) must store the same bits under the same manifest lines; both must be what the oracle
computes on the decoded files.  lhotse's side: CutSet.compute_and_store_features_batch (lhotse/cut/set.py:2197-2408); the real-lhotse
equivalence of the two loaders is tests/test_lhotse_dropin.py::test_ring_loader_behind_the_batch_driver_equals_the_dataloader."""
import gzip
import os
import sys

import numpy as np
import pytest

import lhotse_amd as LA
from _golden import ref32
from oracle.kaldi_ref import RefConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _lines(path, strip):
    with gzip.open(path, "rt") as f:
        return [ln.replace(strip, "@") for ln in f]


@pytest.mark.parametrize("pcm16,half", [(False, False), (True, True)])
def test_ring_leg_stores_what_the_dataloader_leg_stores(tmp_path, pcm16, half):
    import plumbing as P

    paths = P.write_corpus(str(tmp_path / "wav"), n_files=5, seed=11)
    cuts = P.make_cuts(paths, 60)  # 300 cuts = 5 batches of 60: the ring's slots are reused, the later batches are uploaded out of the ring
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    c = P.hip_bulk(ex, cuts, str(tmp_path / "c"), num_workers=2, pcm16=pcm16, half=half, stripes=2, context="forkserver")
    d = P.hip_ring(ex, cuts, str(tmp_path / "d"), num_workers=2, pcm16=pcm16, half=half, stripes=2, context="forkserver")
    assert c["cuts"] == d["cuts"] == 300 and d["batches"] == 5
    for a, b in zip(c["archive_paths"], d["archive_paths"]):
        with open(a, "rb") as fa, open(b, "rb") as fb:
            assert fa.read() == fb.read()
    assert _lines(c["manifest"], str(tmp_path / "c")) == _lines(d["manifest"], str(tmp_path / "d"))
    o32 = ref32(RefConfig(kind="fbank"))
    for k in (0, 61, 299):
        want = o32.extract(P.read_wav(cuts[k].path)[0])
        got = P.read_back(d, k).astype(np.float32)
        assert got.shape == want.shape == (1000, 80)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= (2e-3 if half else 1e-4)


def test_ring_slots_are_page_locked_and_later_batches_take_the_direct_route(tmp_path):
    import plumbing as P

    paths = P.write_corpus(str(tmp_path / "wav"), n_files=4, seed=3)
    cuts = P.make_cuts(paths, 600)  # 2400 cuts = 40 batches over 2 workers' 16 slots
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    d = P.hip_ring(ex, cuts, str(tmp_path / "d"), num_workers=2, stripes=2, context="forkserver")
    assert d["batches"] == 40 and d["ring_slots_page_locked"] >= 2
    assert d["batches_uploaded_straight_from_the_ring"] >= 8  # (every slot's first batch goes through staging; 2 workers own 16 slots)
    e = P.hip_ring(ex, cuts, str(tmp_path / "e"), num_workers=2, stripes=2, context="forkserver", pin=False)
    assert e["ring_slots_page_locked"] == 0 and e["batches_uploaded_straight_from_the_ring"] == 0
    for a, b in zip(d["archive_paths"], e["archive_paths"]):
        with open(a, "rb") as fa, open(b, "rb") as fb:
            assert fa.read() == fb.read()


def test_per_cut_driver_leg_on_the_real_plan(tmp_path):
    """Leg E in a fresh process (the jobs are forked off a process that never touches the GPU): two jobs, each with its own plan."""
    import json
    import subprocess

    import plumbing as P

    P.write_corpus(str(tmp_path / "wav"), n_files=4, seed=2)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "plumbing.py"), "--leg", "E", "--wav-dir", str(tmp_path / "wav"), "--repeat", "10", "--jobs", "2"],
                       capture_output=True, text=True, timeout=600)
    rows = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(rows) == 1, p.stderr[-1500:]
    assert rows[0]["cuts"] == 40 and rows[0]["errors"] is None and rows[0]["cuts_per_s"] > 0 and rows[0]["extractor"] == "hip"
