"""CPU: bench.py's `--config bulk_save` workload (the offline path end to end) driven with the oracle-backed stand-in for the device
plan -- the loop, the save thread, the archive / manifest it leaves behind and its parity leg.  The timing itself is a GPU-box matter."""
import argparse
import gzip
import json
import os

import numpy as np
import pytest
import torch

from _dropin_support import make_cpu_plan


@pytest.mark.parametrize("stripes", [1, 3])
def test_bulk_save_workload_writes_what_the_extractor_computes(monkeypatch, stripes):
    import bench
    import lhotse_amd.extractors as E

    monkeypatch.setattr(E, "_Plan", make_cpu_plan())
    monkeypatch.setattr(bench, "SAMPLES_PER_CUT", 16000)   # 1 s cuts: the oracle-backed plan is slow
    monkeypatch.setattr(bench, "FRAMES_PER_CUT", 100)
    monkeypatch.setattr(bench, "PARITY_CUTS", 8)
    args = argparse.Namespace(cuts=2, no_host_fed=False, stripes=stripes)
    w = bench.BulkSave(torch.device("cuda", 0), 0, args)
    assert w.units == 120 and w.batches[0][0][0].duration == 1.0 and w.fragments_per_s > 0
    w.step()
    first = w.last_root
    w.clear()
    w.step()
    w._dropper.shutdown(wait=True)  # (finished runs are deleted on a helper thread, off the timed path)
    w._dropper = None
    assert not os.path.exists(first) and os.path.isdir(w.last_root)  # one run on disk at a time
    st = w.stats
    assert st["manifest_lines"] == 120 and st["archive_bytes"] == 120 * 100 * 80 * 4 and st["extract_s"] > 0 and st["save_s"] > 0 and st["manifest_s"] > 0
    with gzip.open(os.path.join(w.last_root, "cuts.jsonl.gz"), "rt") as f:
        lines = [json.loads(ln) for ln in f]
    d = lines[61]
    assert d["type"] == "MonoCut" and d["features"]["type"] == "hip-fbank" and d["features"]["storage_type"] == "hip_archive"
    assert d["recording"]["sources"][0]["source"].endswith(".flac") and d["supervisions"][0]["speaker"] == "spk61"
    assert d["custom"] == {"dataloading_info": {"rank": 0, "world_size": 1, "worker_id": None}}
    files = {ln["features"]["storage_path"] for ln in lines}
    assert len(files) == stripes and sum(os.path.getsize(f) for f in files) == 120 * 100 * 80 * 4
    par = w.parity(0)
    # stored == what the (numpy-oracle-backed) plan computed; ref32 of the parity leg is the torch restatement since round 5: rounding apart
    assert par["n"] == 8 and par["rel_l2_max"] < 1e-5
    # round 4's per-cut Python route writes the same lines (storage fields apart: its archive is one file)
    st4 = {}
    root4 = w._one_pass("float32", "hip_archive", st4, "per_cut")
    with gzip.open(os.path.join(root4, "cuts.jsonl.gz"), "rt") as f:
        lines4 = [json.loads(ln) for ln in f]
    for a, b in zip(lines, lines4):
        for k in ("storage_path", "storage_key"):
            a["features"].pop(k), b["features"].pop(k)
    assert lines == lines4
    w._drop(root4)
    # the half-precision archive variant of `extra` leaves binary16 rows
    st16 = {}
    root = w._one_pass("float32", "hip_archive_f16", st16)  # (int16 PCM is converted on the device: GPU box only)
    assert st16["archive_bytes"] == 120 * 100 * 80 * 2
    with gzip.open(os.path.join(root, "cuts.jsonl.gz"), "rt") as f:
        assert json.loads(f.readline())["features"]["storage_key"].endswith(":f16")
    w._drop(root)
    w.close()
