"""CPU: the restated driver loops of tests/_driver_loops.py (what the GPU suite drives the real plan through) against the reference-driver
goldens, with the oracle-backed stand-in for the device plan -- so that a failure of tests/test_gpu_reference_drivers.py on the GPU box
points at the device path and not at the harness.  No reference needed: corpus + goldens are data."""
import numpy as np
import pytest
import torch

from _driver_loops import _per_cut_job, batch_driver, expected_num_frames
from _golden import err_stats, load_driver_goldens
from oracle.driver_corpus import CORPUS, read_pcm16, read_wav, write_corpus


@pytest.fixture()
def cpu_plan(monkeypatch):
    import lhotse_amd.extractors as E
    from _dropin_support import make_cpu_plan

    monkeypatch.setattr(E, "_Plan", make_cpu_plan())


def test_corpus_is_reproducible_and_matches_the_goldens(tmp_path):
    _, meta = load_driver_goldens()
    files = write_corpus(tmp_path)
    assert [list(c) for c in CORPUS] == meta["corpus"]
    for f, g in zip(files, meta["files"]):
        assert (f["id"], f["num_samples"], f["crc"]) == (g["id"], g["num_samples"], g["crc"])
        x = read_wav(f["path"])
        assert x.shape == (1, f["num_samples"]) and x.dtype == np.float32
        assert np.array_equal(x[0], read_pcm16(f["path"]).astype(np.float32) / 32768.0)
    for g in meta["per_cut"]:
        assert g["num_frames"] == expected_num_frames(g["duration"], g["frame_shift"], g["sampling_rate"])


@pytest.mark.parametrize("collate", [False, True])
def test_batch_loop_reproduces_the_reference_driver(tmp_path, cpu_plan, collate):
    import lhotse_amd as LA

    arrays, meta = load_driver_goldens()
    files = {f["id"]: f for f in write_corpus(tmp_path / "wav")}
    ex = LA.HipFbank(LA.HipFbankConfig(device="cpu", edge_rule="batch_zero_pad"))
    tag = f"batch_collate{int(collate)}"
    manifests = batch_driver(ex, files, meta["batches"], str(tmp_path / tag), collate=collate, num_workers=2)
    assert [m["id"] for m in manifests] == [g["id"] for g in meta[tag]]
    for m, g in zip(manifests, meta[tag]):
        assert all(m[k] == g[k] for k in ("num_frames", "num_features", "frame_shift", "sampling_rate", "start", "duration"))
        s = err_stats(np.load(tmp_path / tag / m["storage_key"]), arrays[f"{tag}/{m['id']}"])
        assert s["rel_l2"] <= 1e-4 and s["max_abs"] <= 2e-3, (m["id"], s)


def test_per_cut_job_and_the_fused_minibatch_entry(tmp_path, cpu_plan):
    import lhotse_amd as LA
    from lhotse_amd.input_strategies import FusedMiniBatch

    arrays, meta = load_driver_goldens()
    corpus = write_corpus(tmp_path / "wav")
    ex = LA.HipFbank(LA.HipFbankConfig(device="cpu"))
    for m in _per_cut_job(ex, corpus, str(tmp_path / "f"), 16000):
        s = err_stats(np.load(tmp_path / "f" / m["storage_key"]), arrays[f"per_cut/{m['id']}"])
        assert s["rel_l2"] <= 1e-4 and s["max_abs"] <= 2e-3
    files = {f["id"]: f for f in corpus}
    k2 = meta["k2_plain"]
    audios = [torch.from_numpy(read_wav(files[src]["path"])[0]) for src in k2["source_ids"]]
    exz = LA.HipFbank(LA.HipFbankConfig(device="cpu", edge_rule="batch_zero_pad"))
    feats, lens = FusedMiniBatch(exz).features_of(audios, k2["speed_factors"], k2["num_samples"], 16000)
    want = arrays["k2_plain/inputs"]
    assert tuple(feats.shape) == want.shape and [int(x) for x in lens] == [int(x) for x in arrays["k2_plain/num_frames"]]
    for i, t in enumerate(int(x) for x in lens):
        s = err_stats(feats[i, :t].numpy(), want[i, :t])
        assert s["rel_l2"] <= 1e-4 and s["max_abs"] <= 2e-3
        assert np.array_equal(feats[i, t:].numpy(), want[i, t:])
