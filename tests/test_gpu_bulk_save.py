"""GPU: the device-facing half of the bulk save path (lhotse_amd/storage.py) on a real MI355X -- lhotse itself cannot be
carried to the GPU box, so the manifest half runs under the real lhotse in tests/test_lhotse_dropin.py (oracle-backed plan) and THIS half
runs here without lhotse objects: the loop of compute_and_store_features_batch (lhotse/cut/set.py:2365-2404) -- batches of host waveforms ->
`_batch_features_on_host` (the H2D / kernel / D2H pipeline, on-device binary16 for the half-precision archive) -> ONE `write_packed` per
batch on a background thread -> self-describing keys -> positioned reads -- must hand back, per cut, exactly what `extract` computes."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import storage as S

pytestmark = pytest.mark.gpu


def _batches(seed, nbatches):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(nbatches):
        lens = rs.randint(2000, 90000, size=rs.randint(1, 40))
        out.append([((rs.rand(int(n)) - 0.5) * rs.choice([1.0, 0.1, 1e-3])).astype(np.float32) for n in lens])
    return out


@pytest.mark.parametrize("kind,writer_cls,reader_cls", [
    ("fbank", S.HipArchiveWriter, S.HipArchiveReader),
    ("mfcc", S.HipArchiveWriter, S.HipArchiveReader),
    ("fbank", S.HipArchiveF16Writer, S.HipArchiveF16Reader),
])
def test_driver_loop_without_lhotse_objects(tmp_path, kind, writer_cls, reader_cls):
    ex = (LA.HipFbank if kind == "fbank" else LA.HipMfcc)()
    half = getattr(writer_cls, "np_dtype", "<f4") == "<f2"
    batches = _batches(3, 12)
    keys, futures = [], []
    with writer_cls(tmp_path / "feats", mode="w") as writer, ThreadPoolExecutor(max_workers=1) as saver:
        for waves in batches:  # main thread extracts, one background thread appends: the structure of the reference driver
            host, frames = S._batch_features_on_host(ex, [torch.from_numpy(w) for w in waves], 16000, None, half=half)
            assert host.dtype == (np.float16 if half else np.float32) and host.shape[0] == sum(frames)
            futures.append(saver.submit(writer.write_packed, host, frames))
        for f in futures:
            keys.extend(f.result())
        path = writer.storage_path
    assert path.endswith(".hfa")
    reader = reader_cls(path)
    flat = [w for waves in batches for w in waves]
    assert len(keys) == len(flat)
    total = 0
    for key, w in zip(keys, flat):
        want = ex.extract(w, 16000)
        got = reader.read(key)
        assert got.dtype == np.float32 and got.shape == want.shape
        if half:
            assert key.endswith(":f16") and np.array_equal(got, want.astype(np.float16).astype(np.float32))  # the device's rounding = numpy's
        else:
            assert np.array_equal(got, want)
        if want.shape[0] > 30:  # partial reads touch only their own rows
            assert np.array_equal(reader.read(key, left_offset_frames=7, right_offset_frames=29), got[7:29])
        total += want.size * (2 if half else 4)
    import os

    assert os.path.getsize(path) == total  # nothing but the rows: no index, no padding


def test_appending_resumes_behind_the_existing_rows(tmp_path):
    ex = LA.HipFbank()
    a, b = _batches(5, 2)
    with S.HipArchiveWriter(tmp_path / "run", mode="w") as w:
        host, frames = S._batch_features_on_host(ex, [torch.from_numpy(x) for x in a], 16000, None)
        ka = w.write_packed(host, frames)
    with S.HipArchiveWriter(tmp_path / "run", mode="a") as w:  # an interrupted run continues (overwrite=False in the driver)
        host, frames = S._batch_features_on_host(ex, [torch.from_numpy(x) for x in b], 16000, None)
        kb = w.write_packed(host, frames)
        path = w.storage_path
    r = S.HipArchiveReader(path)
    for key, x in zip(ka + kb, a + b):
        assert np.array_equal(r.read(key), ex.extract(x, 16000))
    assert int(kb[0].split(":")[0]) == sum(int(k.split(":")[1]) for k in ka) * 80 * 4


@pytest.mark.parametrize("kind,pcm,half,zero_pad", [("fbank", False, False, False), ("fbank", True, True, False), ("mfcc", False, True, False),
                                                    ("fbank", False, False, True), ("fbank", True, False, True)])
def test_native_host_pipeline_equals_the_python_pipeline_bit_for_bit(kind, pcm, half, zero_pad, monkeypatch):
    """Round 5: hipfeat_host_pipeline_* (packing threads + chunked H2D / launch / D2H inside the library, asynchronous) hands back,
    for every batch, exactly the matrix `_batch_features_on_host` (the Python pipeline) does -- float32 and int16 PCM inputs, float32
    and binary16 results, both edge rules -- also with several batches in flight and results released out of order."""
    cfg = {"edge_rule": "batch_zero_pad"} if zero_pad else {}
    if pcm or kind == "mfcc":  # the chunked form of the pipeline (several upload / launch / download groups per batch): read at pipeline creation
        monkeypatch.setenv("HIPFEAT_PIPE_CHUNKS", "4")
    ex = (LA.HipFbank(LA.HipFbankConfig(**cfg)) if kind == "fbank" else LA.HipMfcc(LA.HipMfccConfig(**cfg)))
    batches = _batches(11, 9) + [[(np.random.RandomState(1).rand(160000).astype(np.float32) - 0.5) for _ in range(60)]]  # + one 600 s batch
    if pcm:
        batches = [[(w * 32767).astype(np.int16) for w in waves] for waves in batches]
    pend = []
    for waves in batches:  # everything is submitted before anything is waited for (at most 10 results outstanding)
        items = [torch.from_numpy(w) for w in waves[::2]] + list(waves[1::2])  # tensors and numpy arrays mixed
        order = list(range(0, len(waves), 2)) + list(range(1, len(waves), 2))
        p, frames = S._batch_features_pending(ex, items, 16000, None, half=half)
        assert p.ticket is not None, "the native pipeline must be the path taken on a GPU"
        pend.append((p, frames, [waves[i] for i in order]))
    for p, frames, waves in reversed(pend):  # waited for and released in reverse order
        got = p.wait().copy()
        p.release()
        want, wframes = S._batch_features_on_host(ex, [torch.from_numpy(w) for w in waves], 16000, None, half=half)
        assert list(frames) == list(wframes) and got.dtype == want.dtype and got.shape == want.shape
        assert np.array_equal(got, want)
    # a released buffer is reused: steady state allocates nothing new
    pipe = ex._native_pipe()
    for _ in range(3):
        p, _ = S._batch_features_pending(ex, [torch.from_numpy(w) for w in batches[-1]], 16000, None, half=half)
        p.wait()
        p.release()
    with pytest.raises(ValueError):  # too short: worded by the library, raised at submit
        S._batch_features_pending(ex, [torch.zeros(100)], 16000, None)
    ex2 = ex.to("cuda:0")  # moving the extractor drops the pipeline with the plan
    assert ex2._native_pipe() is not pipe


def test_native_archive_and_lines_behind_the_native_pipeline(tmp_path):
    """The whole native save path on the GPU box: submit -> wait -> hipfeat_archive_append (3 stripes) -> hipfeat_manifest_lines; what the
    lines point at is, per cut, exactly what `extract` computes."""
    import json

    ex = LA.HipFbank()
    batches = _batches(21, 6)
    lines = []
    with S.NativeArchive(tmp_path / "feats", mode="w", stripes=3) as ar:
        for bi, waves in enumerate(batches):
            p, frames = S._batch_features_pending(ex, [torch.from_numpy(w) for w in waves], 16000, None)
            file_of, byte_off = ar.append(p.wait(), np.asarray(frames))
            p.release()
            heads = [f'{{"id": "b{bi}c{i}", "storage_path": "'.encode() for i in range(len(waves))]
            tails = [b'"}'] * len(waves)
            blob = ar.lines(heads, tails, np.asarray(frames), np.asarray(frames), file_of, byte_off, 80)
            lines += [json.loads(ln) for ln in blob.decode().splitlines()]
        paths = [str(q) for q in ar.paths]
    flat = [w for waves in batches for w in waves]
    assert len(lines) == len(flat) and {d["storage_path"] for d in lines} == set(paths)
    readers = {q: S.HipArchiveReader(q) for q in paths}
    for d, w in zip(lines, flat):
        assert np.array_equal(readers[d["storage_path"]].read(d["storage_key"]), ex.extract(w, 16000))


def test_moving_the_extractor_with_a_result_outstanding_defers_the_teardown():
    """PendingFeatures.wait() hands out a view of page-locked memory the LIBRARY owns; `extractor.to()` / a dropped plan used to
    destroy the pipeline -- and free that memory -- under a save thread that still held the array.  The teardown (pipeline, then the plan it
    borrows) now waits for the last outstanding batch to be released; the moved extractor works on a new plan meanwhile."""
    ex = LA.HipFbank()
    waves = _batches(31, 1)[0]
    p, frames = S._batch_features_pending(ex, [torch.from_numpy(w) for w in waves], 16000, None)
    assert p.ticket is not None
    arr = p.wait()  # (what the save thread would hold on to)
    old_pipe, old_plan = ex._native_pipe(), ex.plan
    ex.to("cuda:0")  # drops plan + pipeline
    assert old_pipe.handle != 0 and old_plan.handle != 0, "torn down under an outstanding result"
    want = np.concatenate([ex.extract(w, 16000) for w in waves])  # the moved extractor: new plan, new kernels launched, allocator busy
    assert ex.plan is not old_plan and ex._native_pipe() is not old_pipe
    assert np.array_equal(arr, want)  # the old buffer is still intact
    with pytest.raises(LA._lib.HipFeatError):
        old_pipe.submit([torch.from_numpy(waves[0])])  # a closing pipeline takes no new batches
    p.release()
    assert old_pipe.handle == 0 and old_plan.handle == 0  # now it went, pipeline first, then its plan


def test_snip_edges_cut_shorter_than_a_frame_is_an_empty_matrix_in_the_native_pipeline():
    """hipfeat_host_pipeline_submit refused T == 0 where build_descs (and the reference, layers.py:745-746) accept it."""
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = LA.HipFbank(LA.HipFbankConfig(snip_edges=True))
    rs = np.random.RandomState(2)
    waves = [rs.rand(16000).astype(np.float32) - 0.5, rs.rand(300).astype(np.float32) - 0.5, rs.rand(401).astype(np.float32) - 0.5]
    p, frames = S._batch_features_pending(ex, [torch.from_numpy(w) for w in waves], 16000, None)
    assert p.ticket is not None and list(frames) == [98, 0, 1]
    got = p.wait().copy()
    p.release()
    assert got.shape == (99, 80) and np.array_equal(got[:98], ex.extract(waves[0], 16000)) and np.array_equal(got[98:], ex.extract(waves[2], 16000))


def test_overwriting_with_fewer_stripes_removes_the_stale_ones(tmp_path):
    ex = LA.HipFbank()
    waves = _batches(8, 1)[0]
    host, frames = S._batch_features_on_host(ex, [torch.from_numpy(w) for w in waves], 16000, None)
    with S.NativeArchive(tmp_path / "feats", mode="w", stripes=4) as ar:
        ar.append(host, np.asarray(frames))
        assert len(ar.paths) == 4
    assert sorted(p.name for p in tmp_path.iterdir()) == ["feats.1.hfa", "feats.2.hfa", "feats.3.hfa", "feats.hfa"]
    with S.NativeArchive(tmp_path / "feats", mode="w", stripes=2) as ar:
        ar.append(host, np.asarray(frames))
    assert sorted(p.name for p in tmp_path.iterdir()) == ["feats.1.hfa", "feats.hfa"]
