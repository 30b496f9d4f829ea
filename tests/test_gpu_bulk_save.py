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
