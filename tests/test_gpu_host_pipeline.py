"""GPU: the chunked H2D / kernel / D2H pipeline behind extract_batch on HOST inputs (lhotse_amd/extractors.py::_HostPipeline) -- what
CutSet.compute_and_store_features_batch (lhotse/cut/set.py:2393-2398) and OnTheFlyFeatures (dataset/input_strategies.py:441-443) call.
The pipeline must be invisible: every entry form (list of arrays, padded tensor + lengths; float32 / int16 PCM; page-locked or pageable)
gives, bit for bit, what ONE launch over the device-resident batch gives, in memory the caller owns."""
import threading

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import extractors as E

pytestmark = pytest.mark.gpu


@pytest.fixture()
def many_chunks(monkeypatch):
    """Small chunks, so that a test-sized batch crosses the pipeline in >= 4 pieces."""
    monkeypatch.setattr(E._HostPipeline, "MIN_CHUNK_BYTES", 64 << 10)
    monkeypatch.setattr(E._HostPipeline, "MAX_CHUNK_BYTES", 256 << 10)


def _waves(seed, n, lo=4000, hi=60000):
    rs = np.random.RandomState(seed)
    return [(rs.rand(int(k)).astype(np.float32) - 0.5) for k in rs.randint(lo, hi, size=n)]


def _device_answer(ex, waves):
    outs = ex.extract_batch([torch.from_numpy(w).cuda() for w in waves], 16000)
    return [o.cpu().numpy() for o in outs]


def test_chunk_bounds_cover_the_batch():
    nbytes = np.array([640000] * 60)
    b = E._HostPipeline.chunk_bounds(nbytes)
    assert b[0][0] == 0 and b[-1][1] == 60 and all(x[1] == y[0] for x, y in zip(b, b[1:])) and 3 <= len(b) <= 5
    b = E._HostPipeline.chunk_bounds(np.array([640000] * 1024))
    assert b[-1][1] == 1024 and all(x[1] == y[0] for x, y in zip(b, b[1:])) and max(y - x for x, y in b) * 640000 <= (48 << 20) + 640000
    assert E._HostPipeline.chunk_bounds(np.array([100])) == [(0, 1)]


@pytest.mark.parametrize("kind", ["fbank", "mfcc", "spectrogram"])
@pytest.mark.parametrize("edge_rule", ["reflect", "batch_zero_pad"])
def test_list_of_host_arrays_equals_one_device_launch(many_chunks, kind, edge_rule):
    cls, ccls = {"fbank": (LA.HipFbank, LA.HipFbankConfig), "mfcc": (LA.HipMfcc, LA.HipMfccConfig),
                 "spectrogram": (LA.HipSpectrogram, LA.HipSpectrogramConfig)}[kind]
    ex = cls(ccls(device="cuda:0", edge_rule=edge_rule))
    waves = _waves(1, 37)
    want = _device_answer(ex, waves)
    got = ex.extract_batch(waves, 16000)
    assert len(E._HostPipeline.chunk_bounds(np.array([4 * len(w) for w in waves]))) >= 4
    assert isinstance(got, list) and all(isinstance(g, np.ndarray) for g in got)
    for g, w in zip(got, want):
        assert g.shape == w.shape and np.array_equal(g, w)
    # int16 PCM items (half the bytes over PCIe) are the float path of x / 32768
    pcm = [np.round(w * 32767).astype(np.int16) for w in waves]
    want16 = _device_answer(ex, [p.astype(np.float32) / 32768.0 for p in pcm])
    for g, w in zip(ex.extract_batch(pcm, 16000), want16):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.int16])
def test_padded_tensor_with_lengths(many_chunks, pinned, dtype):
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0", edge_rule="batch_zero_pad"))
    waves = _waves(2, 29, 8000, 40000)
    lens = torch.tensor([len(w) for w in waves], dtype=torch.int32)
    x = torch.zeros(len(waves), int(lens.max()), dtype=dtype)
    for i, w in enumerate(waves):
        x[i, : len(w)] = torch.from_numpy(w if dtype == torch.float32 else np.round(w * 32767).astype(np.int16))
    ref_in = x if dtype == torch.float32 else x.to(torch.float32) / 32768.0
    want = ex.extract_batch(ref_in.cuda(), 16000, lengths=lens)  # device-resident: one launch
    if pinned:
        x = x.pin_memory()
    got = ex.extract_batch(x, 16000, lengths=lens)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert isinstance(g, np.ndarray) and np.array_equal(g, w)
    # equal lengths -> one stacked array (extractors.py:548-551 return convention)
    eq = torch.rand(9, 16000) - 0.5
    st = ex.extract_batch(eq.pin_memory() if pinned else eq, 16000, lengths=torch.full((9,), 16000))
    assert isinstance(st, np.ndarray) and st.shape == (9, 100, 80)


def test_results_are_the_callers_memory(many_chunks):
    """A slow consumer (lhotse's save thread) may hold a result over any number of later calls."""
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    first_in = _waves(3, 12)
    first = ex.extract_batch(first_in, 16000)
    snapshot = [f.copy() for f in first]
    for k in range(8):
        ex.extract_batch(_waves(10 + k, 12), 16000)
    for f, s in zip(first, snapshot):
        assert np.array_equal(f, s)
    assert all(np.array_equal(a, b) for a, b in zip(ex.extract_batch(first_in, 16000), snapshot))


def test_two_threads_share_one_extractor(many_chunks):
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    sets = [_waves(20 + t, 15) for t in range(2)]
    want = [_device_answer(ex, s) for s in sets]
    errors = []

    def work(t):
        try:
            for _ in range(6):
                got = ex.extract_batch(sets[t], 16000)
                assert all(np.array_equal(g, w) for g, w in zip(got, want[t]))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_too_short_item_raises_value_error_from_inside_the_pipeline(many_chunks):
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    waves = _waves(5, 20) + [np.zeros(100, dtype=np.float32)] + _waves(6, 5)
    with pytest.raises(ValueError):
        ex.extract_batch(waves, 16000)
    assert len(ex.extract_batch(_waves(7, 6), 16000)) == 6  # the extractor is usable afterwards


def test_float_to_half_on_the_device_and_the_half_pipeline(many_chunks):
    """hipfeat_float_to_half = IEEE round-to-nearest-even (numpy's astype(float16)), and the pipeline's `half` mode hands back exactly the
    binary16 rounding of what the float32 mode hands back."""
    from lhotse_amd import _lib

    lib = _lib.load()
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.randn(100003).astype(np.float32) * 10, np.float32([0.0, -0.0, 65504.0, 65520.0, 1e-8, 6e-5, -23.025851, 1e9, -1e9]),
                        (np.arange(4096, dtype=np.float32) + 0.5) / 1024.0])  # ties, subnormals, overflow to inf
    for off in (0, 1, 3):  # aligned and unaligned starts
        d = torch.from_numpy(x).cuda()[off:]
        o = torch.empty(d.numel(), dtype=torch.float16, device="cuda")
        lib.check("hipfeat_float_to_half", d.data_ptr(), o.data_ptr(), d.numel(), int(torch.cuda.current_stream().cuda_stream))
        with np.errstate(over="ignore"):
            want = x[off:].astype(np.float16)
        assert np.array_equal(o.cpu().numpy().view(np.uint16), want.view(np.uint16))
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    waves = _waves(9, 21)
    f32, frames = ex._host_items_to_host(waves, None)
    f16, frames16 = ex._host_items_to_host(waves, None, half=True)
    assert f16.dtype == torch.float16 and np.array_equal(frames, frames16)
    assert torch.equal(f16, f32.to(torch.float16))


def test_moving_the_extractor_rebuilds_the_pipeline(many_chunks):
    """`to()` dropped the plan but kept the cached pipeline, whose side streams (and events) belong to the device the
    extractor was on before -- compute_and_store_features_sharded does exactly `extractor.to(f"cuda:{LOCAL_RANK}")` after a warm-up.
    (One GPU per box: the move is to the same index under another spelling; what is checked is that the pipeline goes with the plan.)"""
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda"))
    waves = _waves(11, 24)
    first = [np.asarray(o) for o in ex.extract_batch(waves, 16000)]
    pipe0 = ex.__dict__["_pipeline"]
    ex.to("cuda:0")
    assert "_pipeline" not in ex.__dict__ and ex._plan is None
    again = [np.asarray(o) for o in ex.extract_batch(waves, 16000)]
    pipe1 = ex.__dict__["_pipeline"]
    assert pipe1 is not pipe0 and pipe1.device == ex.plan.device
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    # a pipeline left over for another device is never used (belt and braces behind _drop_plan)
    ex.__dict__["_pipeline"] = type("P", (), {"device": torch.device("cuda", 7)})()
    assert ex._pipe().device == ex.plan.device
