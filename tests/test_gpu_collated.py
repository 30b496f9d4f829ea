"""GPU: fused collation (hipfeat_extract_collated) and int16 PCM input (hipfeat_pcm16_to_float), SURVEY 8f row 2."""
import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import _lib
from lhotse_amd.compat import LOG_EPSILON

pytestmark = pytest.mark.gpu


def _waves(seed, lens):
    rng = np.random.RandomState(seed)
    return [(rng.rand(n).astype(np.float32) - 0.5) for n in lens]


@pytest.mark.parametrize("kind", ["fbank", "mfcc", "log-spectrogram", "fbank-8k"])
def test_collated_equals_extract_batch_plus_collate_matrices(kind):
    sr = 16000
    if kind == "fbank":
        ex = LA.HipFbank()
    elif kind == "mfcc":
        ex = LA.HipMfcc()
    elif kind == "log-spectrogram":
        ex = LA.HipLogSpectrogram()
    else:
        sr = 8000
        ex = LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, num_filters=40))  # generic kernel
    xs = _waves(0, [16000, 52345, 160000, 8000, 31999, 160000])
    want = ex.extract_batch([torch.from_numpy(x) for x in xs], sr)
    got, lens = ex.extract_collated(xs, sr)
    F = ex.feature_dim(sr)
    assert got.is_cuda and got.dtype == torch.float32 and lens.dtype == torch.int64
    assert got.shape == (len(xs), max(len(w) for w in want), F)
    assert lens.tolist() == [len(w) for w in want]
    for i, w in enumerate(want):
        w = w if w.is_cuda else w.cuda()  # (log-)spectrogram extractors hand back CPU tensors
        assert torch.equal(got[i, : len(w)], w)
        assert torch.all(got[i, len(w) :] == np.float32(LOG_EPSILON))
    # custom padding value; torch inputs on the device; a single cut
    got0, _ = ex.extract_collated([torch.from_numpy(x).cuda() for x in xs], sr, padding_value=0.0)
    assert torch.equal(got0[2], got[2]) and torch.all(got0[3, int(lens[3]) :] == 0)
    one, l1 = ex.extract_collated([xs[1]], sr)
    assert one.shape == (1, int(lens[1]), F) and torch.equal(one[0], got[1, : int(lens[1])])


def test_collated_with_the_reference_batch_edge_rule():
    ex = LA.HipFbank(LA.HipFbankConfig(edge_rule="batch_zero_pad"))
    xs = _waves(1, [16000, 100000, 160000])
    want = ex.extract_batch([torch.from_numpy(x) for x in xs], 16000)
    got, lens = ex.extract_collated(xs, 16000)
    for i, w in enumerate(want):
        assert torch.equal(got[i, : len(w)], w)


def test_int16_pcm_input_is_bit_identical_to_the_float_path():
    rng = np.random.RandomState(3)
    pcm = [rng.randint(-32768, 32768, size=n).astype(np.int16) for n in (16000, 16001, 16007, 123457, 160000, 800, 5)]
    pcm = pcm[:-1]  # 5 samples is below the minimum length; keep odd sizes and tails
    flt = [p.astype(np.float32) / 32768.0 for p in pcm]  # what soundfile / the stdlib-wave backend produce
    ex = LA.HipFbank()
    a = ex.extract_batch(pcm, 16000)
    b = ex.extract_batch(flt, 16000)
    for u, v in zip(a, b):
        assert isinstance(u, np.ndarray) and np.array_equal(u, v)
    # single cut, torch int16 on host and on device, (1, T) shape
    assert np.array_equal(ex.extract(pcm[3][None], 16000), b[3])
    t = ex.extract(torch.from_numpy(pcm[3]).cuda(), 16000)
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), b[3])
    ca, la = ex.extract_collated(pcm, 16000)
    cb, lb = ex.extract_collated(flt, 16000)
    assert torch.equal(ca, cb) and torch.equal(la, lb)
    with pytest.raises(TypeError):
        ex.extract_batch([pcm[0], flt[1]], 16000)
    with pytest.raises(TypeError):
        ex.extract(pcm[0].astype(np.int32), 16000)


def test_pcm16_conversion_kernel_all_values_and_alignments():
    lib = _lib.load()
    allv = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).cuda()
    for off in (0, 1, 3, 8):
        src = allv[off:]
        out = torch.empty(src.numel() + 5, device="cuda")[5 - (off % 4):][: src.numel()]  # misaligned outputs too
        lib.check("hipfeat_pcm16_to_float", src.data_ptr(), out.data_ptr(), src.numel(), None)
        assert torch.equal(out, src.float() / 32768.0)
    assert lib.raw("hipfeat_pcm16_to_float", 0, 0, 0, None) == 0
    assert lib.raw("hipfeat_pcm16_to_float", 0, 0, 10, None) == _lib.ERR_INVALID


def test_collated_c_abi_errors():
    ex = LA.HipFbank()
    lib, plan = ex.plan.lib, ex.plan
    x = torch.zeros(32000, device="cuda")
    offs, lens = np.array([0, 16000], dtype=np.int64), np.array([16000, 16000], dtype=np.int64)
    out = torch.empty(2, 100, 80, device="cuda")
    nf = np.zeros(2, dtype=np.int64)
    assert lib.raw("hipfeat_extract_collated", plan.handle, x.data_ptr(), _lib.addr(offs), _lib.addr(lens), None, 2, out.data_ptr(), 99, 0.0, _lib.addr(nf), None) == _lib.ERR_INVALID
    assert "rows per cut" in lib.last_error()
    assert lib.raw("hipfeat_extract_collated", 0, x.data_ptr(), _lib.addr(offs), _lib.addr(lens), None, 2, out.data_ptr(), 100, 0.0, None, None) == _lib.ERR_INVALID
    assert lib.raw("hipfeat_extract_collated", plan.handle, x.data_ptr(), _lib.addr(offs), _lib.addr(lens), None, 2, out.data_ptr(), 100, 0.0, _lib.addr(nf), None) == 0
    assert nf.tolist() == [100, 100]
    with pytest.raises(ValueError):
        ex.extract_collated([], 16000)
