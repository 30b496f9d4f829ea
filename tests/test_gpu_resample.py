"""GPU: speed perturbation / sinc resampling (resample_kernel through the C ABI) against the reference's goldens
and the oracle (oracle/resample_ref.py).  Tolerance: the kernel multiplies the reference's own float32 filter
bank (bit-identical, tests/test_resample_api.py); only the float32 summation order differs from the reference's
conv1d, so outputs must agree to max_abs <= 1e-5 for |x| <= 1 (north_star bar: 1e-4)."""
import os

import numpy as np
import pytest
import torch

from lhotse_amd import _lib, augmentation as A
from oracle import kaldi_ref as K
from oracle import resample_ref as R
from oracle.make_golden_resample import CASES
from oracle.signals import crc, make_signal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ABS_TOL = 1e-5


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_resampler_matches_reference_golden(case):
    name, mode, a, b, inputs = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fn = A.HipSpeed(b) if mode == "speed" else A.HipResample(a, b)
    for i, (kind, num, seed) in enumerate(inputs):
        x = make_signal(kind, num, seed)
        assert crc(x) == int(z[f"crc{i}"])
        y = fn(x[None], a)
        want = z[f"out{i}"]
        assert isinstance(y, np.ndarray) and y.dtype == np.float32 and y.shape == (1,) + want.shape
        assert np.abs(y[0] - want).max() <= ABS_TOL, (name, i, np.abs(y[0] - want).max())


@pytest.mark.parametrize("orig,new", [(17600, 16000), (14400, 16000), (15200, 16000), (8000, 16000), (48000, 16000), (16000, 22050)])
def test_ragged_device_batch_against_oracle_float64(orig, new):
    rng = np.random.RandomState(orig % 977)
    lens = [1, 2, 37, 160, 4095, 4096, 4097, 16000, 52345, 160000, 0, 23]
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in lens]
    r = A.get_or_create_resampler(orig, new)
    ys = r.resample_batch(xs)
    assert len(ys) == len(xs)
    for x, y in zip(xs, ys):
        assert y.is_cuda and y.dtype == torch.float32
        want = R.resample(x, orig, new, dtype=np.float64) if len(x) else np.zeros(0)
        assert y.numel() == len(want) == R.resampled_length(len(x), r.orig, r.new)
        if len(x):
            assert np.abs(y.cpu().numpy() - want).max() <= ABS_TOL


def test_tensor_call_shapes_devices_and_channels():
    r = A.get_or_create_resampler(17600, 16000)
    x = torch.from_numpy(np.stack([make_signal("gauss", 30000, s) for s in range(3)]))
    y = r(x)  # cpu in -> cpu out, (C, T) -> (C, T')
    assert not y.is_cuda and y.shape == (3, R.resampled_length(30000, 11, 10))
    for c in range(3):
        assert np.abs(y[c].numpy() - R.resample(x[c].numpy(), 17600, 16000, dtype=np.float64)).max() <= ABS_TOL
    yd = r(x.cuda().reshape(3, 1, 30000))
    assert yd.is_cuda and yd.shape == (3, 1, y.shape[1]) and torch.equal(yd.cpu().reshape(3, -1), y)
    y1 = r(x[0])
    assert y1.shape == (y.shape[1],) and torch.equal(y1, y[0])
    with pytest.raises(TypeError):
        r(x.double())
    assert A.get_or_create_resampler(17600, 16000) is r
    same = A.HipResampleTensor(16000, 16000)
    assert same(x) is x


def test_full_size_properties():
    """BASELINE-sized batch (3 x-speed variants of 10 s cuts): lengths, linearity, DC gain, shift structure."""
    n, L = 256, 160000
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.empty(n, L, device="cuda").uniform_(-0.5, 0.5, generator=g)
    b = torch.empty(n, L, device="cuda").uniform_(-0.5, 0.5, generator=g)
    for factor in (0.9, 1.1):
        r = A.get_or_create_resampler(round(16000 * factor), 16000)
        ya, yb = r(a), r(b)
        assert ya.shape == (n, R.resampled_length(L, r.orig, r.new))
        # linearity: R(2a - 3b) == 2 R(a) - 3 R(b) up to float32 rounding
        yl = r(2 * a - 3 * b)
        assert float((yl - (2 * ya - 3 * yb)).abs().max()) <= 2e-5
        # a constant signal comes out as the filter's DC gain (sum of each polyphase filter ~ 1) away from the edges
        c = r(torch.full((1, 20000), 0.25, device="cuda"))[0, 100:-100]
        gains = r.kernel.astype(np.float64).sum(axis=1)
        assert abs(float(c.max()) - 0.25 * gains.max()) <= 1e-6 and abs(float(c.min()) - 0.25 * gains.min()) <= 1e-6
        # delaying the input by `orig` samples delays the output by exactly `new` samples (polyphase structure)
        xs = torch.cat([torch.zeros(n, r.orig, device="cuda"), a[:, : L - r.orig]], dim=1)
        ys = r(xs)
        assert torch.equal(ys[:, r.new : 100000], ya[:, : 100000 - r.new])
        # spot-check rows against the float64 oracle
        for row in (0, n - 1):
            want = R.resample(a[row].cpu().numpy(), round(16000 * factor), 16000, dtype=np.float64)
            assert np.abs(ya[row].cpu().numpy() - want).max() <= ABS_TOL


def test_speed_then_fbank_stays_on_device_and_matches_oracle_pipeline():
    import lhotse_amd as LA

    rng = np.random.RandomState(5)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in (16000, 40123, 160000)]
    ex = LA.HipFbank()
    ref = K.RefExtractor(K.RefConfig(kind="fbank"), dtype=np.float64)
    for factor in (0.9, 1.1):
        sp = A.HipSpeed(factor)
        perturbed = [sp(torch.from_numpy(x).cuda(), 16000) for x in xs]
        assert all(p.is_cuda for p in perturbed)
        feats = ex.extract_batch(perturbed, 16000)
        for x, p, f in zip(xs, perturbed, feats):
            assert p.numel() == A.perturb_num_samples(len(x), factor) or abs(p.numel() - len(x) / factor) < 1.0
            want = ref.extract(R.speed(x, 16000, factor, dtype=np.float64))
            got = f.cpu().numpy() if isinstance(f, torch.Tensor) else f
            assert got.shape == want.shape
            assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4


def test_c_abi_errors_and_empty_batch():
    lib = _lib.load()
    r = A.get_or_create_resampler(17600, 16000)
    h = np.zeros(1, dtype=np.uint64)
    assert lib.raw("hipfeat_resampler_create", 0, 10, 7, _lib.addr(r.kernel), 0, _lib.addr(h)) == _lib.ERR_INVALID
    assert lib.raw("hipfeat_resampler_create", 11, 10, 7, None, 0, _lib.addr(h)) == _lib.ERR_INVALID
    assert lib.raw("hipfeat_resampler_create", 11, 10, 7, _lib.addr(r.kernel), 99, _lib.addr(h)) == _lib.ERR_HIP
    assert "not available" in lib.last_error()
    assert lib.raw("hipfeat_resample", 0, 0, None, None, 0, 0, None, None) == _lib.ERR_INVALID
    assert lib.raw("hipfeat_resample", r.handle, 0, None, None, 0, 0, None, None) == 0  # empty batch is a no-op
    offs, lens = np.zeros(1, dtype=np.int64), np.array([-1], dtype=np.int64)
    x = torch.zeros(16, device="cuda")
    assert lib.raw("hipfeat_resample", r.handle, x.data_ptr(), _lib.addr(offs), _lib.addr(lens), 1, x.data_ptr(), _lib.addr(offs), None) == _lib.ERR_INVALID
    assert lib.raw("hipfeat_resampler_destroy", 0) == 0
    assert r.resample_batch([]) == []


@pytest.mark.parametrize("orig,new", [(14400, 16000), (17600, 16000), (15200, 16000), (16800, 16000), (8000, 16000), (32000, 16000), (48000, 16000)])
def test_fast_and_generic_kernels_are_bit_identical(orig, new, monkeypatch):
    rng = np.random.RandomState(3)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in (1, 5, 255, 256, 257, 2559, 2560, 2561, 100003, 0, 7)]
    fast = A.HipResampleTensor(orig, new)
    monkeypatch.setenv("HIPFEAT_RESAMPLE_GENERIC", "1")
    generic = A.HipResampleTensor(orig, new)
    monkeypatch.delenv("HIPFEAT_RESAMPLE_GENERIC")
    for a, b in zip(fast.resample_batch(xs), generic.resample_batch(xs)):
        assert a.shape == b.shape and torch.equal(a, b)
