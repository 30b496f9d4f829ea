"""CPU: host side of the speed-perturb / resample path (constants, lengths through the C ABI, transform API)."""
import os

import numpy as np
import pytest

from lhotse_amd import _lib, augmentation as A, constants
from oracle import resample_ref as R
from oracle.make_golden_resample import CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_product_filter_bank_is_bit_identical_to_the_reference(case):
    name, mode, a, b, _ = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    o, n = (round(a * b), a) if mode == "speed" else (a, b)
    k, width, orig, new = constants.sinc_resample_kernel(o, n)
    assert width == int(z["width"]) and k.dtype == np.float32
    assert np.array_equal(k, z["kernel"])  # bit for bit: the values the reference's conv1d multiplies by


def test_filter_bank_rejects_bad_arguments():
    with pytest.raises(ValueError):
        constants.sinc_resample_kernel(0, 16000)
    with pytest.raises(ValueError):
        constants.sinc_resample_kernel(16000, 8000, lowpass_filter_width=0)


def test_resampled_length_in_the_library_follows_the_reference_float32_rounding():
    lib = _lib.load()
    rng = np.random.RandomState(0)
    lens = list(range(0, 300)) + [16000, 160000, 479999, 12345, 16777217, 33554435] + rng.randint(1, 1 << 25, size=2000).tolist()
    for orig, new in [(9, 10), (11, 10), (19, 20), (2, 1), (1, 2), (441, 160), (160, 441)]:
        for L in lens:
            assert lib.raw("hipfeat_resampled_length", int(L), orig, new) == R.resampled_length(int(L), orig, new), (L, orig, new)
    assert lib.raw("hipfeat_resampled_length", 100, 0, 5) == 0


def test_vectorised_output_lengths_equal_the_library(monkeypatch):
    rng = np.random.RandomState(1)
    lens = np.concatenate([np.arange(0, 300), rng.randint(1, 1 << 25, size=3000)]).astype(np.int64)
    lib = _lib.load()
    for orig, new in [(9, 10), (11, 10), (441, 160), (1, 2)]:
        r = A.HipResampleTensor.__new__(A.HipResampleTensor)  # lengths need no device
        r.orig, r.new, r.lib, r.handle = orig, new, lib, 0
        want = np.array([lib.raw("hipfeat_resampled_length", int(n), orig, new) for n in lens])
        assert np.array_equal(r.output_lengths(lens), want)
        assert r.output_length(12345) == lib.raw("hipfeat_resampled_length", 12345, orig, new)


def test_transform_dict_round_trip_and_registry():
    sp = A.HipSpeed(factor=1.1)
    d = sp.to_dict()
    assert d == {"name": "HipSpeed", "kwargs": {"factor": 1.1, "device": "cuda"}}
    assert A.AudioTransform.from_dict(d) == sp
    rs = A.HipResample("44100", 16000)
    assert rs.source_sampling_rate == 44100
    assert A.AudioTransform.from_dict(rs.to_dict()) == rs


def test_reverse_timestamps_follow_the_reference_formulas():
    # docstring example of AudioTransform (transform.py:22-24): speed 0.9 -> (5.0, 10.0)
    off, dur = A.HipSpeed(0.9).reverse_timestamps(offset=5.5555625, duration=11.1111250, sampling_rate=16000)
    assert abs(off - 5.0) < 1e-4 and abs(dur - 10.0) < 1e-4
    off, dur = A.HipSpeed(1.1).reverse_timestamps(offset=1.0, duration=None, sampling_rate=16000)
    assert dur is None and off == A.perturb_num_samples(16000, 1 / 1.1) / 16000
    assert A.HipResample(16000, 16000).reverse_timestamps(1.23456789, 2.0, 16000) == (1.23456789, 2.0)
    off, dur = A.HipResample(16000, 22050).reverse_timestamps(14.727256235827664, None, 22050)
    assert off == 235636 / 16000 and dur is None
    # perturb_num_samples rounding (utils.py:649-654)
    assert A.perturb_num_samples(16000, 1.1) == 14545 and A.perturb_num_samples(16000, 0.9) == 17778
    assert A.perturb_num_samples(3, 2.0) == 2 and A.perturb_num_samples(1, 0.4) == 2  # half up / half down


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipFeatError):
        A.HipSpeed(1.1)(np.zeros((1, 1000), dtype=np.float32), 16000)
    with pytest.raises(_lib.HipFeatError):
        A.HipResampleTensor(16000, 8000, device="cpu")


def test_same_rate_is_identity_without_a_device():
    x = np.arange(10, dtype=np.float32)[None]
    assert A.HipResample(16000, 16000)(x) is x
