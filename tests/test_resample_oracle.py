"""CPU: pins oracle/resample_ref.py (speed perturbation / sinc resampling) to goldens made by the reference."""
import os

import numpy as np
import pytest

from oracle import resample_ref as R
from oracle.make_golden_resample import CASES
from oracle.signals import crc, make_signal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_resample_oracle_matches_reference(case):
    name, mode, a, b, inputs = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    o, n = (round(a * b), a) if mode == "speed" else (a, b)
    k, width, orig, new = R.sinc_kernel(o, n)
    assert width == int(z["width"]) and k.shape == z["kernel"].shape
    np.testing.assert_allclose(k, z["kernel"], rtol=0, atol=2e-7)
    for i, (kind, num, seed) in enumerate(inputs):
        x = make_signal(kind, num, seed)
        assert crc(x) == int(z[f"crc{i}"])
        y = R.speed(x, a, b) if mode == "speed" else R.resample(x, a, b)
        want = z[f"out{i}"]
        assert y.shape == want.shape, (y.shape, want.shape)
        np.testing.assert_allclose(y, want, rtol=0, atol=5e-6)


def test_resampled_length_matches_reference_rounding():
    # resample.py:309 takes the ceil of a float32 value
    for L in [1, 9, 10, 16000, 160000, 479999, 12345]:
        for orig, new in [(9, 10), (11, 10), (2, 1), (1, 2), (441, 160)]:
            assert R.resampled_length(L, orig, new) == int(np.ceil(np.float32(new * L / orig)))
