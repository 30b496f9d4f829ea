"""CPU: the host-side constants the plan uploads are BIT-IDENTICAL to the reference's
nn.Parameters (fixtures written by oracle/make_golden.py from the reference itself)."""
import numpy as np
import pytest

from _golden import CASES, load_case
from lhotse_amd import constants as C

CASE_NAMES = [c["name"] for c in CASES]


@pytest.mark.parametrize("name", CASE_NAMES)
def test_constants_bit_exact(name):
    case, _, z = load_case(name)
    cfg = dict(sampling_rate=16000, frame_length=0.025, frame_shift=0.01, round_to_power_of_two=True, window_type="povey",
               low_freq=20.0, high_freq=-400.0, num_filters=23 if case["kind"] == "mfcc" else 80, norm_filters=False,
               torchaudio_compatible_mel_scale=True, num_ceps=13, cepstral_lifter=22)
    cfg.update({k: v for k, v in case["cfg"].items() if k in cfg})
    n, shift, fft = C.frame_sizes(cfg["sampling_rate"], cfg["frame_length"], cfg["frame_shift"], cfg["round_to_power_of_two"])
    assert fft == int(z["fft_length"])
    w = C.make_window(n, cfg["window_type"])
    assert w.dtype == np.float32 and np.array_equal(w, z["window"])
    if "fb" in z:
        if cfg["torchaudio_compatible_mel_scale"]:
            fb = C.make_kaldi_mel(cfg["num_filters"], fft, cfg["sampling_rate"], cfg["low_freq"], cfg["high_freq"])
        else:
            fb = C.make_htk_mel(cfg["num_filters"], fft, cfg["sampling_rate"], cfg["low_freq"], cfg["high_freq"], cfg["norm_filters"])
        assert fb.dtype == np.float32 and fb.shape == z["fb"].shape
        assert np.array_equal(fb, z["fb"]), np.abs(fb - z["fb"]).max()
    if "dct" in z:
        assert np.array_equal(C.make_dct(cfg["num_ceps"], cfg["num_filters"]), z["dct"])
        assert np.array_equal(C.make_lifter(cfg["num_ceps"], cfg["cepstral_lifter"]), z["lifter"])


def test_mel_band_structure_default():
    """477 non-zeros, <= 2 per bin, bin 0 and bins 244..256 empty -- what the banded GEMM relies on."""
    fb = C.make_kaldi_mel(80, 512, 16000, 20.0, -400.0)
    assert fb.shape == (257, 80)
    assert int((fb != 0).sum()) == 477
    assert int((fb != 0).sum(axis=1).max()) <= 2
    assert not fb[0].any() and not fb[244:].any()


def test_invalid_options():
    with pytest.raises(ValueError):
        C.make_window(400, "kaiser")
    with pytest.raises(ValueError):
        C.make_kaldi_mel(3, 512, 16000, 20.0, -400.0)
    with pytest.raises(ValueError):
        C.make_kaldi_mel(80, 512, 16000, 9000.0, -400.0)
    assert C.MEL_FLOOR == np.finfo(np.float32).eps
