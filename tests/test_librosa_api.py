"""CPU: host side of HipLibrosaFbank -- constants against the oracle, registry / YAML, argument checks."""
import numpy as np
import pytest

import lhotse_amd as LA
from lhotse_amd import compat, constants
from oracle import librosa_ref as L


@pytest.mark.parametrize("sr,fft,m,fmin,fmax", [(22050, 1024, 80, 80, 7600), (16000, 512, 40, 0, None), (24000, 1200, 100, 50, 11000)])
def test_band_limited_slaney_filterbank(sr, fft, m, fmin, fmax):
    mel = constants.make_slaney_mel(m, fft, sr, fmin, fmax)
    assert mel.shape == (fft // 2 + 1, m) and mel.dtype == np.float32 and mel.flags["C_CONTIGUOUS"]
    assert np.array_equal(mel.T, L.mel(sr, fft, m, fmin, fmax))


@pytest.mark.parametrize("window", ["hann", "hamming", "blackman", "boxcar"])
@pytest.mark.parametrize("win,fft", [(1024, 1024), (400, 512), (1000, 1200), (399, 512)])
def test_stft_window(window, win, fft):
    w = constants.make_stft_window(window, win, fft)
    assert w.shape == (fft,) and w.dtype == np.float32
    left = (fft - win) // 2
    assert np.all(w[:left] == 0) and np.all(w[left + win :] == 0)
    assert np.abs(w[left : left + win] - L.get_window(window, win)).max() < 1e-7
    from scipy.signal import get_window

    assert np.abs(w[left : left + win] - get_window(window, win, fftbins=True)).max() < 1e-7


def test_other_scipy_windows_are_accepted():
    from scipy.signal import get_window

    w = constants.make_stft_window("bartlett", 400, 512)
    assert np.allclose(w[56:456], get_window("bartlett", 400, fftbins=True))


def test_registry_yaml_and_surface(tmp_path):
    ex = LA.HipLibrosaFbank()
    assert compat.get_extractor_type("hip-librosa-fbank") is LA.HipLibrosaFbank
    assert ex.feature_dim(22050) == 80 and ex.frame_shift == 256 / 22050 and ex.device == "cuda"
    d = ex.to_dict()
    assert d["feature_type"] == "hip-librosa-fbank" and d["fft_size"] == 1024 and d["hop_size"] == 256 and d["fmin"] == 80 and d["fmax"] == 7600
    assert "win_length" not in d  # None fields are dropped, as in the reference's config
    p = tmp_path / "l.yml"
    LA.HipLibrosaFbank(LA.HipLibrosaFbankConfig(sampling_rate=16000, fft_size=512, hop_size=160, win_length=400, num_mel_bins=40)).to_yaml(p)
    again = compat.FeatureExtractor.from_yaml(p)
    assert isinstance(again, LA.HipLibrosaFbank) and again.config.win_length == 400 and again.config.num_mel_bins == 40
    with pytest.raises(AssertionError, match="sampling_rate"):
        ex.extract(np.zeros(22050, dtype=np.float32), 8000)
    with pytest.raises(AssertionError, match="single-channel"):
        ex.extract(np.zeros((2, 22050), dtype=np.float32), 22050)
    a, b = np.log(np.full((3, 80), 2.0)), np.log(np.full((3, 80), 3.0))
    assert np.allclose(ex.mix(a, b, 2.0), np.log(8.0)) and np.allclose(ex.scale(a, 2.0), np.log(4.0))


def test_bad_win_length():
    ex = LA.HipLibrosaFbank(LA.HipLibrosaFbankConfig(win_length=2048))
    with pytest.raises(ValueError, match="win_length"):
        ex._plan_config()
