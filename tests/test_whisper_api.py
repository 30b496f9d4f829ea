"""CPU: host side of HipWhisperFbank -- constants against the goldens, registry / YAML, C-ABI validation."""
import os

import numpy as np
import pytest

import lhotse_amd as LA
from lhotse_amd import compat, constants
from oracle import whisper_ref as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("n_mels,name", [(80, "whisper_80"), (128, "whisper_128")])
def test_slaney_filterbank_equals_the_one_the_goldens_were_made_with(n_mels, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    mel = constants.make_slaney_mel(n_mels, 400, 16000)
    assert mel.shape == (201, n_mels) and mel.dtype == np.float32 and mel.flags["C_CONTIGUOUS"]
    assert np.array_equal(mel.T, z["filters"])  # bit for bit (independent code, same published formula)


def test_periodic_hann_window_is_torch_hann_window():
    import torch

    w = constants.make_window(400, "hann_periodic")
    assert np.array_equal(w, torch.hann_window(400).numpy())
    assert np.abs(w - W.hann_periodic()).max() < 3e-7


def test_registry_yaml_and_surface(tmp_path):
    ex = LA.HipWhisperFbank()
    assert compat.get_extractor_type("hip-whisper-fbank") is LA.HipWhisperFbank
    assert ex.feature_dim(16000) == 80 and ex.frame_shift == 0.01 and ex.device == "cuda"
    assert ex.to_dict() == {"num_filters": 80, "device": "cuda", "feature_type": "hip-whisper-fbank"}
    p = tmp_path / "w.yml"
    LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=128)).to_yaml(p)
    again = compat.FeatureExtractor.from_yaml(p)
    assert isinstance(again, LA.HipWhisperFbank) and again.config.num_filters == 128
    with pytest.raises(AssertionError, match="sampling_rate"):
        ex.extract(np.zeros(16000, dtype=np.float32), 8000)
    with pytest.raises(ValueError, match="single-channel"):
        ex.extract(np.zeros((2, 16000), dtype=np.float32), 16000)
    a, b = np.log(np.full((3, 80), 2.0)), np.log(np.full((3, 80), 3.0))
    assert np.allclose(ex.mix(a, b, 2.0), np.log(8.0)) and np.allclose(ex.scale(a, 2.0), np.log(4.0))
