"""CPU: the host-side mirror of lhotse's extractor interface -- configs, registry, YAML,
pickling, statics -- behaves like the reference's (lhotse/features/base.py:37-405,
lhotse/features/kaldi/extractors.py).  No GPU call is made."""
import pickle
import warnings

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import compat
from lhotse_amd._lib import HipFeatError

ALL = [
    (LA.HipFbank, LA.HipFbankConfig, "hip-fbank", 80),
    (LA.HipMfcc, LA.HipMfccConfig, "hip-mfcc", 13),
    (LA.HipSpectrogram, LA.HipSpectrogramConfig, "hip-spectrogram", 257),
    (LA.HipLogSpectrogram, LA.HipLogSpectrogramConfig, "hip-log-spectrogram", 257),
]

# field sets of the reference configs (extractors.py:23-44, 155-178, 265-280)
REF_FBANK_FIELDS = dict(sampling_rate=16000, frame_length=0.025, frame_shift=0.01, round_to_power_of_two=True, remove_dc_offset=True,
                        preemph_coeff=0.97, window_type="povey", dither=0.0, snip_edges=False, energy_floor=1e-10, raw_energy=True,
                        use_energy=False, use_fft_mag=False, low_freq=20.0, high_freq=-400.0, num_filters=80, norm_filters=False,
                        torchaudio_compatible_mel_scale=True, device="cpu")


@pytest.mark.parametrize("cls,ccls,name,dim", ALL)
def test_default_construction_registry_and_dims(cls, ccls, name, dim):
    ex = cls()  # must not need a GPU (lhotse/features/base.py:381-388)
    assert isinstance(ex, compat.FeatureExtractor)
    assert ex.name == name and compat.FEATURE_EXTRACTORS[name] is cls
    assert isinstance(ex.config, ccls)
    assert ex.frame_shift == 0.01
    assert ex.feature_dim(16000) == dim
    assert ex.device == "cuda"
    d = ex.to_dict()
    assert d["feature_type"] == name and "num_mel_bins" not in d
    ex2 = compat.FeatureExtractor.from_dict(dict(d))
    assert type(ex2) is cls and ex2.config == ex.config
    assert pickle.loads(pickle.dumps(ex)).config == ex.config


def test_config_is_a_superset_of_the_reference_config():
    cfg = LA.HipFbankConfig.from_dict(dict(REF_FBANK_FIELDS))  # a kaldi-fbank YAML/dict loads unchanged
    for k, v in REF_FBANK_FIELDS.items():
        assert getattr(cfg, k) == v
    assert cfg.edge_rule == "reflect"
    # num_mel_bins alias (extractors.py:46-51)
    assert LA.HipFbankConfig(num_mel_bins=40).num_filters == 40
    assert LA.HipMfccConfig(num_mel_bins=40).num_filters == 40
    assert LA.HipMfccConfig().num_filters == 23 and LA.HipMfccConfig().num_ceps == 13
    with pytest.warns(UserWarning):
        LA.HipFbankConfig(snip_edges=True)
    with pytest.raises(ValueError):
        LA.HipFbankConfig(edge_rule="nope")


def test_yaml_round_trip(tmp_path):
    ex = LA.HipMfcc(LA.HipMfccConfig(num_filters=40, num_ceps=40, sampling_rate=8000, device="cuda:0"))
    p = tmp_path / "mfcc.yml"
    ex.to_yaml(p)
    back = compat.FeatureExtractor.from_yaml(p)
    assert type(back) is LA.HipMfcc and back.config == ex.config
    ex.config.device = torch.device("cuda", 1)
    ex.to_yaml(p)  # torch.device is stored as its type (lhotse/features/base.py:357-364)
    assert compat.FeatureExtractor.from_yaml(p).config.device == "cuda"


def test_mix_statics_match_reference_formulas():
    rs = np.random.RandomState(0)
    a, b = rs.randn(10, 80).astype(np.float32), rs.randn(10, 80).astype(np.float32)
    # extractors.py:134-152
    np.testing.assert_allclose(LA.HipFbank.mix(a, b, 0.5), np.log(np.maximum(1e-10, np.exp(a) + 0.5 * np.exp(b))))
    assert LA.HipFbank.compute_energy(a) == pytest.approx(float(np.sum(np.exp(a))))
    np.testing.assert_allclose(LA.HipFbank.scale(a, 2.0), a + np.log(2.0))
    # extractors.py:360-372
    np.testing.assert_allclose(LA.HipSpectrogram.mix(a, b, 0.5), a + 0.5 * b)
    assert LA.HipSpectrogram.compute_energy(a) == pytest.approx(float(a.sum()))
    np.testing.assert_allclose(LA.HipLogSpectrogram.scale(a, 2.0), 2.0 * a)
    with pytest.raises(ValueError):  # Mfcc defines no feature-domain mix (base.py:97-150)
        LA.HipMfcc.mix(a, b, 1.0)


def test_no_gpu_means_loud_failure_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ex = LA.HipFbank()
    with pytest.raises(HipFeatError, match="no CPU fallback"):
        ex.extract(np.zeros(16000, dtype=np.float32), 16000)
    with pytest.raises(HipFeatError):
        ex.extract_batch([np.zeros(16000, dtype=np.float32)], 16000)
    with pytest.raises(HipFeatError):
        LA.HipFbank(LA.HipFbankConfig(device="cpu")).extract(np.zeros(16000, dtype=np.float32), 16000)
    with pytest.raises(AssertionError):  # sampling-rate contract (extractors.py:95-100)
        ex.extract(np.zeros(16000, dtype=np.float32), 8000)


def test_to_device_and_frame_count_contract():
    ex = LA.HipFbank()
    ex.__dict__["_pipeline"] = object()  # a host pipeline built for the previous device (its streams live there)
    assert ex.to("cuda:1") is ex and ex.config.device == "cuda:1"
    assert "_pipeline" not in ex.__dict__ and ex._plan is None and ex._staging is None
    # lhotse/utils.py:424-434 -- validate_features asserts this on every stored matrix
    for s in (140, 159, 160, 16000, 160000, 100050):
        assert compat.compute_num_frames_from_samples(s, 0.01, 16000) == (s + 80) // 160
