"""CPU: the kaldifeat-shaped option surface -- dict layout, registry, option mapping."""
import pickle

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import compat


def test_config_dict_layout_matches_kaldifeat_wrapper():
    d = LA.HipKaldifeatFbankConfig().to_dict()
    assert d["frame_opts"] == {
        "dither": 0.0, "preemph_coeff": 0.97, "remove_dc_offset": True, "window_type": "povey", "round_to_power_of_two": True,
        "blackman_coeff": 0.42, "snip_edges": False, "samp_freq": 16000.0, "frame_shift_ms": 10.0, "frame_length_ms": 25.0,
    }
    assert d["mel_opts"] == {"num_bins": 80, "low_freq": 20.0, "high_freq": -400.0, "vtln_low": 100.0, "vtln_high": -500.0,
                             "debug_mel": False, "htk_mode": False}
    assert d["use_energy"] is False and d["use_log_fbank"] is True and d["use_power"] is True and d["chunk_size"] == 120000
    back = LA.HipKaldifeatFbankConfig.from_dict(d)
    assert back == LA.HipKaldifeatFbankConfig()
    m = LA.HipKaldifeatMfccConfig()
    assert m.mel_opts.num_bins == 23 and m.num_ceps == 13 and m.cepstral_lifter == 22.0
    assert LA.HipKaldifeatMfccConfig.from_dict(m.to_dict()) == m


def test_registry_and_feature_extractor_round_trip(tmp_path):
    ex = LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(mel_opts=LA.HipKaldifeatMelOptions(num_bins=40)))
    assert compat.get_extractor_type("hip-kaldifeat-fbank") is LA.HipKaldifeatFbank
    assert ex.feature_dim(16000) == 40 and ex.frame_shift == 0.01 and ex.device == "cuda"
    p = tmp_path / "ex.yml"
    ex.to_yaml(p)
    again = compat.FeatureExtractor.from_yaml(p)
    assert isinstance(again, LA.HipKaldifeatFbank) and again.config == ex.config
    mf = pickle.loads(pickle.dumps(LA.HipKaldifeatMfcc()))
    assert mf.feature_dim(16000) == 13 and mf._inner is None


def test_option_mapping_onto_the_plan_config(monkeypatch):
    import lhotse_amd.extractors as E

    fo = LA.HipKaldifeatFrameOptions(sampling_rate=8000, frame_length=0.032, frame_shift=0.016, window_type="blackman", blackman_coeff=0.4,
                                     snip_edges=True, preemph_coeff=0.9, remove_dc_offset=False)
    ex = LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(frame_opts=fo, mel_opts=LA.HipKaldifeatMelOptions(num_bins=23, low_freq=60, high_freq=3800),
                                                         use_energy=True, use_power=False, energy_floor=1e-3))
    c = ex.inner.config
    assert (c.sampling_rate, c.frame_length, c.frame_shift, c.window_type, c.snip_edges, c.preemph_coeff, c.remove_dc_offset) == (
        8000, 0.032, 0.016, "blackman", True, 0.9, False)
    assert (c.num_filters, c.low_freq, c.high_freq, c.use_energy, c.use_fft_mag, c.energy_floor, c.edge_rule) == (23, 60, 3800, True, True, 1e-3, "reflect")
    assert c.blackman_coeff == 0.4
    m = LA.HipKaldifeatMfcc(LA.HipKaldifeatMfccConfig(num_ceps=20, cepstral_lifter=0.0, mel_opts=LA.HipKaldifeatMelOptions(num_bins=30))).inner.config
    assert (m.num_ceps, m.cepstral_lifter, m.num_filters) == (20, 0.0, 30)


@pytest.mark.parametrize("bad", [dict(mel_opts=LA.HipKaldifeatMelOptions(htk_mode=True)), dict(mel_opts=LA.HipKaldifeatMelOptions(debug_mel=True))])
def test_unsupported_options_fail_loudly(bad):
    with pytest.raises(NotImplementedError):
        LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(**bad)).inner
    assert LA.HipKaldifeatMfcc(LA.HipKaldifeatMfccConfig(use_energy=True)).inner.config.use_energy is True  # Kaldi: energy replaces C0


def test_sampling_rate_mismatch_and_no_cpu_fallback():
    import torch

    ex = LA.HipKaldifeatFbank()
    with pytest.raises(AssertionError, match="Mismatched sampling rate"):
        ex.extract(np.zeros(1000, dtype=np.float32), 8000)
    if not torch.cuda.is_available():
        from lhotse_amd import _lib

        with pytest.raises(_lib.HipFeatError):
            ex.extract(np.zeros(16000, dtype=np.float32), 16000)


def test_htk_compat_and_linear_fbank_are_column_operations_on_the_kernel_output():
    """Kaldi's feature-fbank.cc / feature-mfcc.cc: htk_compat moves the energy / C0 column last (C0 x sqrt(2) when it is a
    cepstral coefficient), use_log_fbank=False leaves the mel energies linear.  Checked on the hook itself (no GPU needed)."""
    t = torch.arange(12, dtype=torch.float32).reshape(2, 6) / 10.0
    fb = LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(use_energy=True, htk_compat=True))
    assert torch.equal(fb._post(t), torch.cat([t[:, 1:], t[:, :1]], dim=1))
    lin = LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(use_energy=True, use_log_fbank=False))._post(t)
    assert torch.equal(lin[:, :1], t[:, :1]) and torch.allclose(lin[:, 1:], torch.exp(t[:, 1:]))  # the log-energy column stays a log
    assert torch.allclose(LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(use_log_fbank=False))._post(t), torch.exp(t))
    assert LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(htk_compat=True))._post(t) is t  # nothing to move without an energy column
    mf = LA.HipKaldifeatMfcc(LA.HipKaldifeatMfccConfig(htk_compat=True))
    assert torch.allclose(mf._post(t), torch.cat([t[:, 1:], t[:, :1] * 2.0 ** 0.5], dim=1))
    me = LA.HipKaldifeatMfcc(LA.HipKaldifeatMfccConfig(htk_compat=True, use_energy=True))
    assert torch.equal(me._post(t), torch.cat([t[:, 1:], t[:, :1]], dim=1))
