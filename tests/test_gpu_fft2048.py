"""GPU: fft2048c_kernel (44.1 / 48 kHz log-mel filterbanks, librosa-style log-mel with n_fft 2048) against the float64 oracle and
against the two older kernels that take the same configurations (wave-per-frame, generic)."""
import numpy as np
import pytest
import torch

from _golden import err_stats
from _golden import ref32
from _hip import make_hip
from oracle.kaldi_ref import RefConfig, RefExtractor

pytestmark = pytest.mark.gpu


def _waves(sr, seed):
    rs = np.random.RandomState(seed)
    n0 = int(0.025 * sr)
    hop = int(0.01 * sr)
    lens = [3 * sr + 17, n0 + 5, 130 * hop, 130 * hop + hop // 2, 130 * hop - 1, sr // 2, 10 * sr, 257 * hop + 3]
    amps = [1.0, 1.0, 0.3, 0.9, 0.01, 1.0, 0.5, 1e-3]
    ws = [((rs.rand(n).astype(np.float32) - 0.5) * a) for n, a in zip(lens, amps)]
    ws[6][: 3 * sr] += (0.4 * np.sin(2 * np.pi * 997.0 / sr * np.arange(3 * sr))).astype(np.float32)  # a tone on top of noise
    return ws


@pytest.mark.parametrize("sr,cfg,kernel", [
    (48000, {}, "fft2048c_kernel<19,0>"),
    (44100, {}, "fft2048c_kernel<18,1>"),            # odd hop (441 samples): unaligned pair reads for the second frame of a wave
    (48000, {"num_filters": 128}, "fft2048c_kernel<19,0>"),
    (44100, {"num_filters": 64, "low_freq": 50.0, "high_freq": 16000.0, "preemph_coeff": 0.0, "remove_dc_offset": False, "window_type": "hanning"}, "fft2048c_kernel<18,1>"),
    (48000, {"snip_edges": True, "frame_length": 0.04}, "fft2048c_kernel<32,0>"),  # 1920-sample frames
])
def test_ragged_batches_against_the_oracle_and_the_older_kernels(sr, cfg, kernel, monkeypatch):
    full = dict(cfg, sampling_rate=sr)
    ex = make_hip("fbank", full)
    assert ex.kernel_name.startswith(kernel), ex.kernel_name
    monkeypatch.setenv("HIPFEAT_NO_WAVE_AUTONOMOUS", "1")
    wv = make_hip("fbank", full)
    assert "wave_kernel<16>" in wv.kernel_name, wv.kernel_name
    monkeypatch.setenv("HIPFEAT_FORCE_GENERIC", "1")
    gen = make_hip("fbank", full)
    assert "generic" in gen.kernel_name
    monkeypatch.delenv("HIPFEAT_FORCE_GENERIC")
    monkeypatch.delenv("HIPFEAT_NO_WAVE_AUTONOMOUS")
    ws = _waves(sr, 48 + len(cfg))
    if cfg.get("snip_edges"):
        ws = [w for w in ws if len(w) >= int(0.04 * sr)]
    o32, o64 = ref32(RefConfig(kind="fbank", **full)), RefExtractor(RefConfig(kind="fbank", **full), np.float64)
    outs = ex.extract_batch([torch.from_numpy(w) for w in ws], sr)
    for w, o, a, b in zip(ws, outs, wv.extract_batch(ws, sr), gen.extract_batch(ws, sr)):
        got = o.cpu().numpy()
        want, truth = o32.extract(w), o64.extract(w)
        assert got.shape == want.shape == a.shape == b.shape
        s = err_stats(got, want)
        assert s["rel_l2"] <= 1e-4 and s["max_abs"] <= max(2e-3, 3 * err_stats(want, truth)["max_abs"]), (len(w), s)
        assert np.abs(got - a).max() <= 2e-3 and np.abs(got - b).max() <= 2e-3, (len(w), np.abs(got - a).max(), np.abs(got - b).max())
        assert np.array_equal(ex.extract(w, sr), got)  # batch == per cut, bit for bit


def test_filterbanks_outside_the_schedule_stay_on_the_wave_kernel():
    ex = make_hip("fbank", {"sampling_rate": 48000, "num_filters": 40})  # groups of four filters up to ~460 bins wide
    assert "wave_kernel<16>" in ex.kernel_name, ex.kernel_name


def test_long_uniform_batch_and_determinism():
    sr = 48000
    ex = make_hip("fbank", {"sampling_rate": sr})
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(64, 10 * sr, device="cuda", generator=g) - 0.5
    a = ex.extract_batch(x, sr)
    b = ex.extract_batch(x, sr)
    assert a.shape == (64, 1000, 80) and torch.equal(a, b) and bool(torch.isfinite(a).all())
    one = ex.extract_batch(x[17:18], sr)
    assert torch.equal(one.reshape(1000, 80), a[17])
    o64 = RefExtractor(RefConfig(kind="fbank", sampling_rate=sr), np.float64)
    for i in (0, 63):
        truth = o64.extract(x[i].cpu().numpy())
        assert err_stats(a[i].cpu().numpy(), truth)["rel_l2"] <= 1e-4
