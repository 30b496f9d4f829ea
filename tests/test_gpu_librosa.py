"""GPU: HipLibrosaFbank (kind HIPFEAT_LIBROSA_FBANK: fft1024c_kernel for n_fft 1024 with an even hop, fft2048c_kernel for n_fft 2048, wave_kernel for the other power-of-two FFT sizes,
generic_kernel otherwise)
against goldens produced by the reference's LibrosaFbank.extract (librosa's stft / mel restated, see
oracle/librosa_ref.py) and against the oracle on seeded inputs.

Tolerance.  Output units are log10, the reference (librosa) runs its STFT in float64; the bar is 1e-4 absolute in
log10 units (2.3e-4 relative in the mel magnitude) for every bin whose magnitude is within 60 dB of the loudest bin of
its frame; below that the rounding noise of a float32 FFT (~1e-7 of the frame's peak) is no longer 1e-4 of the bin, so
the error there is bounded in the linear domain by 1e-6 of the frame's peak instead.  Broadband inputs never touch the
second clause, and the tests assert that."""
import os

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from oracle import librosa_ref as L
from oracle.make_golden_librosa import CASES
from oracle.signals import crc, make_signal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(got, want, ctx, broadband=True):
    assert got.shape == want.shape and got.dtype == np.float32, ctx
    err = np.abs(got.astype(np.float64) - want)
    if broadband:
        assert err.max() <= 1e-4, (ctx, err.max())
        return
    lin_g, lin_w = 10.0 ** got.astype(np.float64), 10.0 ** want.astype(np.float64)
    peak = lin_w.max(axis=1, keepdims=True)
    ok = (err <= 1e-4) | (np.abs(lin_g - lin_w) <= 1e-6 * peak)
    assert ok.all(), (ctx, err.max(), np.argwhere(~ok)[:4])
    loud = lin_w >= 1e-3 * peak
    assert err[loud].max() <= 1e-4, (ctx, err[loud].max())


def _cfg(over):
    return LA.HipLibrosaFbankConfig(**over)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_librosa_matches_reference_golden(case):
    name, over, inputs = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ex = LA.HipLibrosaFbank(_cfg(over))
    fft = over.get("fft_size", 1024)
    assert any(k in ex.kernel_name for k in ("wave_kernel", "fft1024c_kernel", "fft2048c_kernel")) == (fft & (fft - 1) == 0), ex.kernel_name
    for i, (kind, n, seed) in enumerate(inputs):
        x = make_signal(kind, n, seed)
        assert crc(x) == int(z[f"crc{i}"])
        got = ex.extract(x, ex.config.sampling_rate)
        assert isinstance(got, np.ndarray)
        _close(got, z[f"out{i}"], (name, i, kind), broadband=kind in ("uniform", "gauss", "speechlike"))
        if kind == "zeros":
            assert np.abs(got + 10.0).max() <= 5e-6  # log10(eps), librosa_fbank.py:127 (v_log_f32 times log10(2): a couple of ulps)


@pytest.mark.parametrize(
    "over,kernel",
    [
        ({}, "fft1024c_kernel<32>"),  # the librosa defaults (22.05 kHz, n_fft 1024, hop 256): wave-autonomous kernel
        ({"sampling_rate": 16000, "fft_size": 512, "hop_size": 128, "num_mel_bins": 64, "fmin": 20, "fmax": None}, "wave_kernel<4>"),
        ({"sampling_rate": 44100, "fft_size": 2048, "hop_size": 512, "win_length": 1764, "num_mel_bins": 128, "fmin": 0, "fmax": 16000}, "fft2048c_kernel<32,0>"),
        ({"sampling_rate": 16000, "fft_size": 400, "hop_size": 160, "num_mel_bins": 80, "fmin": 0, "fmax": 8000}, "generic"),
        ({"sampling_rate": 8000, "fft_size": 256, "hop_size": 80, "win_length": 200, "window": "blackman", "num_mel_bins": 23, "fmin": 100, "fmax": 3800}, "generic"),
        ({"sampling_rate": 22050, "fft_size": 1024, "hop_size": 275, "win_length": 1000, "window": "hamming"}, "wave_kernel<8>"),
    ],
)
def test_configs_and_ragged_batches_against_the_oracle(over, kernel):
    ex = LA.HipLibrosaFbank(_cfg(over))
    assert kernel in ex.kernel_name, ex.kernel_name
    sr, fft, hop = ex.config.sampling_rate, ex.config.fft_size, ex.config.hop_size
    rng = np.random.RandomState(fft + hop)
    lens = [sr, fft // 2 + 1, fft, 3 * sr + 17, hop * 40 + hop // 2 - 1, hop * 40 + hop // 2, 7777]
    xs = [(rng.rand(n).astype(np.float32) - 0.5) * s for n, s in zip(lens, [1.0, 1.0, 0.3, 0.9, 0.01, 1.0, 0.5])]
    okw = {k: v for k, v in over.items()}
    outs = ex.extract_batch([torch.from_numpy(x) for x in xs], sr)
    assert isinstance(outs, list) and all(o.is_cuda for o in outs)
    for x, o in zip(xs, outs):
        want = L.logmelfilterbank(x, **okw)
        assert o.shape[0] == (len(x) + hop // 2) // hop
        _close(o.cpu().numpy(), want, (kernel, len(x)))
        assert np.array_equal(ex.extract(x, sr), o.cpu().numpy())  # batch == per cut, bit for bit
    col, nfr = ex.extract_collated(xs, sr)
    assert nfr.tolist() == [len(o) for o in outs] and col.shape == (len(xs), max(nfr.tolist()), ex.config.num_mel_bins)
    for i, o in enumerate(outs):
        assert torch.equal(col[i, : len(o)], o) and torch.all(col[i, len(o) :] == np.float32(LA.compat.LOG_EPSILON))
    pcm = [np.round(x * 32767).astype(np.int16) for x in xs[:3]]
    for p, o in zip(pcm, ex.extract_batch(pcm, sr)):
        _close(o, L.logmelfilterbank(p.astype(np.float32) / 32768.0, **okw), (kernel, "pcm16"))


def test_the_three_kernels_agree(monkeypatch):
    """librosa defaults on the wave-autonomous fft1024 kernel, the wave-per-frame kernel and the generic kernel."""
    x = make_signal("speechlike", 66150, 3)
    fast = LA.HipLibrosaFbank()
    assert "fft1024c_kernel" in fast.kernel_name
    a = fast.extract(x, 22050)
    monkeypatch.setenv("HIPFEAT_NO_WAVE_AUTONOMOUS", "1")
    wv = LA.HipLibrosaFbank()
    assert "wave_kernel" in wv.kernel_name
    c = wv.extract(x, 22050)
    monkeypatch.setenv("HIPFEAT_NO_WAVE_KERNEL", "1")
    g = LA.HipLibrosaFbank()
    assert "generic" in g.kernel_name
    b = g.extract(x, 22050)
    monkeypatch.delenv("HIPFEAT_NO_WAVE_KERNEL")
    monkeypatch.delenv("HIPFEAT_NO_WAVE_AUTONOMOUS")
    want = L.logmelfilterbank(x)
    _close(a, want, "fft1024c")
    _close(c, want, "wave")
    _close(b, want, "generic")


def test_too_short_and_shapes():
    ex = LA.HipLibrosaFbank()
    with pytest.raises(ValueError, match="reflect padding"):
        ex.extract(np.zeros(512, dtype=np.float32), 22050)  # single reflection only (torch.stft refuses too; np.pad would reflect repeatedly)
    y = ex.extract(np.zeros((1, 22050), dtype=np.float32), 22050)
    assert y.shape == (86, 80)
    t = ex.extract(torch.zeros(22050), 22050)
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.shape == (86, 80)
    same = ex.extract_batch(np.zeros((3, 22050), dtype=np.float32), 22050)
    assert isinstance(same, np.ndarray) and same.shape == (3, 86, 80)
