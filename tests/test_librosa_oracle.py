"""CPU: pins oracle/librosa_ref.py's lhotse-side steps to goldens produced by the reference's LibrosaFbank.extract, and
cross-checks the restated librosa.stft against torch.stft (librosa itself is not available offline)."""
import os

import numpy as np
import pytest
import torch

from oracle import librosa_ref as L
from oracle.make_golden_librosa import CASES
from oracle.signals import crc, make_signal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_librosa_oracle_matches_reference(case):
    name, over, inputs = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    for i, (kind, n, seed) in enumerate(inputs):
        x = make_signal(kind, n, seed)
        assert crc(x) == int(z[f"crc{i}"])
        want = z[f"out{i}"]
        got = L.logmelfilterbank(x, **over)
        hop = over.get("hop_size", 256)
        assert got.shape == want.shape == ((n + hop // 2) // hop, over.get("num_mel_bins", 80))
        assert np.abs(got - want).max() <= 1e-5, (name, i, np.abs(got - want).max())


@pytest.mark.parametrize("n_fft,hop,win,window", [(1024, 256, None, "hann"), (512, 160, 400, "hamming"), (1200, 300, 1000, "blackman"), (400, 160, None, "hann")])
def test_restated_stft_agrees_with_torch_stft(n_fft, hop, win, window):
    x = make_signal("gauss", 12345, 11)
    ours = L.stft(x.astype(np.float64), n_fft=n_fft, hop_length=hop, win_length=win, window=window)
    wl = n_fft if win is None else win
    tw = {"hann": torch.hann_window, "hamming": torch.hamming_window, "blackman": torch.blackman_window}[window](wl, periodic=True, dtype=torch.float64)
    ref = torch.stft(torch.from_numpy(x.astype(np.float64)), n_fft, hop, wl, window=tw, center=True, pad_mode="reflect", return_complex=True).numpy()
    assert ours.shape == ref.shape == (1 + n_fft // 2, 1 + len(x) // hop)
    assert np.abs(ours - ref).max() <= 2e-5 * np.abs(ref).max()


def test_mel_with_band_limits():
    f = L.mel(22050, 1024, 80, 80, 7600)
    assert f.shape == (80, 513) and f.dtype == np.float32 and f.min() >= 0
    freqs = np.fft.rfftfreq(1024, 1 / 22050)
    assert np.all(f[:, freqs <= 80] == 0) and np.all(f[:, freqs >= 7600] == 0)
    assert np.all(np.diff(f.argmax(axis=1)) >= 0)
    from oracle.whisper_ref import slaney_mel_filters

    assert np.array_equal(L.mel(16000, 400, 80), slaney_mel_filters(16000, 400, 80))  # the pinned-by-tripwire whisper filters


def test_rows():
    for n, hop in [(22050, 256), (22143, 256), (22144, 256), (513, 256), (1000, 160), (30011, 300)]:
        x = make_signal("uniform", n, 1)
        assert L.logmelfilterbank(x, hop_size=hop).shape[0] == (n + hop // 2) // hop
