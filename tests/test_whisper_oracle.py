"""CPU: pins oracle/whisper_ref.py to goldens produced by the reference's log_mel_spectrogram, and checks the
known properties of the restated slaney filterbank (librosa itself is not available offline)."""
import os

import numpy as np
import pytest

from oracle import whisper_ref as W
from oracle.make_golden_whisper import CASES
from oracle.signals import crc, make_signal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_whisper_oracle_matches_reference(case):
    name, n_mels, inputs = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    filters = W.slaney_mel_filters(16000, 400, n_mels)
    assert np.array_equal(filters, z["filters"])
    for i, (kind, n, seed) in enumerate(inputs):
        x = make_signal(kind, n, seed)
        assert crc(x) == int(z[f"crc{i}"])
        want = z[f"out{i}"]
        got = W.log_mel_spectrogram(x, filters)
        assert got.shape == want.shape == (W.num_rows(n), n_mels)
        # output scale is log10/4: 1e-4 here == 1e-3 in natural-log units
        assert np.abs(got - want).max() <= 2e-4, (name, i, np.abs(got - want).max())
        truth = W.log_mel_spectrogram(x, filters, dtype=np.float64)
        assert np.abs(want - truth).max() <= 2e-4


def test_rows_and_zero_padding_row():
    filters = W.slaney_mel_filters()
    x = make_signal("gauss", 16080, 4)  # 16080 // 160 = 100 computed frames, (16080 + 80) // 160 = 101 rows
    y = W.log_mel_spectrogram(x, filters)
    assert y.shape == (101, 80) and np.all(y[100] == 0.0) and not np.any(np.all(y[:100] == 0.0, axis=1))
    with pytest.raises(ValueError):
        W.log_mel_spectrogram(np.zeros(200, dtype=np.float32), filters)


def test_slaney_filterbank_properties():
    f = W.slaney_mel_filters(16000, 400, 80)
    assert f.shape == (80, 201) and f.dtype == np.float32 and f.min() >= 0
    assert np.all(f[:, 0] == 0)  # fmin = 0 is the left foot of the first triangle
    peaks = f.argmax(axis=1)
    assert np.all(np.diff(peaks) >= 0) and peaks[0] == 1 and peaks[-1] >= 190
    # slaney scale: linear below 1 kHz -> the first triangles are equally wide (66.67 Hz apart, bins are 40 Hz)
    centres = W.mel_to_hz_slaney(np.linspace(W.hz_to_mel_slaney(0.0), W.hz_to_mel_slaney(8000.0), 82))
    assert np.allclose(np.diff(centres[:15]), np.diff(centres[:15])[0])
    assert np.allclose(W.hz_to_mel_slaney(W.mel_to_hz_slaney(np.arange(0, 60, 0.5))), np.arange(0, 60, 0.5))
    assert abs(float(W.hz_to_mel_slaney(1000.0)) - 15.0) < 1e-12
    # area normalisation: each continuous triangle has unit area -> discrete sums ~ 1 / (40 Hz bin width) for wide filters
    wide = f[40:].astype(np.float64).sum(axis=1) * 40.0
    assert np.all(np.abs(wide - 1.0) < 0.08)
    # known-answer tripwires: the leading non-zero weights of librosa.filters.mel(sr=16000, n_fft=400, n_mels=80 / 128),
    # i.e. of the arrays OpenAI Whisper ships as assets/mel_filters.npz ("mel_80", "mel_128") -- quoted from memory of
    # that published asset, which is not available offline
    assert abs(float(f[0, 1]) - 0.02486259) < 5e-9 and abs(float(f[1, 1]) - 0.00199082) < 5e-9 and abs(float(f[1, 2]) - 0.02287177) < 5e-9
    assert abs(float(W.slaney_mel_filters(16000, 400, 128)[0, 1]) - 0.01237399) < 5e-9
    # window
    w = W.hann_periodic()
    assert w[0] == 0 and abs(w[200] - 1) < 1e-7 and np.allclose(w[1:], w[1:][::-1])
