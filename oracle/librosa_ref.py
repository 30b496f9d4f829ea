"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement (numpy) of lhotse's librosa log-mel front end: ``logmelfilterbank``
(lhotse/features/librosa_fbank.py:66-137) as called by ``LibrosaFbank.extract`` (:159-161).

Parity status
  * lhotse's own steps -- ``np.abs``, ``spc @ mel.T``, ``log10(max(1e-10, .))``, ``compute_num_frames`` rows,
    ``pad_or_truncate_features`` (:119-137, :43-63): PINNED -- tests/golden/librosa_*.npz are produced by running the
    reference's ``LibrosaFbank.extract`` itself (oracle/make_golden_librosa.py) with the two librosa entry points it
    calls served by this file.
  * ``librosa.stft`` and ``librosa.filters.mel``: third party (librosa, unpinned in the reference's
    docs/requirements; not available offline), restated here from their published algorithm -- "parity unpinned" for
    those two.  ``stft`` is cross-checked against ``torch.stft`` (same published semantics: centred frames, "reflect"
    edges, periodic window zero-padded to n_fft) in tests/test_librosa_oracle.py; ``mel`` is the whisper row's
    ``slaney_mel_filters`` with fmin/fmax.
"""
from __future__ import annotations

import numpy as np

from oracle.whisper_ref import hz_to_mel_slaney, mel_to_hz_slaney

EPSILON = 1e-10  # lhotse/utils.py EPSILON


def get_window(window: str, win_length: int) -> np.ndarray:
    """scipy.signal.get_window(window, win_length, fftbins=True) for the cosine-sum family (float64)."""
    n = np.arange(win_length, dtype=np.float64)
    coeffs = {"hann": (0.5, 0.5), "hamming": (0.54, 0.46), "blackman": (0.42, 0.5, 0.08), "boxcar": (1.0,)}[window]
    w = np.zeros(win_length)
    for k, a in enumerate(coeffs):
        w += (-1) ** k * a * np.cos(2.0 * np.pi * k * n / win_length)
    return w


def stft(y: np.ndarray, n_fft: int = 2048, hop_length=None, win_length=None, window: str = "hann", center: bool = True,
         pad_mode: str = "reflect") -> np.ndarray:
    """librosa.stft -> complex64 (1 + n_fft/2, 1 + len(y) // hop): window of win_length samples centred in n_fft
    (librosa.util.pad_center), signal padded by n_fft // 2 on both sides (np.pad mode="reflect"), frames hop apart,
    rfft of window * frame."""
    assert center and pad_mode == "reflect"
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 4 if hop_length is None else hop_length
    w = np.zeros(n_fft)
    lpad = (n_fft - win_length) // 2
    w[lpad : lpad + win_length] = get_window(window, win_length)
    y = np.asarray(y)
    yp = np.pad(y, (n_fft // 2, n_fft // 2), mode="reflect")
    nfr = 1 + (len(yp) - n_fft) // hop_length
    idx = (np.arange(nfr) * hop_length)[:, None] + np.arange(n_fft)[None, :]
    return np.fft.rfft(w[None, :] * yp[idx], axis=1).T.astype(np.complex64)


def mel(sr: int, n_fft: int, n_mels: int = 128, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm="slaney") -> (n_mels, 1 + n_fft // 2) float32."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(float(fmin)), hz_to_mel_slaney(float(fmax)), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, len(fftfreqs)))
    for i in range(n_mels):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    weights *= (2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def num_rows(num_samples: int, hop: int) -> int:
    """compute_num_frames(duration=S/sr, frame_shift=hop/sr) (lhotse/utils.py:410-421) in integers."""
    return (num_samples + hop // 2) // hop


def logmelfilterbank(audio: np.ndarray, sampling_rate: int = 22050, fft_size: int = 1024, hop_size: int = 256, win_length=None,
                     window: str = "hann", num_mel_bins: int = 80, fmin=80, fmax=7600, eps: float = EPSILON) -> np.ndarray:
    """librosa_fbank.py:66-137 for one waveform (T,) -> (num_rows, num_mel_bins)."""
    audio = np.asarray(audio).reshape(-1)
    spc = np.abs(stft(audio, n_fft=fft_size, hop_length=hop_size, win_length=win_length, window=window)).T
    basis = mel(sampling_rate, fft_size, num_mel_bins, 0 if fmin is None else fmin, sampling_rate / 2 if fmax is None else fmax)
    feats = np.log10(np.maximum(eps, np.dot(spc, basis.T)))
    rows = num_rows(len(audio), hop_size)
    assert 0 <= feats.shape[0] - rows <= 1  # 1 + S // hop rows came out: pad_or_truncate_features (:43-63) only truncates
    return feats[:rows]
