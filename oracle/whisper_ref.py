"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement (numpy) of lhotse's Whisper log-mel front end:
``log_mel_spectrogram`` (lhotse/features/whisper_fbank.py:17-85) as called by ``WhisperFbank.extract`` (:139-167)
with ``n_fft=400, hop_length=160, window=torch.hann_window(400)`` (periodic) and a slaney mel filterbank.

Parity status
  * arithmetic (reflect-pad STFT, power, mel GEMM, log10 / dynamic-range clamp / affine, zero padding row): PINNED --
    tests/test_whisper_oracle.py checks it against tests/golden/whisper_*.npz produced by running the reference's
    ``log_mel_spectrogram`` itself (oracle/make_golden_whisper.py).
  * the filterbank VALUES: the reference takes them from ``librosa.filters.mel(sr=16000, n_fft=400, n_mels=80)``
    (whisper_fbank.py:116-119); librosa (third party, unpinned in the reference's setup.py) is not available offline,
    so ``slaney_mel_filters`` restates its published algorithm (librosa/filters.py ``mel``: slaney mel scale, triangular
    weights from ramps, ``norm="slaney"`` area normalisation, float32) -- "parity unpinned" for these constants.
    Known properties are asserted instead (shape, partition-like overlap, peak positions, area normalisation).
"""
from __future__ import annotations

import numpy as np

N_FFT = 400
HOP = 160
SAMPLING_RATE = 16000


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def slaney_mel_filters(sr: int = SAMPLING_RATE, n_fft: int = N_FFT, n_mels: int = 80) -> np.ndarray:
    """-> (n_mels, 1 + n_fft // 2) float32, as librosa.filters.mel(sr=sr, n_fft=n_fft, n_mels=n_mels)."""
    fftfreqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(0.0), hz_to_mel_slaney(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, len(fftfreqs)), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def hann_periodic(n: int = N_FFT, dtype=np.float32) -> np.ndarray:
    """torch.hann_window(n) (periodic=True): 0.5 - 0.5 cos(2 pi i / n)"""
    i = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * i / n)).astype(dtype)


def num_rows(num_samples: int, hop: int = HOP) -> int:
    """compute_num_frames_from_samples (lhotse/utils.py:424-434) -- rows of the returned matrix."""
    return (num_samples + hop // 2) // hop


def log_mel_spectrogram(audio: np.ndarray, filters: np.ndarray, n_fft: int = N_FFT, hop: int = HOP, dtype=np.float32) -> np.ndarray:
    """whisper_fbank.py:48-85 for one waveform (T,) -> (num_rows, n_mels)."""
    x = np.asarray(audio, dtype=dtype).reshape(-1)
    pad = n_fft // 2
    if len(x) <= pad:
        raise ValueError(f"waveform of {len(x)} samples is not longer than the reflect padding ({pad})")  # torch.stft raises
    xp = np.pad(x, (pad, pad), mode="reflect")  # torch.stft(center=True, pad_mode="reflect")
    nfr = 1 + (len(xp) - n_fft) // hop
    idx = (np.arange(nfr) * hop)[:, None] + np.arange(n_fft)[None, :]
    frames = xp[idx] * hann_periodic(n_fft, dtype)[None, :]
    spec = np.fft.rfft(frames.astype(dtype), axis=1)
    mag = (np.abs(spec[:-1]) ** 2).astype(dtype)  # drop the last frame (:63)
    mel = mag @ filters.astype(dtype).T  # (T, n_mels)
    log_spec = np.log10(np.maximum(mel, dtype(1e-10)))
    log_spec = np.maximum(log_spec, log_spec.max() - dtype(8.0))
    log_spec = ((log_spec + dtype(4.0)) / dtype(4.0)).astype(dtype)
    rows = num_rows(len(x), hop)
    if rows > log_spec.shape[0]:
        log_spec = np.concatenate([log_spec, np.zeros((rows - log_spec.shape[0], log_spec.shape[1]), dtype)])
    return log_spec
