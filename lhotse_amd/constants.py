"""
Host-side constants of the feature pipeline (window, mel filterbank, DCT, lifter, sizes).

They are computed once per plan on the host and handed to the device library as float32
arrays (include/hipfeat.h, hipfeat_plan_create), so the kernels never re-derive them.
Each function evaluates the reference's formula with the reference's dtype flow, so the
arrays are bit-identical to the ``nn.Parameter``s the reference layers hold
(tests/test_constants.py checks that against fixtures produced by the reference):

  window   lhotse/features/kaldi/layers.py:921-940
  mel      lhotse/features/kaldi/layers.py:960-1017 (+ zero column / transpose :553), :873-907
  dct      lhotse/features/kaldi/layers.py:697-706
  lifter   lhotse/features/kaldi/layers.py:681-695
  sizes    lhotse/features/kaldi/layers.py:114-116, :264-265
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

MEL_FLOOR = float(torch.finfo(torch.float32).eps)  # layers.py:536-538
LOG_SPEC_OFFSET = 1e-15  # layers.py:467

WINDOW_TYPES = ("hamming", "hanning", "povey", "rectangular", "blackman", "hann_periodic")


def frame_sizes(sampling_rate: int, frame_length: float, frame_shift: float, round_to_power_of_two: bool) -> Tuple[int, int, int]:
    """(samples per frame, samples per shift, fft length)."""
    n = int(math.floor(frame_length * sampling_rate))
    shift = int(math.floor(frame_shift * sampling_rate))
    fft = (1 if n == 0 else 2 ** (n - 1).bit_length()) if round_to_power_of_two else n
    return n, shift, fft


def make_window(n: int, window_type: str, blackman_coeff: float = 0.42) -> np.ndarray:
    # torch's float32 window kernels are used so the values equal the reference's bit for bit
    if window_type == "hanning":
        w = torch.hann_window(n, periodic=False)
    elif window_type == "hamming":
        w = torch.hamming_window(n, periodic=False, alpha=0.54, beta=0.46)
    elif window_type == "povey":
        w = torch.hann_window(n, periodic=False).pow(0.85)
    elif window_type == "hann_periodic":  # torch.hann_window(n), the STFT window of whisper_fbank.py:115,120
        w = torch.hann_window(n)
    elif window_type == "rectangular":
        w = torch.ones(n, dtype=torch.float32)
    elif window_type == "blackman":
        step = 2 * math.pi / n
        i = torch.arange(n, dtype=torch.float32)
        w = blackman_coeff - 0.5 * torch.cos(step * i) + (0.5 - blackman_coeff) * torch.cos(2 * step * i)
    else:
        raise ValueError(f"Invalid window type: {window_type} (expected one of {WINDOW_TYPES})")
    return np.ascontiguousarray(w.to(torch.float32).numpy())


def _mel_of_hz(f):
    return 1127.0 * np.log(1 + f / 700)


def make_kaldi_mel(num_filters: int, fft: int, sampling_rate: float, low_freq: float, high_freq: float) -> np.ndarray:
    """Kaldi/torchaudio triangular filters, shape (fft/2+1, num_filters), float32."""
    if num_filters <= 3:
        raise ValueError("Must have at least 3 mel bins")
    if fft % 2 != 0:
        raise ValueError(f"fft length {fft} must be even for a mel filterbank")
    nyquist = 0.5 * sampling_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    if not (0.0 <= low_freq < nyquist and 0.0 < high_freq <= nyquist and low_freq < high_freq):
        raise ValueError(f"Bad values in options: low-freq {low_freq} and high-freq {high_freq} vs. nyquist {nyquist}")
    f32 = np.float32
    mel_lo = float(_mel_of_hz(low_freq))
    mel_step = (float(_mel_of_hz(high_freq)) - mel_lo) / (num_filters + 1)
    idx = np.arange(num_filters, dtype=f32)[None, :]  # filters along columns from the start
    left = f32(mel_lo) + idx * f32(mel_step)
    center = f32(mel_lo) + (idx + f32(1.0)) * f32(mel_step)
    right = f32(mel_lo) + (idx + f32(2.0)) * f32(mel_step)
    hz = f32(sampling_rate / fft) * np.arange(fft // 2, dtype=f32)
    mel = (f32(1127.0) * np.log(f32(1) + hz / f32(700)))[:, None]
    rising = (mel - left) / (center - left)
    falling = (right - mel) / (right - center)
    tri = np.maximum(f32(0), np.minimum(rising, falling))
    out = np.zeros((fft // 2 + 1, num_filters), dtype=f32)  # last row = Nyquist bin, all zero
    out[: fft // 2] = tri
    return out


def make_htk_mel(num_filters: int, fft: int, sampling_rate: int, low_freq: float, high_freq: Optional[float], norm_filters: bool) -> np.ndarray:
    """The ``torchaudio_compatible_mel_scale=False`` variant, shape (fft/2+1, num_filters)."""
    if high_freq is None or high_freq == 0:
        high_freq = sampling_rate / 2
    if high_freq < 0:
        high_freq = sampling_rate / 2 + high_freq
    edges = np.linspace(_mel_of_hz(low_freq), _mel_of_hz(high_freq), num_filters + 2)
    bin_mel = _mel_of_hz(np.linspace(0, sampling_rate, fft))[: fft // 2]
    out = np.zeros((fft // 2 + 1, num_filters), dtype=np.float32)
    for j in range(num_filters):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        up = (lo < bin_mel) & (bin_mel <= mid) & (bin_mel < hi)
        down = (bin_mel > mid) & (bin_mel < hi) & (lo < bin_mel)
        out[: fft // 2, j][up] = (bin_mel[up] - lo) / (mid - lo)
        out[: fft // 2, j][down] = (hi - bin_mel[down]) / (hi - mid)
    if norm_filters:
        out = out / np.sum(out, axis=0, keepdims=True)
    return np.ascontiguousarray(out, dtype=np.float32)


def make_stft_window(window: str, win_length: int, fft: int) -> np.ndarray:
    """The analysis window of ``librosa.stft(window=..., win_length=...)``: ``scipy.signal.get_window(window, win_length,
    fftbins=True)`` (periodic) zero-padded on both sides to ``fft`` samples (librosa.util.pad_center)."""
    n = np.arange(win_length, dtype=np.float64)
    x = 2.0 * np.pi * n / win_length
    if window in ("hann", "hanning"):
        w = 0.5 - 0.5 * np.cos(x)
    elif window == "hamming":
        w = 0.54 - 0.46 * np.cos(x)
    elif window == "blackman":
        w = 0.42 - 0.5 * np.cos(x) + 0.08 * np.cos(2.0 * x)
    elif window in ("boxcar", "rectangular", "ones", "rect", "box"):
        w = np.ones(win_length)
    else:
        from scipy.signal import get_window  # any other scipy window name

        w = get_window(window, win_length, fftbins=True)
    left = (fft - win_length) // 2
    out = np.zeros(fft, dtype=np.float64)
    out[left : left + win_length] = w
    return out.astype(np.float32)


def make_dct(num_ceps: int, num_filters: int) -> np.ndarray:
    """(num_filters, num_ceps) DCT-II basis, first column scaled by 1/sqrt(2)."""
    rows = torch.arange(float(num_filters)).unsqueeze(1)
    cols = torch.arange(float(num_ceps))
    basis = torch.cos(math.pi / float(num_filters) * (rows + 0.5) * cols)
    basis[:, 0] *= 1.0 / math.sqrt(2.0)
    basis *= math.sqrt(2.0 / float(num_filters))
    return np.ascontiguousarray(basis.numpy(), dtype=np.float32)


def make_lifter(num_ceps: int, q: int) -> np.ndarray:
    if q <= 0:
        return np.ones(num_ceps, dtype=np.float32)
    v = 1 + 0.5 * q * torch.sin(math.pi * torch.arange(num_ceps, dtype=torch.float32) / q)
    return np.ascontiguousarray(v.numpy(), dtype=np.float32)


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tuple[np.ndarray, int, int, int]:
    """Filter bank of the polyphase sinc resampler: (kernel[new][2*width + orig] float32, width, orig, new)
    with orig/new divided by their gcd -- lhotse/augmentation/resample.py:184-281 (hann-windowed sinc,
    evaluated in float64 and stored as float32 like the reference's cached buffer).  Evaluated with torch so
    the transcendental kernels (and therefore the bits) are the reference's."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Frequencies must be positive integers")
    if lowpass_filter_width <= 0:
        raise ValueError("Low pass filter width should be positive.")
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    taps = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    # the phase offsets are float32 in the reference (arange without dtype, :249-253) and promoted by the sum
    t = torch.arange(0, -new, -1)[:, None, None] / new + taps
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    k = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    k = k * window * scale
    return np.ascontiguousarray(k.to(torch.float32).reshape(new, 2 * width + orig).numpy()), width, orig, new


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * (3.0 / 200.0)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / (np.log(6.4) / 27.0), lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3.0))


def make_slaney_mel(num_filters: int, fft: int, sampling_rate: int, fmin: float = 0.0, fmax: Optional[float] = None) -> np.ndarray:
    """(fft/2+1, M) float32 -- the transpose of ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` (slaney mel scale,
    triangles built from frequency ramps, area ("slaney") normalisation), which the reference loads in
    lhotse/features/whisper_fbank.py:116-119 and multiplies from the left (:65), and builds with fmin/fmax in
    lhotse/features/librosa_fbank.py:121-125."""
    fmax = sampling_rate / 2.0 if fmax is None else float(fmax)
    bins = np.fft.rfftfreq(fft, 1.0 / sampling_rate)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(float(fmin)), _slaney_hz_to_mel(fmax), num_filters + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - bins[None, :]
    rising = -ramps[:-2] / width[:-1, None]
    falling = ramps[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return np.ascontiguousarray(tri.astype(np.float32).T)
