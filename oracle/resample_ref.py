"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement (numpy) of the sinc resampler behind lhotse's speed perturbation:
``Speed.__call__`` (lhotse/augmentation/torchaudio.py:37-42) -> ``get_or_create_resampler(round(sr*factor), sr)``
-> ``ResampleTensor`` (lhotse/augmentation/resample.py:42-142) =
``_get_sinc_resample_kernel`` (:184-281) + ``_apply_sinc_resample_kernel`` (:284-315).

Parity status: PINNED -- tests/test_resample_oracle.py checks it against tests/golden/resample_*.npz, produced by
oracle/make_golden_resample.py from the reference itself.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tuple[np.ndarray, int, int, int]:
    """-> (kernel[new][2*width+orig] float32, width, orig, new) with orig/new reduced by their gcd.
    resample.py:184-281 (hann-windowed sinc, evaluated in float64, cached as float32)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    # NB the phase term is float32 in the reference (torch.arange(..., dtype=None) / new_freq, resample.py:249-253)
    # and only then promoted to float64 by the addition
    phase = (np.arange(0, -new, -1, dtype=np.float32)[:, None] / np.float32(new)).astype(np.float64)
    t = phase + idx
    t = t * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * scale
    return k.astype(np.float32), width, orig, new


def resampled_length(num_samples: int, orig: int, new: int) -> int:
    """resample.py:309: ceil(new * length / orig) -- evaluated in float32 by torch.as_tensor(float)!"""
    return int(np.ceil(np.float32(new * num_samples / orig)))


def resample(x: np.ndarray, orig_freq: int, new_freq: int, dtype=np.float32) -> np.ndarray:
    """resample.py:284-315 for one waveform (T,)."""
    if int(orig_freq) == int(new_freq):
        return np.asarray(x)
    k, width, orig, new = sinc_kernel(orig_freq, new_freq)
    x = np.asarray(x, dtype=dtype).reshape(-1)
    length = len(x)
    xp = np.concatenate([np.zeros(width, dtype), x, np.zeros(width + orig, dtype)])
    kw = 2 * width + orig
    nj = (len(xp) - kw) // orig + 1  # conv1d output length with stride orig
    # frames[j, i] = xp[j*orig + i]
    idx = (np.arange(nj) * orig)[:, None] + np.arange(kw)[None, :]
    y = xp[idx] @ k.astype(dtype).T  # (nj, new)
    y = y.reshape(-1)
    return y[: resampled_length(length, orig, new)].astype(dtype)


def speed(x: np.ndarray, sampling_rate: int, factor: float, dtype=np.float32) -> np.ndarray:
    """torchaudio.py:37-42"""
    return resample(x, round(sampling_rate * factor), sampling_rate, dtype)
