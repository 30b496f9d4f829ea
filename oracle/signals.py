"""
TEST INFRASTRUCTURE -- deterministic input signals shared by oracle/make_golden.py,
tests/ and bench.py.  ``np.random.RandomState`` is used because its stream is frozen
across numpy versions, so the GPU box regenerates bit-identical inputs; every golden
file also stores a CRC of its input to detect drift.
"""
from __future__ import annotations

import zlib

import numpy as np


def make_signal(kind: str, num_samples: int, seed: int = 0, sampling_rate: int = 16000) -> np.ndarray:
    """float32 waveform in [-1, 1] of the given family."""
    rs = np.random.RandomState(seed)
    if kind == "uniform":  # U(-1,1)*0.5
        x = rs.uniform(-1.0, 1.0, size=num_samples) * 0.5
    elif kind == "gauss":  # unit-variance-ish Gaussian clipped to [-1,1]
        x = np.clip(rs.randn(num_samples) * 0.25, -1.0, 1.0)
    elif kind == "tone":  # SURVEY section 8c known-answer input (sr-independent formula uses 16 kHz)
        n = np.arange(num_samples, dtype=np.float64)
        x = 0.5 * np.sin(2 * np.pi * 440 * n / 16000) + 0.25 * np.sin(2 * np.pi * 3000 * n / 16000)
    elif kind == "speechlike":  # AM-modulated coloured noise with silences and a DC offset
        n = np.arange(num_samples, dtype=np.float64)
        w = rs.randn(num_samples)
        w = np.convolve(w, np.array([1.0, 0.9, 0.5, 0.2]), mode="same")
        env = np.clip(np.sin(2 * np.pi * 3.0 * n / sampling_rate + seed), 0.0, 1.0) ** 2
        x = 0.15 * w * env + 0.01 + 1e-4 * rs.randn(num_samples)
        x = np.clip(x, -1.0, 1.0)
    elif kind == "zeros":
        x = np.zeros(num_samples)
    elif kind == "dc":
        x = np.full(num_samples, 0.25)
    elif kind == "impulse":
        x = np.zeros(num_samples)
        x[num_samples // 3] = 1.0
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(x, dtype=np.float32)


def crc(x: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(x).tobytes()) & 0xFFFFFFFF
