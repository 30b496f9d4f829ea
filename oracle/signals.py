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
    elif kind in ("voiced", "voiced_quiet"):
        # A speech-like signal with the dynamic range white noise lacks (white noise is the EASIEST input for a log-mel):
        # a glottal pulse train with a wandering f0 (90-220 Hz) through three formant resonators, unvoiced (noise) stretches,
        # silences with a noise floor 70 dB down, a small DC offset; "voiced_quiet" is the same at 1e-3 of the amplitude.
        from scipy.signal import lfilter

        n = np.arange(num_samples, dtype=np.float64)
        t = n / sampling_rate
        f0 = 150.0 + 60.0 * np.sin(2 * np.pi * 0.7 * t + seed) + 10.0 * np.sin(2 * np.pi * 5.1 * t)
        phase = np.cumsum(f0) / sampling_rate
        pulses = np.diff(np.floor(phase), prepend=0.0)  # one unit impulse per glottal cycle
        src = lfilter([1.0], [1.0, -0.96], pulses)       # -6 dB/octave glottal tilt
        noise = rs.randn(num_samples)
        seg = np.floor(t * 4.0 + 0.37 * seed).astype(np.int64) % 5  # 250 ms segments: voiced, voiced, unvoiced, voiced, silence
        x = np.where(seg == 2, 0.05 * noise, np.where(seg == 4, 0.0, src))
        for fc, bw in ((700.0, 110.0), (1220.0, 140.0), (2600.0, 200.0)):
            if fc < 0.45 * sampling_rate:
                r = np.exp(-np.pi * bw / sampling_rate)
                x = lfilter([1.0 - r], [1.0, -2.0 * r * np.cos(2 * np.pi * fc / sampling_rate), r * r], x)
        x = x / (np.abs(x).max() + 1e-12) * 0.6
        x = x + 0.6 * 10.0 ** (-70.0 / 20.0) * rs.randn(num_samples) + 0.003
        if kind == "voiced_quiet":
            x = x * 1e-3
        x = np.clip(x, -1.0, 1.0)
    elif kind == "speechlike_quiet":  # the AM-modulated coloured noise at 1e-3 of its amplitude (60 dB down)
        return np.ascontiguousarray(make_signal("speechlike", num_samples, seed, sampling_rate).astype(np.float64) * 1e-3, dtype=np.float32)
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
