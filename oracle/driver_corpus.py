"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

The small int16 WAV corpus behind the "reference-driver goldens" (round 6): oracle/make_golden_drivers.py runs LHOTSE'S OWN DRIVERS with the
reference's Fbank over these files in the authoring container and commits what they stored / returned
(tests/golden/drivers.npz + drivers.json); tests/test_gpu_reference_drivers.py regenerates the very same files on the GPU box (stdlib
`wave`, seeded signals, CRC-checked) and drives the real HIP plan through the same call shapes.  bench.py --config plumbing writes its
10 s cuts with the same writer.

Why goldens and not the reference itself on the GPU box: a Python reference cannot travel in any form (task rules); what travels is data --
inputs and the reference's outputs.
"""
from __future__ import annotations

import os
import wave
import zlib
from typing import Dict, List, Optional

import numpy as np

from .signals import make_signal

SAMPLING_RATE = 16000
# (id, signal family, samples, seed): mixed lengths 0.3 ... 2.5 s, realistic dynamic range (voiced / speechlike) next to noise
CORPUS = [
    ("utt0", "voiced", 16000, 1),
    ("utt1", "uniform", 24000, 2),
    ("utt2", "speechlike", 12345, 3),
    ("utt3", "voiced", 32000, 4),
    ("utt4", "uniform", 8000, 5),
    ("utt5", "voiced", 40000, 6),
    ("utt6", "speechlike", 5000, 7),
    ("utt7", "uniform", 20480, 8),
]


def pcm16(kind: str, num_samples: int, seed: int, sampling_rate: int = SAMPLING_RATE) -> np.ndarray:
    """The int16 samples of one file."""
    return (make_signal(kind, num_samples, seed, sampling_rate).astype(np.float64) * 32767.0).astype(np.int16)


def write_wav(path, pcm: np.ndarray, sampling_rate: int = SAMPLING_RATE) -> None:
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sampling_rate)
        f.writeframes(np.ascontiguousarray(pcm, dtype=np.int16).tobytes())


def read_wav(path, offset_samples: int = 0, num_samples: Optional[int] = None) -> np.ndarray:
    """-> (1, T) float32 in [-1, 1): int16 / 32768, what lhotse's audio backends hand to `Cut.load_audio()` (soundfile's float32 read
    of a PCM_16 file is the same division) -- and what tests/_dropin_support.StdlibWaveBackend does under the real lhotse."""
    with wave.open(str(path), "rb") as f:
        n, ch = f.getnframes(), f.getnchannels()
        f.setpos(offset_samples)
        raw = f.readframes(n - offset_samples if num_samples is None else num_samples)
    return np.frombuffer(raw, dtype=np.int16).reshape(-1, ch).T.astype(np.float32) / 32768.0


def read_pcm16(path) -> np.ndarray:
    with wave.open(str(path), "rb") as f:
        return np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).copy()


def write_corpus(directory, corpus=CORPUS) -> List[Dict]:
    """Write the files; -> [{"id", "path", "num_samples", "crc"}] in corpus order."""
    os.makedirs(directory, exist_ok=True)
    out = []
    for cid, kind, n, seed in corpus:
        pcm = pcm16(kind, n, seed)
        path = os.path.join(str(directory), f"{cid}.wav")
        write_wav(path, pcm)
        out.append({"id": cid, "path": path, "num_samples": int(n), "crc": zlib.crc32(pcm.tobytes()) & 0xFFFFFFFF})
    return out
