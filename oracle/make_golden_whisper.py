#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden vectors for the Whisper log-mel row, produced by running the REFERENCE's
``log_mel_spectrogram`` (lhotse/features/whisper_fbank.py:17-85) exactly as ``WhisperFbank.extract`` calls it
(:158-165), with the filterbank restated in oracle/whisper_ref.py (librosa is not available offline)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.make_golden import import_reference  # noqa: E402
from oracle.signals import crc, make_signal  # noqa: E402
from oracle.whisper_ref import slaney_mel_filters  # noqa: E402

CASES = [  # (name, n_mels, [(signal, n, seed)])
    ("whisper_80", 80, [("uniform", 16000, 1), ("speechlike", 40123, 2), ("tone", 16000, 0), ("gauss", 16079, 3), ("gauss", 16080, 4),
                        ("uniform", 201, 5), ("zeros", 3200, 0), ("impulse", 4800, 0)]),
    ("whisper_128", 128, [("uniform", 32000, 6), ("speechlike", 24001, 7)]),
]


def main():
    import_reference()
    from lhotse.features.whisper_fbank import log_mel_spectrogram

    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, n_mels, inputs in CASES:
        filters = slaney_mel_filters(16000, 400, n_mels)
        arrays = {"filters": filters}
        for i, (kind, n, seed) in enumerate(inputs):
            x = make_signal(kind, n, seed)
            y = log_mel_spectrogram(x[None, :], filters=torch.from_numpy(filters), n_fft=400, window=torch.hann_window(400), n_mels=n_mels, device="cpu")
            arrays[f"out{i}"] = y.numpy().astype(np.float32)
            arrays[f"crc{i}"] = np.array(crc(x), dtype=np.uint64)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrays)
        print(name, [arrays[f"out{i}"].shape for i in range(len(inputs))])


if __name__ == "__main__":
    main()
