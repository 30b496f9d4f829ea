#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden vectors for the librosa log-mel row, produced by running the REFERENCE's
``LibrosaFbank.extract`` (lhotse/features/librosa_fbank.py:139-161).  librosa itself is not available offline: the two
entry points the reference calls (``librosa.stft``, ``librosa.filters.mel``) are served by the restatements in
oracle/librosa_ref.py, so these vectors pin lhotse's own steps (magnitude, mel product, log10 floor, row count /
truncation), not librosa's."""
import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import librosa_ref  # noqa: E402
from oracle.make_golden import import_reference  # noqa: E402
from oracle.signals import crc, make_signal  # noqa: E402

CASES = [  # (name, config overrides, [(signal, n, seed)])
    ("librosa_default", {}, [("uniform", 22050, 1), ("speechlike", 50001, 2), ("tone", 22050, 0), ("gauss", 22143, 3), ("gauss", 22144, 4),
                            ("uniform", 513, 5), ("zeros", 4096, 0), ("impulse", 6000, 0)]),
    ("librosa_16k_win", {"sampling_rate": 16000, "fft_size": 512, "hop_size": 160, "win_length": 400, "num_mel_bins": 40, "fmin": 0, "fmax": None,
                         "window": "hamming"}, [("uniform", 16000, 6), ("speechlike", 24001, 7), ("gauss", 1000, 8)]),
    ("librosa_24k_odd", {"sampling_rate": 24000, "fft_size": 1200, "hop_size": 300, "num_mel_bins": 100, "fmin": 50, "fmax": 11000},
     [("uniform", 24000, 9), ("gauss", 30011, 10)]),
]


def install_librosa_stub():
    m = types.ModuleType("librosa")
    m.__spec__ = importlib.machinery.ModuleSpec("librosa", None)
    m.__file__ = "<stub librosa: oracle/librosa_ref.py>"
    m.stft = librosa_ref.stft
    m.filters = types.SimpleNamespace(mel=librosa_ref.mel)
    sys.modules["librosa"] = m


def main():
    import_reference()
    install_librosa_stub()
    from lhotse.features.librosa_fbank import LibrosaFbank, LibrosaFbankConfig

    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, over, inputs in CASES:
        cfg = LibrosaFbankConfig(**over)
        ex = LibrosaFbank(cfg)
        arrays = {"config": np.array(repr(sorted(over.items())))}
        for i, (kind, n, seed) in enumerate(inputs):
            x = make_signal(kind, n, seed)
            y = ex.extract(x, cfg.sampling_rate)
            arrays[f"out{i}"] = np.asarray(y).astype(np.float32)
            arrays[f"crc{i}"] = np.array(crc(x), dtype=np.uint64)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrays)
        print(name, [arrays[f"out{i}"].shape for i in range(len(inputs))])


if __name__ == "__main__":
    main()
