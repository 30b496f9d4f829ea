#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden vectors for the speed-perturb / sinc-resample row, produced by running the
REFERENCE (lhotse/augmentation/torchaudio.py Speed, lhotse/augmentation/resample.py ResampleTensor)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.make_golden import import_reference  # noqa: E402
from oracle.signals import crc, make_signal  # noqa: E402

CASES = [  # (name, mode, a, b, [(signal, n, seed)])
    ("resample_speed09", "speed", 16000, 0.9, [("uniform", 16000, 1), ("speechlike", 4001, 2), ("uniform", 10, 3)]),
    ("resample_speed11", "speed", 16000, 1.1, [("uniform", 16000, 4), ("tone", 12345, 0), ("uniform", 9, 5)]),
    ("resample_speed095", "speed", 16000, 0.95, [("uniform", 8000, 6)]),
    ("resample_8k_16k", "resample", 8000, 16000, [("uniform", 4000, 7)]),
    ("resample_16k_8k", "resample", 16000, 8000, [("uniform", 8000, 8)]),
    ("resample_441_16", "resample", 44100, 16000, [("uniform", 4410, 9)]),
    ("resample_16_2205", "resample", 16000, 22050, [("uniform", 3200, 10)]),
]


def main():
    import_reference()
    from lhotse.augmentation.resample import Resample as ResampleTensor
    from lhotse.augmentation.torchaudio import Speed

    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, mode, a, b, inputs in CASES:
        arrays = {}
        for i, (kind, n, seed) in enumerate(inputs):
            x = make_signal(kind, n, seed)
            if mode == "speed":
                y = Speed(factor=b)(x[None, :], a)[0]
            else:
                y = ResampleTensor(a, b)(torch.from_numpy(x)[None, :])[0].numpy()
            arrays[f"out{i}"] = np.asarray(y, dtype=np.float32)
            arrays[f"crc{i}"] = np.array(crc(x), dtype=np.uint64)
        if mode == "speed":
            rs = ResampleTensor(round(a * b), a)
        else:
            rs = ResampleTensor(a, b)
        arrays["kernel"] = rs.kernel.numpy()[:, 0, :]
        arrays["width"] = np.array(rs.width)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrays)
        print(name, [arrays[f"out{i}"].shape for i in range(len(inputs))], arrays["kernel"].shape)


if __name__ == "__main__":
    main()
