#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden vectors for the post-feature transforms, produced by running the REFERENCE's
``SpecAugment.forward`` and ``GlobalMVN.forward / inverse`` (lhotse/dataset/signal_transforms.py) on CPU with seeded
RNGs (python ``random``, ``numpy.random``, ``torch`` CPU generator -- all three are used by the reference)."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.make_golden import import_reference  # noqa: E402

# name, constructor kwargs, (B, T, F), supervision segments or None, seed
CASES = [
    ("specaug_default", {}, (4, 260, 80), None, 1),
    ("specaug_small_warp", {"time_warp_factor": 10, "num_frame_masks": 3, "frames_mask_size": 20, "p": 1.0}, (5, 150, 40), None, 2),
    ("specaug_nowarp", {"time_warp_factor": None, "num_feature_masks": 1, "features_mask_size": 8, "num_frame_masks": 1, "frames_mask_size": 30, "p": 1.0}, (4, 333, 23), None, 3),
    ("specaug_supervisions", {"time_warp_factor": 20, "p": 1.0}, (4, 500, 32),
     [[0, 0, 500], [1, 10, 200], [1, 250, 240], [2, 100, 390], [3, 0, 120], [3, 100, 200], [3, 280, 400]], 4),
    ("specaug_p_half", {"time_warp_factor": 5, "p": 0.5}, (12, 120, 16), None, 5),
]


def make_input(shape, seed):
    """log-mel-like values plus LOG_EPSILON padding rows at the end of every other sequence."""
    rng = np.random.RandomState(1000 + seed)
    x = (rng.randn(*shape) * 3.0 - 8.0).astype(np.float32)
    for b in range(1, shape[0], 2):
        x[b, shape[1] - 17 * b :] = np.float32(-23.025850929940457)
    return x


def seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main():
    import_reference()
    from lhotse.dataset.signal_transforms import GlobalMVN, SpecAugment

    out_dir = os.path.join(ROOT, "tests", "golden")
    arrays = {}
    for name, kw, shape, sup, seed in CASES:
        x = make_input(shape, seed)
        tfm = SpecAugment(**kw)
        seed_all(seed)
        y = tfm(torch.from_numpy(x), supervision_segments=None if sup is None else torch.tensor(sup, dtype=torch.int32))
        arrays[name] = y.numpy().astype(np.float32)
        print(name, y.shape, float((y.numpy() != x).mean()))
    rng = np.random.RandomState(7)
    x = (rng.randn(3, 50, 80) * 4 - 6).astype(np.float32)
    mvn = GlobalMVN(80)
    mvn.load_state_dict({"norm_means": torch.from_numpy(rng.randn(80).astype(np.float32) - 7), "norm_stds": torch.from_numpy((rng.rand(80) * 3 + 0.5).astype(np.float32))})
    arrays["mvn_in"] = x
    arrays["mvn_means"] = mvn.norm_means.numpy()
    arrays["mvn_stds"] = mvn.norm_stds.numpy()
    arrays["mvn_forward"] = mvn(torch.from_numpy(x)).numpy()
    arrays["mvn_inverse"] = mvn.inverse(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(out_dir, "specaug.npz"), **arrays)


if __name__ == "__main__":
    main()
