"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement (numpy, float32) of what lhotse's SpecAugment does to a batch once its random choices are fixed
(lhotse/dataset/signal_transforms.py:173-371), and of GlobalMVN (:50-60).

  * ``bicubic_rows`` -- ``F.interpolate(x[None, None], size=(out, F), mode="bicubic", align_corners=False)`` along the
    time axis (the feature axis keeps its size, so its interpolation is the identity): torch's cubic convolution,
    A = -0.75, source index ``scale * (dst + 0.5) - 0.5``, neighbours clamped to the edges (ATen UpSample.h).
  * ``time_warp`` (:338-371), ``apply`` = warp segments, then per-sequence mean fill of the mask regions (:239-266).

Parity: PINNED -- tests/golden/specaug_*.npz hold outputs of the reference's ``SpecAugment.forward`` /
``GlobalMVN.forward`` themselves for seeded RNGs (oracle/make_golden_specaug.py); tests/test_specaug_oracle.py replays the
same seeds through the product's draw logic + this restatement.
"""
from __future__ import annotations

import numpy as np

A = np.float32(-0.75)
f32 = np.float32


def _fma(a, b, c):
    """fmaf on float32 arrays: the product is exact in float64, one rounding at the end."""
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.float64(c)).astype(np.float32)


def _cc1(x):  # ((A + 2) x - (A + 3)) x x + 1
    return _fma(_fma(A + f32(2), x, -(A + f32(3))) * x, x, 1.0)


def _cc2(x):  # ((A x - 5A) x + 8A) x - 4A
    return _fma(_fma(_fma(A, x, -f32(5) * A), x, f32(8) * A), x, -f32(4) * A)


def bicubic_rows(x: np.ndarray, out_len: int) -> np.ndarray:
    """x: (in_len, F) float32 -> (out_len, F) float32.  Multiply-adds are fused (one rounding), as in the reference's
    compiled kernels (FMA contraction: x86 AVX2 / AVX-512 builds of ATen and its GPU kernels alike) -- measured: with
    unfused arithmetic the source index of long segments differs in the last bit and the result by up to 3e-4."""
    x = np.asarray(x, dtype=np.float32)
    in_len = x.shape[0]
    scale = f32(in_len) / f32(out_len)
    real = _fma(scale, np.arange(out_len, dtype=np.float32) + f32(0.5), -0.5)
    fl = np.floor(real)
    t = (real - fl).astype(np.float32)
    idx = fl.astype(np.int64)
    w = [_cc2(t + f32(1)), _cc1(t), _cc1(f32(1) - t), _cc2(f32(2) - t)]
    out = np.zeros((out_len, x.shape[1]), dtype=np.float32)
    for j in range(4):
        out = _fma(w[j][:, None], x[np.clip(idx - 1 + j, 0, in_len - 1)], out)
    return out


def time_warp(x: np.ndarray, center: int, warped: int) -> np.ndarray:
    t = x.shape[0]
    return np.concatenate([bicubic_rows(x[:center], warped), bicubic_rows(x[center:], t - warped)], axis=0)


def apply(features: np.ndarray, seg_rounds, masks) -> np.ndarray:
    """features (B, T, F); seg_rounds: list of record arrays (sequence, start, num_frames, center, warped) applied in
    order; masks: records (sequence, axis, begin, end) filled with the mean of the warped sequence."""
    out = np.array(features, dtype=np.float32, copy=True)
    for segs in seg_rounds:
        for s in segs:
            b, st, n = int(s["sequence"]), int(s["start"]), int(s["num_frames"])
            out[b, st : st + n] = time_warp(out[b, st : st + n], int(s["center"]), int(s["warped"]))
    means = {}
    for m in masks:
        b = int(m["sequence"])
        if b not in means:
            means[b] = np.float32(out[b].astype(np.float64).mean())
    for m in masks:
        b, lo, hi = int(m["sequence"]), int(m["begin"]), int(m["end"])
        if int(m["axis"]) == 1:
            out[b, lo:hi, :] = means[b]
        else:
            out[b, :, lo:hi] = means[b]
    return out


def global_mvn(x, means, stds, inverse=False):
    x, means, stds = (np.asarray(a, dtype=np.float32) for a in (x, means, stds))
    return (x * stds + means) if inverse else ((x - means) / stds)
