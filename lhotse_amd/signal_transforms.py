"""
Post-feature transforms on the collated ``(B, T, F)`` batch, on the HIP path:

* ``HipGlobalMVN``    -- drop-in for ``GlobalMVN`` (lhotse/dataset/signal_transforms.py:16-60): one fused kernel,
  bit-identical results.
* ``HipSpecAugment``  -- drop-in for ``SpecAugment`` (:121-371): same constructor, ``state_dict`` and ``forward(features,
  supervision_segments)``; the random decisions are made on the host with *the reference's own RNG calls in the
  reference's order* (``random.random``, ``np.random.randint``, ``torch.randint``, ``torch.rand`` on the CPU generator),
  so a seeded run masks and warps exactly the frames the reference would; the whole batch is then processed by two
  kernels (time warp + clone + per-sequence sums; mean fill of the masked regions) instead of ~15 small torch kernels
  per sequence.

Both need the batch on the GPU (there is no CPU fallback) and return a new tensor on the same device.
"""
from __future__ import annotations

import math
import random
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib

__all__ = ["HipGlobalMVN", "HipSpecAugment"]


def _require_cuda(features: torch.Tensor, who: str) -> torch.Tensor:
    if not isinstance(features, torch.Tensor) or features.device.type != "cuda":
        raise _lib.HipFeatError(_lib.ERR_INVALID, f"{who} runs on an AMD GPU: pass a float32 tensor on a 'cuda' device (there is no CPU fallback)")
    if features.dtype != torch.float32:
        raise TypeError(f"{who}: expected float32 features, got {features.dtype}")
    return features.contiguous()


class HipGlobalMVN(torch.nn.Module):
    """Apply global mean and variance normalization (GlobalMVN, signal_transforms.py:16-60)."""

    def __init__(self, feature_dim: int):
        super().__init__()
        self.feature_dim = feature_dim
        self.register_buffer("norm_means", torch.zeros(feature_dim))
        self.register_buffer("norm_stds", torch.ones(feature_dim))

    @classmethod
    def from_cuts(cls, cuts, max_cuts: Optional[int] = None, extractor=None) -> "HipGlobalMVN":
        stats = cuts.compute_global_feature_stats(max_cuts=max_cuts, extractor=extractor)
        stats = {name: torch.as_tensor(value) for name, value in stats.items()}
        (feature_dim,) = stats["norm_means"].shape
        mvn = cls(feature_dim)
        mvn.load_state_dict(stats)
        return mvn

    @classmethod
    def from_file(cls, stats_file) -> "HipGlobalMVN":
        stats = torch.load(stats_file)
        (feature_dim,) = stats["norm_means"].shape
        mvn = cls(feature_dim)
        mvn.load_state_dict(stats)
        return mvn

    def to_file(self, stats_file):
        torch.save(self.state_dict(), stats_file)

    def _run(self, features: torch.Tensor, inverse: int) -> torch.Tensor:
        x = _require_cuda(features, "HipGlobalMVN")
        if x.shape[-1] != self.feature_dim:
            raise RuntimeError(f"The size of tensor a ({x.shape[-1]}) must match the size of tensor b ({self.feature_dim}) at non-singleton dimension {x.ndim - 1}")
        means = self.norm_means.to(device=x.device, dtype=torch.float32).contiguous()
        stds = self.norm_stds.to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.load().check(
                "hipfeat_global_mvn", x.data_ptr(), out.data_ptr(), means.data_ptr(), stds.data_ptr(), x.numel() // self.feature_dim,
                self.feature_dim, inverse, int(torch.cuda.current_stream(x.device).cuda_stream),
            )
        return out

    def forward(self, features: torch.Tensor, supervision_segments: Optional[torch.IntTensor] = None) -> torch.Tensor:
        return self._run(features, 0)

    def inverse(self, features: torch.Tensor) -> torch.Tensor:
        return self._run(features, 1)


class HipSpecAugment(torch.nn.Module):
    """SpecAugment (time warp, feature masks, frame masks) for a ``(B, T, F)`` batch in two launches."""

    def __init__(
        self,
        time_warp_factor: Optional[int] = 80,
        num_feature_masks: int = 2,
        features_mask_size: int = 27,
        num_frame_masks: int = 10,
        frames_mask_size: int = 100,
        max_frames_mask_fraction: float = 0.15,
        p=0.9,
        fast_rng: bool = False,
    ):
        super().__init__()
        assert 0 <= p <= 1
        assert num_feature_masks >= 0
        assert num_frame_masks >= 0
        assert features_mask_size > 0
        assert frames_mask_size > 0
        self.time_warp_factor = time_warp_factor
        self.num_feature_masks = num_feature_masks
        self.features_mask_size = features_mask_size
        self.num_frame_masks = num_frame_masks
        self.frames_mask_size = frames_mask_size
        self.max_frames_mask_fraction = max_frames_mask_fraction
        self.p = p
        # extension (not part of the reference's state): draw all random numbers of a batch with a few vectorised numpy calls --
        # the same distributions, but NOT the reference's random streams; ~20x less host time per batch (0.8 ms -> 0.04 ms at B = 40)
        self.fast_rng = fast_rng

    def _draw_fast(self, batch: int, num_frames: int, feature_dim: int):
        """Vectorised equivalent of ``draw`` without supervision segments: same distributions as _forward_single (:217-266)."""
        rng = np.random.default_rng(np.random.randint(0, 2**31 - 1))  # follows numpy's global seed
        apply = rng.random(batch) <= self.p
        segs = []
        factor = self.time_warp_factor
        if factor is not None and factor >= 1 and num_frames - factor > factor + 1:
            center = rng.integers(factor + 1, num_frames - factor, size=batch)
            warped = rng.integers(center - factor, center + factor + 1)
            for b in np.nonzero(apply & (warped != center))[0]:
                segs.append((int(b), 0, num_frames, int(center[b]), int(warped[b])))
        masks = []

        def add(axis, size, mask_size, times):
            if times == 0:
                return
            values = rng.integers(0, int(mask_size), size=(batch, times))
            starts = (rng.random((batch, times), dtype=np.float32) * (size - values).astype(np.float32)).astype(np.int64)
            for b in np.nonzero(apply)[0]:
                for a, v in zip(starts[b].tolist(), values[b].tolist()):
                    lo, hi, _ = slice(a, a + v).indices(size)
                    masks.append((int(b), axis, lo, max(lo, hi)))

        add(2, feature_dim, self.features_mask_size, self.num_feature_masks)
        max_tot = self.max_frames_mask_fraction * num_frames
        n_frame_masks = min(self.num_frame_masks, math.ceil(max_tot / self.frames_mask_size))
        add(1, num_frames, min(self.frames_mask_size, max_tot // n_frame_masks), n_frame_masks)
        seg_rounds = [np.array(segs, dtype=_lib.WARP_SEGMENT_DTYPE)] if segs else []
        return seg_rounds, np.array(masks, dtype=_lib.MASK_DTYPE)

    # -- the reference's random decisions, call for call -------------------------------------------------------------
    def _draw_warp(self, t: int) -> Optional[Tuple[int, int]]:
        """time_warp (signal_transforms.py:338-353): the (center, warped) pair, or None when nothing is warped."""
        factor = self.time_warp_factor
        if factor is None or factor < 1:
            return None
        if t - factor <= factor + 1:
            return None
        center = np.random.randint(factor + 1, t - factor)
        warped = np.random.randint(center - factor, center + factor + 1)
        if warped == center:
            return None
        return int(center), int(warped)

    @staticmethod
    def _draw_masks(size: int, mask_size, mask_times: int) -> List[Tuple[int, int]]:
        """mask_along_axis_optimized (:297-335): [begin, end) of every mask along an axis of ``size`` entries."""
        values = torch.randint(int(0), int(mask_size), (1, mask_times)).numpy().reshape(-1)
        # torch.rand(1, n) * (size - values): float32 times int64 -> float32; .long() truncates.  Same arithmetic in numpy
        # (the generator calls above and below are what must stay torch's; the rest is host time, 60 sequences per batch)
        min_values = torch.rand(1, mask_times).numpy().reshape(-1) * (size - values).astype(np.float32)
        starts = min_values.astype(np.int64)
        # a mask wider than the axis (features_mask_size > F) makes `size - values` and hence the start negative; the
        # reference then slices with it (features[:, :, start:end], :316-332), so Python's slice rules decide the region
        out = []
        for a, b in zip(starts.tolist(), (starts + values).tolist()):
            lo, hi, _ = slice(a, b).indices(size)
            out.append((lo, max(lo, hi)))
        return out

    def _draw_single(self, t: int, f: int, warp: bool, mask: bool):
        """_forward_single (:217-266) for a (t, f) matrix -> ((center, warped) | None, [(axis, begin, end)])."""
        if random.random() > self.p:
            return None, []
        seg = self._draw_warp(t) if warp else None
        masks: List[Tuple[int, int, int]] = []
        if mask:
            masks += [(2, a, b) for a, b in self._draw_masks(f, self.features_mask_size, self.num_feature_masks)]
            max_tot_mask_frames = self.max_frames_mask_fraction * t
            num_frame_masks = min(self.num_frame_masks, math.ceil(max_tot_mask_frames / self.frames_mask_size))
            max_mask_frames = min(self.frames_mask_size, max_tot_mask_frames // num_frame_masks)
            masks += [(1, a, b) for a, b in self._draw_masks(t, max_mask_frames, num_frame_masks)]
        return seg, masks

    def draw(self, batch: int, num_frames: int, feature_dim: int, supervision_segments=None):
        """All random decisions of one ``forward`` call -> (rounds of warp segments, masks) as numpy records.  Warp
        segments of one sequence that overlap go to successive rounds (the reference applies them one after another)."""
        segs: List[Tuple[int, int, int, int, int]] = []
        masks: List[Tuple[int, int, int, int]] = []
        if supervision_segments is None:
            for b in range(batch):
                seg, ms = self._draw_single(num_frames, feature_dim, True, True)
                if seg is not None:
                    segs.append((b, 0, num_frames, seg[0], seg[1]))
                masks += [(b, ax, lo, hi) for ax, lo, hi in ms]
        else:
            rows = supervision_segments.tolist() if hasattr(supervision_segments, "tolist") else list(supervision_segments)
            for b, start, n in rows:
                b, start, n = int(b), int(start), int(n)
                lo, hi = self._slice_bounds(start, start + n, num_frames)
                seg, _ = self._draw_single(max(0, hi - lo), feature_dim, True, False)
                if seg is not None:
                    segs.append((b if b >= 0 else b + batch, lo, hi - lo, seg[0], seg[1]))
            for b in range(batch):
                _, ms = self._draw_single(num_frames, feature_dim, False, True)
                masks += [(b, ax, lo, hi) for ax, lo, hi in ms]
        rounds: List[List[Tuple[int, int, int, int, int]]] = []
        placed: List[Tuple[int, Tuple[int, int, int, int, int]]] = []
        for s in segs:
            r = 0
            for pr, q in placed:
                if q[0] == s[0] and q[1] < s[1] + s[2] and s[1] < q[1] + q[2]:
                    r = max(r, pr + 1)
            placed.append((r, s))
            while len(rounds) <= r:
                rounds.append([])
            rounds[r].append(s)
        seg_rounds = [np.array(r, dtype=_lib.WARP_SEGMENT_DTYPE) for r in rounds]
        return seg_rounds, np.array(masks, dtype=_lib.MASK_DTYPE)

    @staticmethod
    def _slice_bounds(start: int, end: int, size: int) -> Tuple[int, int]:
        lo, hi, _ = slice(start, end).indices(size)  # what features[b, start:end] selects
        return lo, max(lo, hi)

    def forward(self, features: torch.Tensor, supervision_segments: Optional[torch.IntTensor] = None, *args, **kwargs) -> torch.Tensor:
        assert len(features.shape) == 3, "SpecAugment only supports batches of single-channel feature matrices."
        x = _require_cuda(features, "HipSpecAugment")
        B, T, F = x.shape
        if self.fast_rng and supervision_segments is None:
            seg_rounds, masks = self._draw_fast(B, T, F)
        else:
            seg_rounds, masks = self.draw(B, T, F, supervision_segments)
        return apply_specaug(x, seg_rounds, masks)

    def state_dict(self, **kwargs) -> Dict[str, Any]:
        return dict(
            time_warp_factor=self.time_warp_factor,
            num_feature_masks=self.num_feature_masks,
            features_mask_size=self.features_mask_size,
            num_frame_masks=self.num_frame_masks,
            frames_mask_size=self.frames_mask_size,
            max_frames_mask_fraction=self.max_frames_mask_fraction,
            p=self.p,
        )

    def load_state_dict(self, state_dict: Dict[str, Any]):
        for k in ("time_warp_factor", "num_feature_masks", "features_mask_size", "num_frame_masks", "frames_mask_size", "max_frames_mask_fraction", "p"):
            setattr(self, k, state_dict.get(k, getattr(self, k)))


def apply_specaug(x: torch.Tensor, seg_rounds: List[np.ndarray], masks: np.ndarray) -> torch.Tensor:
    """Run ``hipfeat_specaug`` on a contiguous float32 ``(B, T, F)`` device tensor: one call per round of
    non-overlapping warp segments (almost always one), the masks in the last."""
    lib = _lib.load()
    B, T, F = x.shape
    rounds = list(seg_rounds) if len(seg_rounds) else [np.zeros(0, dtype=_lib.WARP_SEGMENT_DTYPE)]
    with torch.cuda.device(x.device):
        stream = int(torch.cuda.current_stream(x.device).cuda_stream)
        for i, segs in enumerate(rounds):
            last = i == len(rounds) - 1
            m = masks if last else masks[:0]
            segs = np.ascontiguousarray(segs)
            m = np.ascontiguousarray(m)
            out = torch.empty_like(x)
            lib.check("hipfeat_specaug", x.data_ptr(), out.data_ptr(), B, T, F, _lib.addr(segs) if len(segs) else None, len(segs),
                      _lib.addr(m) if len(m) else None, len(m), stream)
            x = out
    return x
