"""
Embarrassingly parallel sharding of cuts over the GPUs of one node.

The reference shards feature extraction the same way on CPU workers:
``CutSet(LazySlicer(self.data, k=i, n=num_jobs))`` -- worker i takes items i, i+n, i+2n, ...
and writes its own ``feats-{i}`` storage; manifests are combined at the end
(lhotse/cut/set.py:2141-2160, :2194).  Rank / world size discovery mirrors the samplers
(lhotse/dataset/sampling/base.py:143-163): an initialised process group wins, then the
RANK / WORLD_SIZE environment variables, then (0, 1).

There is NO collective on the data path.  The only communication is optional bookkeeping
(cut counts, elapsed time) through ``torch.distributed`` -- RCCL on GPUs, gloo on CPU.
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Sequence, Tuple, TypeVar

T = TypeVar("T")


def rank_and_world() -> Tuple[int, int]:
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:  # pragma: no cover
        pass
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_indices(num_items: int, rank: int, world: int) -> range:
    """Indices of the items rank ``rank`` of ``world`` owns (round-robin, as LazySlicer(k, n))."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0, {world})")
    return range(rank, num_items, world)


def shard(items: Sequence[T], rank: int, world: int) -> List[T]:
    return [items[i] for i in shard_indices(len(items), rank, world)]


def shard_iter(items: Iterable[T], rank: int, world: int) -> Iterator[T]:
    """Lazy variant for manifests that are streamed rather than indexed."""
    for i, it in enumerate(items):
        if i % world == rank:
            yield it


def shard_by_duration(durations: Sequence[float], world: int) -> List[List[int]]:
    """Duration-balanced alternative for mixed-length corpora: longest
    first, each cut to the currently lightest rank.  Deterministic; returns the index lists."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += durations[i]
    for lst in out:
        lst.sort()
    return out


def all_reduce_stats(num_cuts: int, elapsed: float, device=None) -> Tuple[int, float]:
    """(total cuts over all ranks, max elapsed over ranks); a no-op without a process group."""
    try:
        import torch
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([float(num_cuts)], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            e = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
            return int(round(t.item())), float(e.item())
    except ImportError:  # pragma: no cover
        pass
    return num_cuts, elapsed
