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


# ---- NUMA placement of a rank ----------------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    """"0-31,128-159" -> [0, ..., 31, 128, ..., 159] (the format of /sys/.../local_cpulist and node*/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_pci_address(device_index: int, sysfs: str = "/sys"):
    """PCI address "dddd:bb:dd.f" of HIP device `device_index` as this process sees it (visible-device masks applied by the runtime):
    torch's device properties first; else the KFD topology in enumeration order (GPU nodes only), which is HIP's order when no mask is set."""
    try:
        import torch

        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(device_index)
            if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
                return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        pass
    if any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return None  # the index no longer maps onto the topology order
    nodes = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    gpus = []
    try:
        for n in sorted(os.listdir(nodes), key=int):
            props = {}
            with open(os.path.join(nodes, n, "properties")) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    props[k] = v.strip()
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
                gpus.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    except (OSError, ValueError):
        return None
    return gpus[device_index] if device_index < len(gpus) else None


def bind_to_gpu_numa_node(device_index: int, sysfs: str = "/sys", apply: bool = True) -> dict:
    """Pin this process -- every existing thread, and so every thread started later: the packing pool, the save thread, torch's
    intra-op pool -- to the CPUs of the NUMA node GPU `device_index` is attached to.  Called before the first pinned allocation:
    under Linux's default local-allocation policy the staging buffers are then first-touched on that node, so H2D / D2H copies do not
    cross the inter-socket link and the ranks of a node spread over its memory controllers instead of all starting on node 0.
    (The reference has no counterpart: its workers are CPU-only, lhotse/cut/set.py:2141-2195; 8 GPU ranks with pinned staging are
    where placement starts to matter.)

    Never raises: returns {"bound": bool, "node": int | None, "cpus": n, "pci": str | None, "why": str} for the caller's log.
    HIPFEAT_NUMA_BIND=0 switches it off."""
    info = {"bound": False, "node": None, "cpus": 0, "pci": None, "why": ""}
    if os.environ.get("HIPFEAT_NUMA_BIND", "1") in ("0", "off", "false"):
        info["why"] = "HIPFEAT_NUMA_BIND=0"
        return info
    if not hasattr(os, "sched_setaffinity"):
        info["why"] = "no sched_setaffinity on this platform"
        return info
    pci = _gpu_pci_address(device_index, sysfs)
    info["pci"] = pci
    if pci is None:
        info["why"] = "PCI address of the GPU unknown"
        return info
    base = os.path.join(sysfs, "bus/pci/devices", pci)
    try:
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        info["why"] = f"{base}/numa_node unreadable"
        return info
    info["node"] = node
    if node < 0:
        info["why"] = "the platform reports no NUMA affinity for this GPU (numa_node = -1: single node, or a VM without topology)"
        return info
    cpus: List[int] = []
    for path in (os.path.join(sysfs, f"devices/system/node/node{node}/cpulist"), os.path.join(base, "local_cpulist")):
        try:
            with open(path) as f:
                cpus = _parse_cpulist(f.read())
            if cpus:
                break
        except (OSError, ValueError):
            continue
    allowed = set(os.sched_getaffinity(0))
    cpus = sorted(c for c in cpus if c in allowed)  # never widen a cgroup / taskset restriction
    if not cpus:
        info["why"] = f"node {node} has no CPU this process may run on"
        return info
    info["cpus"] = len(cpus)
    if not apply:
        info["why"] = "dry run"
        return info
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids or [0]:
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:
            pass  # a thread that has just exited
    info["bound"] = True
    info["why"] = f"{len(tids)} thread(s) pinned to the {len(cpus)} CPUs of NUMA node {node} (GPU {device_index} at {pci})"
    return info


# ---- the sharded extraction driver --------------------------------------------------------------------------------------
class _Rendezvous:
    """The two barriers of the sharded driver (everybody has extracted / rank 0 has combined).  In order of preference: the caller's
    process group (RCCL or gloo); a gloo group created here from torchrun's MASTER_ADDR / MASTER_PORT (host-side only -- the barrier
    carries no data, so it does not need RCCL) and destroyed again; marker files next to the shards for ranks that were only given
    RANK / WORLD_SIZE.

    Marker-file mode.  File names carry a run token shared by the ranks of one launch (HIPFEAT_RUN_ID, else torchrun's
    run id, else the parent process id) -- but a token can repeat: ranks started by hand from ONE shell share its pid across runs, which
    is exactly the advertised per-shard resume.  The barriers are therefore keyed by a NONCE that is agreed per launch and cannot be
    satisfied by leftovers: every rank publishes a fresh random id in its ``hello`` file; rank 0 draws the nonce and publishes it in
    the ``nonce`` file together with the hello ids it has seen (re-published from inside its barrier loops whenever a hello changes);
    rank r adopts the nonce only from a nonce file that quotes ITS current id.  A stale nonce / hello / barrier marker of an earlier run
    quotes other ids and is ignored; every rank sweeps its own leftovers of the same token when it starts, and its own files of this
    launch that nobody can be waiting for when it finishes.  A rank that fails leaves a ``failed`` marker, so that the others stop
    waiting at once instead of after the timeout; it counts only if it is not older than the waiting rank's own hello file, so the
    marker of a failed earlier launch is inert (and swept by its rank when that starts again)."""

    _calls = 0  # rendezvous created in this process (the same number on every rank: they all make the same calls)

    def __init__(self, marker_dir, rank: int, world: int, timeout: float, device_index=None, seq: int = None):
        self.dir, self.rank, self.world, self.timeout = marker_dir, rank, world, timeout
        self.dist = None
        self.owns_group = False
        self.device_index = device_index
        self.token = os.environ.get("HIPFEAT_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or f"ppid{os.getppid()}"
        self.nonce = None
        if world == 1:
            return
        _Rendezvous._calls += 1
        self.seq = _Rendezvous._calls if seq is None else seq  # part of the marker names and of the store's port: a second call never meets the first one's leftovers
        try:
            import datetime

            import torch.distributed as dist

            if dist.is_available():
                if dist.is_initialized():
                    self.dist = dist
                elif "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ:
                    # a store of its own (rank 0 hosts it above MASTER_PORT, a fresh port per call): the launcher's agent store keeps the
                    # keys of earlier groups, which a second call in the same processes would trip over
                    store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1 + self.seq, world,
                                          is_master=(rank == 0), timeout=datetime.timedelta(seconds=min(timeout, 600.0)), wait_for_workers=False)
                    dist.init_process_group("gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout))
                    self.dist, self.owns_group = dist, True
        except ImportError:  # pragma: no cover
            pass
        if self.dist is None:
            self._marker_mode_hello()

    # ---- marker files ------------------------------------------------------------------------------------------------------------
    def _path(self, tag: str, rank=None, nonce: str = None):
        from pathlib import Path

        parts = [f".{tag}", self.token, str(self.seq)] + ([nonce] if nonce else []) + ([str(rank)] if rank is not None else [])
        return Path(self.dir) / "-".join(parts)

    @staticmethod
    def _write_atomic(path, text: str) -> None:
        tmp = path.with_name(path.name + f".tmp{os.getpid()}")
        tmp.write_text(text)
        os.replace(tmp, path)

    @staticmethod
    def _read(path):
        try:
            return path.read_text()
        except OSError:
            return None

    def _marker_mode_hello(self) -> None:
        import glob
        import uuid

        # this rank's leftovers of earlier launches with the same token (any call number, any nonce): nobody of THIS launch waits for
        # them -- except the files of the previous call in this very process, which a slower rank may still be polling
        for tag in ("hello", "extracted", "combined", "failed"):
            keep = f".{tag}-{self.token}-{self.seq - 1}-"
            for f in glob.glob(os.path.join(glob.escape(str(self.dir)), f".{tag}-{glob.escape(self.token)}-*-{self.rank}")):
                if os.path.basename(f).startswith(keep):
                    continue
                try:
                    os.unlink(f)
                except OSError:
                    pass
        self.my_id = uuid.uuid4().hex
        if self.rank == 0:
            for f in glob.glob(os.path.join(glob.escape(str(self.dir)), f".nonce-{glob.escape(self.token)}-*")):
                if os.path.basename(f) == f".nonce-{self.token}-{self.seq - 1}":
                    continue
                try:
                    os.unlink(f)
                except OSError:
                    pass
            self.nonce = uuid.uuid4().hex
            self._published = None
        self._write_atomic(self._path("hello", self.rank), self.my_id)
        if self.rank == 0:
            self._publish()

    def _hello_ids(self):
        return {r: self._read(self._path("hello", r)) for r in range(self.world)}

    def _publish(self) -> None:
        """Rank 0: (re-)publish the nonce with the hello ids currently on disk."""
        import json

        ids = {str(r): i for r, i in self._hello_ids().items() if i}
        if ids != self._published:
            self._write_atomic(self._path("nonce"), json.dumps({"nonce": self.nonce, "ids": ids}))
            self._published = ids

    def _resolve(self) -> bool:
        """Rank r > 0: adopt the nonce once rank 0 has quoted this rank's current id next to it."""
        import json

        if self.nonce is not None:
            return True
        raw = self._read(self._path("nonce"))
        if raw:
            try:
                doc = json.loads(raw)
            except ValueError:
                return False
            if doc.get("ids", {}).get(str(self.rank)) == self.my_id:
                self.nonce = doc["nonce"]
                return True
        return False

    def _failed_ranks(self):
        """Ranks whose `failed` marker is not older than this rank's own hello file (both times come from the file system that holds
        them): the marker of an earlier launch with the same token is older and does not count.  (A rank that fails before this one
        has even started is therefore not noticed, and this rank waits for it until the timeout.)"""
        try:
            born = os.stat(self._path("hello", self.rank)).st_mtime_ns
        except OSError:
            return []
        out = []
        for r in range(self.world):
            try:
                if r != self.rank and os.stat(self._path("failed", r)).st_mtime_ns >= born:
                    out.append(r)
            except OSError:
                pass
        return out

    def barrier(self, tag: str) -> None:
        import time

        if self.world == 1:
            return
        if self.dist is not None:
            if self.dist.get_backend() == "nccl" and self.device_index is not None:
                self.dist.barrier(device_ids=[self.device_index])  # RCCL must not guess this rank's GPU
            else:
                self.dist.barrier()
            return
        deadline = time.time() + self.timeout
        written = False
        while True:
            if self.rank == 0:
                self._publish()
            if self._resolve():
                if not written:
                    self._path(tag, self.rank, self.nonce).write_text("done")
                    written = True
                missing = [r for r in range(self.world) if not self._path(tag, r, self.nonce).exists()]
                if not missing:
                    return
            else:
                missing = [0]
            failed = self._failed_ranks()
            if failed:
                raise RuntimeError(f"sharded extraction: ranks {failed} failed before '{tag}' (see their own tracebacks)")
            if time.time() > deadline:
                raise TimeoutError(f"sharded extraction: ranks {missing} did not reach '{tag}' within {self.timeout:.0f} s")
            time.sleep(0.02)

    def failed(self) -> None:
        """Called on the way out of a failing rank (marker-file mode: process groups have their own timeouts)."""
        if self.world > 1 and self.dist is None:
            try:
                self._write_atomic(self._path("failed", self.rank), self.my_id)
            except OSError:  # pragma: no cover
                pass

    def close(self) -> None:
        if self.owns_group:
            self.dist.destroy_process_group()
            self.owns_group = False
        if self.world > 1 and self.dist is None and self.nonce is not None:
            # this rank's marker of the first barrier: everybody is past it once "combined" was reached.  The "combined" marker, the hello
            # and (rank 0) the nonce file stay -- a slower rank may still be polling them -- and are swept by the next launch.
            try:
                self._path("extracted", self.rank, self.nonce).unlink()
            except OSError:
                pass


class _OwnedCuts:
    """The cuts of `cuts` whose input index `owner_of` assigns to `rank`, lazily (one pass over the source per iteration)."""

    def __init__(self, cuts, owner_of, rank: int):
        self.cuts, self.owner_of, self.rank = cuts, owner_of, rank

    def __iter__(self):
        owner_of, rank = self.owner_of, self.rank
        return (c for i, c in enumerate(self.cuts) if owner_of[i] == rank)

    def __len__(self) -> int:
        return int((self.owner_of == self.rank).sum())


def shard_paths(storage_path, manifest_path, rank: int):
    """(storage, manifest) of one rank: `<storage_path>/feats-<rank>` as the reference names its per-job storages
    (lhotse/cut/set.py:2141-2153) and `cuts-<rank>.jsonl.gz` next to the combined manifest."""
    from pathlib import Path

    sp = Path(storage_path)
    mp = Path(manifest_path)
    stem = mp.name
    for ext in (".jsonl.gz", ".jsonl", ".json.gz", ".json", ".yaml.gz", ".yaml", ".yml"):  # only the manifest extension goes:
        if stem.endswith(ext):  # "cuts.train.jsonl.gz" and "cuts.dev.jsonl.gz" in one directory must not share shard manifests
            stem = stem[: -len(ext)]
            break
    return sp / f"feats-{rank}", mp.parent / f"{stem}-{rank}.jsonl.gz"


def combine_shard_manifests(cuts, manifest_path, shard_manifests: Sequence, owner) -> "object":
    """Merge the per-rank manifests into ONE manifest in the order of `cuts` (the reference's `combine` concatenates job after job,
    lhotse/cut/set.py:2194; restoring the input order makes the result independent of the number of GPUs).  Streaming: every shard
    manifest is read once, in step with one pass over the input ids; cuts a rank dropped (audio that failed to load) are skipped.
    `owner(i, cut)` = rank that owned input cut i."""
    from lhotse import CutSet
    from lhotse.serialization import load_manifest_lazy

    readers = [iter(load_manifest_lazy(p)) for p in shard_manifests]
    heads = [next(r, None) for r in readers]
    with CutSet.open_writer(manifest_path, overwrite=True) as w:
        for i, cut in enumerate(cuts):
            r = owner(i, cut)
            h = heads[r]
            if h is not None and h.id == cut.id:
                w.write(h)
                heads[r] = next(readers[r], None)
    left = [h.id for h in heads if h is not None]
    if left:
        raise RuntimeError(f"sharded extraction: shard manifests hold cuts that are not in the input, or are out of order: {left[:3]}")
    return w.open_manifest()


def compute_and_store_features_sharded(
    cuts,
    extractor,
    storage_path,
    manifest_path,
    batch_duration: float = 600.0,
    num_workers: int = 4,
    collate: bool = False,
    augment_fn=None,
    storage_type=None,
    overwrite: bool = False,
    balance: str = "round_robin",
    rank: int = None,
    world: int = None,
    barrier_timeout: float = 3600.0,
    numa_bind: bool = True,
    archive_stripes: int = 1,
    loader: str = None,
    loader_start_method: str = None,
    worker_init_fn=None,
    wav_pcm16: bool = False,
):
    """Feature extraction of one CutSet over the GPUs of a node: the multi-GPU form of ``compute_and_store_features_batch``.

    Every rank calls this with the SAME arguments (one process per GPU: ``torchrun --nproc-per-node 8 script.py``).  Rank *r* of *W*
    (``rank_and_world()``: process group, else RANK / WORLD_SIZE) takes its shard of the cuts -- ``balance="round_robin"``: cuts
    r, r+W, ... exactly as the reference's ``CutSet(LazySlicer(self.data, k=i, n=num_jobs))`` (lhotse/cut/set.py:2158-2160), lazily;
    ``balance="duration"``: duration-balanced shards for mixed-length corpora (needs one pass over the durations) -- runs the bulk
    batch driver on GPU ``LOCAL_RANK`` into its own storage ``<storage_path>/feats-r`` and manifest ``cuts-r.jsonl.gz`` (per-shard
    resume included: an interrupted run continues where each rank stopped), and after a barrier rank 0 merges the shard manifests into
    ``manifest_path`` in input order.  NO collective touches the data path; the barrier is the only communication.

    ``archive_stripes``: files per rank's ``hip_archive`` (``feats-r.hfa``, ``feats-r.1.hfa``, ...; see ``compute_and_store_features_batch``).
    ``loader`` / ``loader_start_method`` / ``worker_init_fn`` / ``wav_pcm16``: every rank's loader, as in ``compute_and_store_features_batch`` (the
    shared-memory ring by default where it applies).

    ``numa_bind`` (default on, for world > 1): before the extractor touches its GPU -- i.e. before any pinned staging buffer exists and
    before the packing / save threads start -- the rank is pinned to the CPUs of its GPU's NUMA node (``bind_to_gpu_numa_node``;
    a no-op with a logged reason where the platform reports no topology).  ``HIPFEAT_NUMA_BIND=0`` overrides.

    Returns the combined CutSet on rank 0 and the rank's own shard CutSet elsewhere."""
    from pathlib import Path

    from lhotse import CutSet
    from lhotse.lazy import LazySlicer

    from .storage import compute_and_store_features_batch

    if rank is None or world is None:
        rank, world = rank_and_world()
    if balance not in ("round_robin", "duration"):
        raise ValueError(f"balance must be 'round_robin' or 'duration', got {balance!r}")
    if manifest_path is None:
        raise ValueError("sharded extraction needs a manifest_path: the shards meet on disk")
    manifest_path = Path(manifest_path)
    storage_path = Path(storage_path)
    storage_path.mkdir(parents=True, exist_ok=True)
    manifest_path.parent.mkdir(parents=True, exist_ok=True)
    # one GPU per rank: a bare "cuda" device becomes this rank's GPU (LOCAL_RANK as torchrun sets it)
    dev = str(getattr(extractor.config, "device", "cuda"))
    device_index = None
    if dev == "cuda" and world > 1:
        device_index = int(os.environ.get("LOCAL_RANK", rank))
        try:
            import torch

            if torch.cuda.is_available():  # a launcher may narrow the visible devices per rank
                device_index %= torch.cuda.device_count()
        except ImportError:  # pragma: no cover
            pass
    elif dev.startswith("cuda:"):
        device_index = int(dev.split(":")[1])
    if numa_bind and world > 1 and device_index is not None:
        import logging

        logging.getLogger(__name__).info("rank %d: NUMA placement: %s", rank, bind_to_gpu_numa_node(device_index))
    if dev == "cuda" and world > 1:
        extractor.to(f"cuda:{device_index}")

    if balance == "round_robin":
        mine = CutSet(LazySlicer(cuts.data, k=rank, n=world)) if world > 1 else cuts

        def owner(i, cut):
            return i % world

    else:
        # one pass over the durations (nothing else of a cut is kept), then the rank's cuts are yielded lazily by a second pass when the
        # batch driver iterates -- the corpus is never materialised (it used to be, twice per rank)
        import numpy as np

        parts = shard_by_duration([c.duration for c in cuts], world)
        owner_of = np.zeros(sum(len(ix) for ix in parts), dtype=np.int32)
        for r, idx in enumerate(parts):
            owner_of[idx] = r
        mine = CutSet(_OwnedCuts(cuts, owner_of, rank)) if world > 1 else cuts

        def owner(i, cut):
            return int(owner_of[i])

    sub_storage, sub_manifest = shard_paths(storage_path, manifest_path, rank)
    meet = _Rendezvous(manifest_path.parent, rank, world, barrier_timeout, device_index)
    try:
        out = compute_and_store_features_batch(mine, extractor, sub_storage, manifest_path=sub_manifest, batch_duration=batch_duration,
                                               num_workers=num_workers, collate=collate, augment_fn=augment_fn, storage_type=storage_type,
                                               overwrite=overwrite, archive_stripes=archive_stripes, loader=loader,
                                               loader_start_method=loader_start_method, worker_init_fn=worker_init_fn, wav_pcm16=wav_pcm16)
        meet.barrier("extracted")
        if rank == 0:
            out = combine_shard_manifests(cuts, manifest_path, [shard_paths(storage_path, manifest_path, r)[1] for r in range(world)], owner)
        meet.barrier("combined")
    except BaseException:
        meet.failed()
        raise
    finally:
        meet.close()
    return out
