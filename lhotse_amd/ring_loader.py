"""
A batch loader whose worker processes write decoded audio straight into slots of ONE shared-memory ring (round 6).

Why: with real audio decoding in front of the GPU path, what bounds `compute_and_store_features_batch` is the transport of decoded audio out
of a `torch.utils.data.DataLoader`'s workers -- every batch becomes fresh shared-memory segments that are created, filled, passed by file
descriptor, mapped and unmapped again in the main process, and unpickled there by one interpreter: 3.8-4.1 k cuts/s with lhotse's
per-cut arrays, 5.5-6.4 k with one packed tensor per batch (profiles/r06_loader_transport_probe.txt; the consumer's `munmap` of one 38 MB
batch alone is 5.5 ms).  Here the memory is allocated ONCE: a worker takes a free slot, decodes its batch into it (packed, every cut on a
16-byte boundary, which is what the extractor's staging wants) and sends back a few hundred bytes (lengths + whatever small metadata the
loading function returns); the main process hands 1-D views of the slot to the extractor and gives the slot back when the library has
packed the batch.  No per-batch mapping, no descriptor passing, no large unpickling.

Every worker OWNS its slots (batch i goes to worker i mod W, into one of that worker's slots): a process pays a page fault for every
page of the ring it touches for the first time, and with a free-for-all pool every worker ends up touching every slot -- measured on the
GPU box (16-CPU cgroup quota) as 7 / 12 / 27 / 70 s of SYSTEM time for the same 12 800 cuts with 8 / 16 / 32 / 64 workers
(profiles/r06_host_limits.txt).  With owned slots a worker faults its own 3-5 slots once.

Nothing here knows lhotse: `load_batch(spec, out_bytes) -> (bytes_used, meta)` runs in the workers (lhotse_amd.storage supplies the one that
calls `cut.load_audio()` and serialises the manifest-line halves; tools/plumbing.py one that decodes WAV files).  Batches are delivered in
submission order.  Workers are started by `fork` unless this process already holds a live HIP context (lhotse_amd/_lib.py: the fork hazard),
then by a fork server -- the ring is NAMED shared memory, so either kind of worker attaches to it.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np

ALIGN = 16  # bytes: every cut of a slot starts on a 16-byte boundary


def _worker(shm_name: str, slot_bytes: int, num_slots: int, task_q, result_q, load_batch, init_fn, worker_id: int) -> None:
    from multiprocessing import shared_memory

    try:
        import torch

        torch.set_num_threads(1)  # as a DataLoader's worker does
    except Exception:  # noqa: BLE001
        pass
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        ring = np.ndarray((slot_bytes * num_slots,), dtype=np.uint8, buffer=shm.buf)
        if init_fn is not None:
            init_fn(worker_id)
        while True:
            task = task_q.get()
            if task is None:
                return
            idx, slot, spec = task
            try:
                used, meta = load_batch(spec, ring[slot * slot_bytes : (slot + 1) * slot_bytes])
                result_q.put((idx, slot, int(used), meta, None))
            except BaseException as e:  # noqa: BLE001 -- the error travels to the consumer, the worker lives on
                import traceback

                result_q.put((idx, slot, 0, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
            del spec
    finally:
        del ring
        shm.close()


class RingBatch:
    """One delivered batch: `data` = uint8 view of the used part of its slot (valid until `release()`), `meta` = what `load_batch` returned,
    `spec` = the object the batch was asked for with (it never left this process)."""

    __slots__ = ("index", "slot", "data", "meta", "spec", "_loader", "_released")

    def __init__(self, loader, index, slot, data, meta, spec=None):
        self._loader, self.index, self.slot, self.data, self.meta, self.spec, self._released = loader, index, slot, data, meta, spec, False

    def release(self) -> None:
        """The slot may be overwritten from now on (callable from any thread, once)."""
        if not self._released:
            self._released = True
            self.data = None
            self._loader._give_back(self.slot)

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001
            pass


class RingLoader:
    def __init__(self, load_batch: Callable[[Any, np.ndarray], Tuple[int, Any]], num_workers: int, slot_bytes: int, num_slots: Optional[int] = None,
                 start_method: Optional[str] = None, worker_init_fn: Optional[Callable[[int], None]] = None, preload: Iterable[str] = (),
                 consumer_holds: int = 12):
        """`num_slots` (total; rounded up to a whole number per worker) -- default: two per worker (one being filled, one finished) + its
        share of the `consumer_holds` slots the extractor / save threads keep at any time."""
        import multiprocessing as mp
        from multiprocessing import shared_memory

        assert num_workers >= 1 and slot_bytes > 0
        self.num_workers = W = int(num_workers)
        self.slot_bytes = (int(slot_bytes) + 4095) & ~4095
        per_worker = 2 + -(-int(consumer_holds) // W) if num_slots is None else max(1, -(-int(num_slots) // W))
        self.slots_per_worker = per_worker
        self.num_slots = per_worker * W
        if start_method is None:
            from . import _lib

            start_method = "forkserver" if _lib.hip_live() else "fork"
        self.start_method = start_method
        if start_method == "forkserver":
            # (the workers are forked off the SERVER: with the heavy imports done there once a worker starts in milliseconds)
            mp.set_forkserver_preload(["numpy", "torch", "lhotse_amd.ring_loader", *preload])
        ctx = mp.get_context(start_method)
        self._shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes * self.num_slots)
        self._ring = np.ndarray((self.slot_bytes * self.num_slots,), dtype=np.uint8, buffer=self._shm.buf)
        self._tasks = [ctx.Queue() for _ in range(W)]  # one per worker: batch i is loaded by worker i mod W into a slot that worker owns
        self._results = ctx.Queue()
        # worker w owns slots w * per_worker ... ; stacks: a released slot is the next one filled
        self._free: List[List[int]] = [list(range((w + 1) * per_worker - 1, w * per_worker - 1, -1)) for w in range(W)]
        self._freed = threading.Condition()
        self._procs = [ctx.Process(target=_worker, args=(self._shm.name, self.slot_bytes, self.num_slots, self._tasks[w], self._results, load_batch, worker_init_fn, w),
                                   daemon=True) for w in range(W)]
        for p in self._procs:
            p.start()
        self._closed = self._broken = False
        self._lock = threading.Lock()
        self._pin = None  # (lib, device index, queue of slots to page-lock, thread, {slot: registered?})

    # -- page-locking the ring for the GPU ----------------------------------------------------------------------------------------
    def pin_for(self, lib, device_index: int) -> None:
        """From now on every slot is page-locked for DMA by GPU `device_index` (libhipfeat's hipfeat_host_register) the first time a
        batch is delivered in it -- on a background thread, so that batch itself still takes the library's staging copy; from its second
        use on, the host pipeline uploads the slot's cuts straight out of the ring (no packing threads, no second copy of the audio).
        Call it once the GPU is in use (not before the workers are forked: ring_loader's own rule).  A refused registration (locked-memory
        limits) is remembered and the slot stays on the staging route."""
        if self._pin is not None or self._closed:
            return
        todo: "queue.SimpleQueue[Optional[int]]" = queue.SimpleQueue()
        state: Dict[int, bool] = {}
        base = self._ring.ctypes.data

        def work():
            while True:
                slot = todo.get()
                if slot is None:
                    return
                st = lib.raw("hipfeat_host_register", int(device_index), base + slot * self.slot_bytes, self.slot_bytes)
                state[slot] = st == 0

        th = threading.Thread(target=work, name="ring-pin", daemon=True)
        self._pin = (lib, int(device_index), todo, th, state)
        th.start()

    def pinned_slots(self) -> int:
        return 0 if self._pin is None else sum(1 for ok in self._pin[4].values() if ok)

    def _give_back(self, slot: int) -> None:
        with self._freed:
            self._free[slot // self.slots_per_worker].append(slot)
            self._freed.notify_all()

    # -- iteration ----------------------------------------------------------------------------------------------------------------
    def batches(self, specs: Iterable[Any]) -> Iterator[RingBatch]:
        """Load every spec of `specs` (in the workers, as far ahead as their slots allow) and yield the batches in submission order."""
        if self._broken:
            raise RuntimeError("ring loader: an earlier pass failed or was abandoned half-way (slots and results of it are still in flight): close this loader")
        self._broken = True  # (until this pass has delivered everything it handed out)
        it = iter(specs)
        W = self.num_workers
        submitted = delivered = 0
        exhausted = False
        done: Dict[int, Tuple[int, int, Any]] = {}
        asked: Dict[int, Any] = {}
        while True:
            # hand out work: batch i to worker i mod W while that worker has a slot (never blocks: a worker without one holds OLDER batches,
            # which are delivered first anyway)
            while not exhausted:
                w = submitted % W
                with self._freed:
                    slot = self._free[w].pop() if self._free[w] else None
                if slot is None:
                    break
                try:
                    spec = next(it)
                except StopIteration:
                    exhausted = True
                    self._give_back(slot)
                    break
                self._tasks[w].put((submitted, slot, spec))
                asked[submitted] = spec
                submitted += 1
            if delivered == submitted and exhausted:
                self._broken = False
                return
            if delivered in done:
                slot, used, meta = done.pop(delivered)
                if self._pin is not None and slot not in self._pin[4]:
                    self._pin[4][slot] = False  # (asked for; True once the pin thread has it registered)
                    self._pin[2].put(slot)
                yield RingBatch(self, delivered, slot, self._ring[slot * self.slot_bytes : slot * self.slot_bytes + used], meta, asked.pop(delivered))
                delivered += 1
                continue
            if delivered == submitted:  # nothing in flight and the next worker has no slot: wait for the consumer to release one of its
                with self._freed:
                    if not self._free[submitted % W]:
                        self._freed.wait(timeout=1.0)
                continue
            try:
                idx, slot, used, meta, err = self._results.get(timeout=1.0)
            except queue.Empty:
                dead = [p.pid for p in self._procs if not p.is_alive()]
                if dead:
                    raise RuntimeError(f"ring loader: worker process(es) {dead} died")
                continue
            if err is not None:
                raise RuntimeError(f"ring loader: loading batch {idx} failed in a worker:\n{err}")
            if used > self.slot_bytes:
                raise RuntimeError(f"ring loader: batch {idx} needs {used} bytes, a slot has {self.slot_bytes}")
            done[idx] = (slot, used, meta)

    # -- teardown -----------------------------------------------------------------------------------------------------------------
    def close(self) -> None:
        with self._lock:
            if self._closed:
                return
            self._closed = True
        for q in self._tasks:
            try:
                q.put(None)
            except Exception:  # noqa: BLE001
                pass
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if self._pin is not None:  # (the caller has collected every batch it submitted out of this ring: nothing reads it any more)
            lib, _, todo, th, state = self._pin
            todo.put(None)
            th.join(timeout=30)
            base = self._ring.ctypes.data
            for slot, ok in state.items():
                if ok:
                    lib.raw("hipfeat_host_unregister", base + slot * self.slot_bytes)
            self._pin = None
        self._ring = None
        try:
            self._shm.close()
        finally:
            try:
                self._shm.unlink()
            except FileNotFoundError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def pack_into(out: np.ndarray, arrays: List[np.ndarray]) -> Tuple[int, np.ndarray, np.ndarray]:
    """Copy 1-D arrays of one dtype into `out` (uint8), each on an ALIGN-byte boundary -> (bytes used, element offsets, lengths)."""
    item = arrays[0].dtype.itemsize
    per = ALIGN // item
    lens = np.array([a.shape[0] for a in arrays], dtype=np.int64)
    offs = np.zeros(len(arrays) + 1, dtype=np.int64)
    np.cumsum((lens + per - 1) // per * per, out=offs[1:])
    used = int(offs[-1]) * item
    if used > out.shape[0]:
        raise ValueError(f"batch of {used} bytes does not fit a ring slot of {out.shape[0]} bytes (raise slot_bytes / lower batch_duration)")
    flat = out[:used].view(arrays[0].dtype)
    for a, o, n in zip(arrays, offs, lens):
        flat[o : o + n] = a
    return used, offs[:-1].copy(), lens


class SlotWriter:
    """Appends 1-D arrays of ONE dtype to a slot as they are loaded, each on an ALIGN-byte boundary.  Copying every cut the moment it is
    decoded (instead of collecting the batch and packing it at the end) keeps ONE decoded array alive at a time: the allocator hands the
    same block out again and again, where 60 arrays alive at once are 60 fresh mappings per batch -- ~150 page faults per 10 s cut,
    measured as the larger half of a worker's time (tools/loader_worker_probe.py)."""

    def __init__(self, out: np.ndarray):
        self.out, self.used, self.dtype, self.offs, self.lens = out, 0, None, [], []

    def add(self, a: np.ndarray) -> bool:
        """-> False (nothing written) when `a` is of another dtype or does not fit any more."""
        if self.dtype is None:
            self.dtype = a.dtype
        elif a.dtype != self.dtype:
            return False
        item = self.dtype.itemsize
        nbytes = a.shape[0] * item
        if self.used + nbytes > self.out.shape[0]:
            return False
        self.out[self.used : self.used + nbytes].view(self.dtype)[:] = a
        self.offs.append(self.used // item)
        self.lens.append(a.shape[0])
        self.used = (self.used + nbytes + ALIGN - 1) & ~(ALIGN - 1)
        return True

    def arrays(self) -> List[np.ndarray]:
        """Copies of what was added (for a batch that turns out not to fit: it then travels as arrays)."""
        flat = self.out.view(self.dtype) if self.dtype is not None else None
        return [flat[o : o + n].copy() for o, n in zip(self.offs, self.lens)]

    def finish(self) -> Tuple[int, np.ndarray, np.ndarray]:
        """-> (bytes used, element offsets, lengths)"""
        return min(self.used, self.out.shape[0]), np.array(self.offs, dtype=np.int64), np.array(self.lens, dtype=np.int64)
