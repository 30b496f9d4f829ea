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
            self._loader._free.put(self.slot)

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001
            pass


class RingLoader:
    def __init__(self, load_batch: Callable[[Any, np.ndarray], Tuple[int, Any]], num_workers: int, slot_bytes: int, num_slots: Optional[int] = None,
                 start_method: Optional[str] = None, worker_init_fn: Optional[Callable[[int], None]] = None, preload: Iterable[str] = ()):
        import multiprocessing as mp
        from multiprocessing import shared_memory

        assert num_workers >= 1 and slot_bytes > 0
        self.num_workers = int(num_workers)
        self.slot_bytes = (int(slot_bytes) + 4095) & ~4095
        self.num_slots = int(num_slots or (2 * self.num_workers + 12))  # two per worker + what the extractor / save threads hold on to
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
        self._tasks, self._results = ctx.Queue(), ctx.Queue()
        self._free: "queue.LifoQueue[int]" = queue.LifoQueue()  # (a stack: a released slot is the next one filled -- the working set stays what is in flight)
        for s in reversed(range(self.num_slots)):
            self._free.put(s)
        self._procs = [ctx.Process(target=_worker, args=(self._shm.name, self.slot_bytes, self.num_slots, self._tasks, self._results, load_batch, worker_init_fn, w),
                                   daemon=True) for w in range(self.num_workers)]
        for p in self._procs:
            p.start()
        self._closed = False
        self._lock = threading.Lock()

    # -- iteration ----------------------------------------------------------------------------------------------------------------
    def batches(self, specs: Iterable[Any]) -> Iterator[RingBatch]:
        """Load every spec of `specs` (in the workers, up to one per free slot ahead) and yield the batches in submission order."""
        it = iter(specs)
        submitted = delivered = 0
        exhausted = False
        done: Dict[int, Tuple[int, int, Any]] = {}
        asked: Dict[int, Any] = {}
        while True:
            # hand out work while slots are free (never blocks: what is not free yet is picked up on a later turn)
            while not exhausted:
                try:
                    slot = self._free.get_nowait()
                except queue.Empty:
                    break
                try:
                    spec = next(it)
                except StopIteration:
                    exhausted = True
                    self._free.put(slot)
                    break
                self._tasks.put((submitted, slot, spec))
                asked[submitted] = spec
                submitted += 1
            if delivered == submitted and exhausted:
                return
            if delivered in done:
                slot, used, meta = done.pop(delivered)
                yield RingBatch(self, delivered, slot, self._ring[slot * self.slot_bytes : slot * self.slot_bytes + used], meta, asked.pop(delivered))
                delivered += 1
                continue
            if delivered == submitted:  # nothing in flight and no slot free: wait for the consumer to release one
                slot = self._free.get()
                self._free.put(slot)
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
        for _ in self._procs:
            try:
                self._tasks.put(None)
            except Exception:  # noqa: BLE001
                pass
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
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
