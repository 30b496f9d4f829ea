"""CPU: lhotse_amd/ring_loader.py on its own -- order, slot ownership, back-pressure, several passes, start methods, failures.  (Behind the
batch driver under the real lhotse: tests/test_lhotse_dropin.py; on the real plan with page-locked slots: tests/test_gpu_host_register.py.)"""
import os
import threading
import time

import numpy as np
import pytest

from lhotse_amd.ring_loader import RingLoader, SlotWriter


class Load:
    """spec = (seed, [lengths...], sleep_ms): float32 noise per length, cut by cut into the slot; meta carries the worker's pid."""

    def __call__(self, spec, out):
        seed, lens, sleep_ms = spec
        if sleep_ms:
            time.sleep(sleep_ms * 1e-3)
        rs = np.random.RandomState(seed)
        w = SlotWriter(out)
        for n in lens:
            assert w.add(rs.rand(n).astype(np.float32))
        used, offs, ln = w.finish()
        return used, {"offs": offs, "lens": ln, "pid": os.getpid(), "seed": seed}


def _check(rb, spec):
    seed, lens, _ = spec
    assert rb.meta["seed"] == seed and rb.spec is spec
    rs = np.random.RandomState(seed)
    flat = rb.data.view(np.float32)
    for o, n, want in zip(rb.meta["offs"].tolist(), rb.meta["lens"].tolist(), lens):
        assert n == want and o % 4 == 0 and np.array_equal(flat[o : o + n], rs.rand(n).astype(np.float32))


@pytest.mark.parametrize("start", ["fork", "forkserver"])
def test_order_ownership_and_several_passes(start):
    rng = np.random.RandomState(1)
    specs = [(s, rng.randint(1, 3000, size=rng.randint(1, 6)).tolist(), int(rng.randint(0, 3))) for s in range(120)]
    with RingLoader(Load(), num_workers=3, slot_bytes=1 << 16, num_slots=9, start_method=start) as rl:
        assert rl.num_slots == 9 and rl.slots_per_worker == 3 and rl.start_method == start
        for _ in range(2):  # the same loader serves pass after pass
            owner = {}
            held = []
            for i, rb in enumerate(rl.batches(specs)):
                assert rb.index == i
                _check(rb, specs[i])
                # batch i is loaded by worker i mod W into one of THAT worker's slots: a slot is only ever written by one process
                assert rb.slot // rl.slots_per_worker == i % 3
                assert owner.setdefault(rb.slot, rb.meta["pid"]) == rb.meta["pid"]
                held.append(rb)
                if len(held) > 4:  # a consumer with a lag of four batches
                    held.pop(0).release()
            for rb in held:
                rb.release()
            assert len({v for v in owner.values()}) == 3
        with pytest.raises(RuntimeError, match="handed out|abandoned|failed"):
            it = rl.batches(specs)
            next(it)
            it.close()  # abandoned half-way: the loader refuses another pass (slots / results of the old one are in flight)
            next(rl.batches(specs))


def test_back_pressure_from_another_thread_and_worker_init():
    """The consumer releases slots from a SAVE thread (as the batch driver does); a consumer slower than the workers never loses a batch
    and never sees a slot overwritten while it holds it; worker_init_fn runs in every worker."""
    marks = []

    def init(worker_id):  # (forked workers: a closure is fine)
        os.environ["RING_TEST_WORKER"] = str(worker_id)

    class LoadEnv(Load):
        def __call__(self, spec, out):
            used, meta = super().__call__(spec, out)
            meta["env"] = os.environ.get("RING_TEST_WORKER")
            return used, meta

    specs = [(s, [2000, 100 + s], 0) for s in range(60)]
    done = []
    lock = threading.Lock()

    def save(rb, i):
        time.sleep(0.004)  # slower than the workers
        _check(rb, specs[i])  # still intact when the save thread gets to it
        with lock:
            done.append(i)
        rb.release()

    from concurrent.futures import ThreadPoolExecutor

    with RingLoader(LoadEnv(), num_workers=2, slot_bytes=1 << 15, num_slots=4, start_method="fork", worker_init_fn=init) as rl, ThreadPoolExecutor(max_workers=1) as pool:
        futs = []
        for i, rb in enumerate(rl.batches(specs)):
            assert rb.meta["env"] == str(i % 2)
            futs.append(pool.submit(save, rb, i))
        for f in futs:
            f.result()
    assert done == list(range(60))


def test_failures_are_reported_not_hung():
    class Bad(Load):
        def __call__(self, spec, out):
            if spec[0] == 7:
                raise ValueError("bad cut")
            if spec[0] == 1000:
                os._exit(3)  # the worker process dies
            return super().__call__(spec, out)

    with RingLoader(Bad(), num_workers=2, slot_bytes=1 << 15, start_method="fork") as rl:
        with pytest.raises(RuntimeError, match="bad cut"):
            for rb in rl.batches([(s, [10], 0) for s in range(20)]):
                rb.release()
    with RingLoader(Bad(), num_workers=2, slot_bytes=1 << 15, start_method="fork") as rl:
        t0 = time.time()
        with pytest.raises(RuntimeError, match="died"):
            for rb in rl.batches([(0, [10], 0), (1000, [10], 0), (2, [10], 0)]):
                rb.release()
        assert time.time() - t0 < 20
    # a batch larger than its slot is the loading function's business (SlotWriter.add says no); close() is idempotent and frees the segment
    rl = RingLoader(Load(), num_workers=1, slot_bytes=4096, start_method="fork")
    name = rl._shm.name
    with pytest.raises(RuntimeError, match="AssertionError"):
        list(rl.batches([(0, [5000], 0)]))
    rl.close()
    rl.close()
    from multiprocessing import shared_memory

    with pytest.raises(FileNotFoundError):
        shared_memory.SharedMemory(name=name)
