"""GPU: ABI v5 -- uploads straight out of page-locked memory of the CALLER (hipfeat_host_register), the route the shared-memory ring
loader takes once its slots are registered.  The staging route (hipfeat_host_pipeline_submit's packing threads) is the reference here:
a batch that goes the direct way must give the same bits.  lhotse's side of this is the batch driver's main loop
(lhotse/cut/set.py:2374-2398); the real-lhotse equivalence of the ring loader is in tests/test_lhotse_dropin.py."""
import mmap

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import _lib
from lhotse_amd.ring_loader import RingLoader, SlotWriter

pytestmark = pytest.mark.gpu


def _page_aligned(nbytes):
    m = mmap.mmap(-1, nbytes)  # anonymous, page-aligned
    return m, np.frombuffer(m, dtype=np.uint8)


def _direct(ex):
    return int(ex.plan.lib.raw("hipfeat_host_pipeline_direct_batches", ex._native_pipe().handle))


@pytest.mark.parametrize("pcm16,half", [(False, False), (True, False), (False, True), (True, True)])
def test_batches_in_registered_memory_are_uploaded_from_there_and_give_the_same_bits(pcm16, half):
    rng = np.random.RandomState(7)
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    lens = [16000, 12345, 48000, 8001, 160000, 4000]
    dt = np.int16 if pcm16 else np.float32
    waves = [(rng.randint(-20000, 20000, size=n).astype(np.int16) if pcm16 else (rng.rand(n).astype(np.float32) - 0.5)) for n in lens]
    m, buf = _page_aligned(1 << 21)
    w = SlotWriter(buf)
    for a in waves:
        assert w.add(a)
    used, offs, ln = w.finish()
    views = [buf[:used].view(dt)[o : o + n] for o, n in zip(offs.tolist(), ln.tolist())]
    # staging route first (nothing registered yet)
    p = ex.submit_host_items(views, 16000, half=half)
    want = p.wait().copy()
    p.release()
    assert _direct(ex) == 0
    lib = ex.plan.lib
    lib.check("hipfeat_host_register", 0, buf.ctypes.data, buf.shape[0])
    try:
        with pytest.raises(_lib.HipFeatError, match="registered already"):
            lib.check("hipfeat_host_register", 0, buf.ctypes.data, buf.shape[0])
        for k in range(3):
            p = ex.submit_host_items(views, 16000, half=half)
            got = p.wait().copy()
            p.release()
            assert np.array_equal(got, want)
        assert _direct(ex) == 3
        # the same cuts handed over in another order are not back to back: staging route, same rows in that order
        order = [3, 0, 5, 1, 2, 4]
        p = ex.submit_host_items([views[i] for i in order], 16000, half=half)
        got = p.wait().copy()
        p.release()
        assert _direct(ex) == 3
        rows = np.concatenate([[0], np.cumsum([(n + 80) // 160 for n in lens])])
        assert np.array_equal(got, np.concatenate([want[rows[i] : rows[i + 1]] for i in order]))
        # a batch that only partly lies in the registered range: staging route
        outside = waves[0].copy()
        p = ex.submit_host_items([views[0], outside], 16000, half=half)
        got = p.wait().copy()
        p.release()
        assert _direct(ex) == 3 and np.array_equal(got[:100], want[:100]) and np.array_equal(got[100:], want[:100])
    finally:
        lib.check("hipfeat_host_unregister", buf.ctypes.data)
    with pytest.raises(_lib.HipFeatError, match="not registered"):
        lib.check("hipfeat_host_unregister", buf.ctypes.data)
    # unregistered again: the staging route, same bits
    p = ex.submit_host_items(views, 16000, half=half)
    got = p.wait().copy()
    p.release()
    assert _direct(ex) == 3 and np.array_equal(got, want)
    if not half and not pcm16:
        one = ex.extract(torch.from_numpy(waves[1]), 16000)
        assert np.array_equal(want[100 : 100 + one.shape[0]], one.cpu().numpy() if isinstance(one, torch.Tensor) else one)
    del views, buf, w
    m.close()


class _Noise:
    """load_batch of the ring loader: spec = (seed, [lengths]) -> float32 noise, cut by cut into the slot."""

    def __call__(self, spec, out):
        seed, lens = spec
        rs = np.random.RandomState(seed)
        w = SlotWriter(out)
        for n in lens:
            assert w.add(rs.rand(n).astype(np.float32) - 0.5)
        used, offs, ln = w.finish()
        return used, {"offs": offs, "lens": ln}


def test_ring_loader_with_page_locked_slots_feeds_the_host_pipeline():
    """Worker processes fill the slots; the slots are page-locked as they come into use; from its second use on a slot's batch is uploaded
    straight out of the ring.  Every batch equals the extractor's own per-cut result bit for bit, whichever route it took."""
    specs = [(s, [16000 + 160 * s, 32000, 8000 + s]) for s in range(40)]
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    with RingLoader(_Noise(), num_workers=2, slot_bytes=1 << 19, num_slots=4, start_method="forkserver") as rl:
        held = []
        for i, rb in enumerate(rl.batches(specs)):
            flat = rb.data.view(np.float32)
            views = [flat[o : o + n] for o, n in zip(rb.meta["offs"].tolist(), rb.meta["lens"].tolist())]
            p = ex.submit_host_items(views, 16000)
            rl.pin_for(ex.plan.lib, 0)
            held.append((i, rb, p))
            while len(held) > 2:
                k, b, q = held.pop(0)
                got = q.wait().copy()
                q.release()
                b.release()
                rs = np.random.RandomState(specs[k][0])
                want = np.concatenate([ex.extract(torch.from_numpy(rs.rand(n).astype(np.float32) - 0.5), 16000).cpu().numpy() for n in specs[k][1]])
                assert np.array_equal(got, want), k
        for k, b, q in held:
            q.wait()
            q.release()
            b.release()
        assert rl.pinned_slots() >= 2
        assert _direct(ex) >= 20  # (every slot's first batch went through staging, most of the rest straight out of the ring)


class _NoiseThenBoom(_Noise):
    def __call__(self, spec, out):
        if spec[0] == 25:
            raise ValueError("boom")
        return super().__call__(spec, out)


def test_a_failed_run_drains_the_pipeline_before_the_page_locked_ring_is_unmapped():
    """A worker fails half-way through: batches submitted out of page-locked slots are still queued / in flight when the consumer gives up.
    NativeHostPipeline.drain() (what the product's driver calls before RingLoader.close()) waits for them, the slots are unregistered
    and unmapped afterwards, and the extractor works on."""
    specs = [(s, [16000, 24000]) for s in range(40)]
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    pend = []
    rl = RingLoader(_NoiseThenBoom(), num_workers=2, slot_bytes=1 << 18, num_slots=4, start_method="forkserver")
    try:
        with pytest.raises(RuntimeError, match="boom"):
            for rb in rl.batches(specs):
                flat = rb.data.view(np.float32)
                views = [flat[o : o + n] for o, n in zip(rb.meta["offs"].tolist(), rb.meta["lens"].tolist())]
                pend.append((rb, ex.submit_host_items(views, 16000)))
                rl.pin_for(ex.plan.lib, 0)
                if len(pend) > 3:  # (a consumer that collects with a lag of three batches)
                    b, p = pend.pop(0)
                    p.wait()
                    p.release()
                    b.release()
        assert pend  # submitted, not collected
    finally:
        ex._native_pipe().drain()
        rl.close()
    for b, p in pend:
        p.release()
    del pend
    x = torch.from_numpy(np.random.RandomState(0).rand(16000).astype(np.float32) - 0.5)
    assert ex.extract(x, 16000).shape == (100, 80)


@pytest.mark.parametrize("pcm16", [False, True])
def test_submit_packed_equals_submit_on_the_views(pcm16):
    """submit_host_packed(flat, offs, lens) -- base pointer + offsets, no view per cut -- is submit_host_items on the 1-D views, bit for
    bit; bad arguments are refused before anything is queued."""
    rng = np.random.RandomState(5)
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    dt = np.int16 if pcm16 else np.float32
    lens = [16000, 9999, 32000, 4800]
    buf = np.zeros(1 << 19, dtype=np.uint8)
    w = SlotWriter(buf)
    for n in lens:
        assert w.add(rng.randint(-9000, 9000, size=n).astype(np.int16) if pcm16 else (rng.rand(n).astype(np.float32) - 0.5))
    used, offs, ln = w.finish()
    flat = buf[:used].view(dt)
    p = ex.submit_host_items([flat[o : o + n] for o, n in zip(offs.tolist(), ln.tolist())], 16000)
    want = p.wait().copy()
    p.release()
    q = ex.submit_host_packed(flat, offs, ln, 16000)
    got = q.wait().copy()
    assert list(q.frames) == [(n + 80) // 160 for n in lens]
    q.release()
    assert np.array_equal(got, want)
    with pytest.raises(ValueError, match="outside the buffer"):
        ex.submit_host_packed(flat, offs, ln + 10 ** 6, 16000)
    with pytest.raises(TypeError, match="1-D C-contiguous"):
        ex.submit_host_packed(flat.astype(np.float64), offs, ln, 16000)
    with pytest.raises(ValueError):
        ex.submit_host_packed(flat, offs[:2], ln, 16000)
