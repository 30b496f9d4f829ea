"""
The multi-GPU product path on CPU: lhotse_amd.compute_and_store_features_sharded run by TWO (and EIGHT) spawned ranks (a real world_size-2 / -8 group over
gloo, then again with nothing but RANK / WORLD_SIZE) under the real lhotse, with the oracle-backed stand-in for the device plan (no GPU
here).  What must hold (reference: CutSet.compute_and_store_features(num_jobs=N), lhotse/cut/set.py:2141-2195 -- LazySlicer shards,
per-job `feats-{i}` storage, combined manifests):
  * the combined manifest equals the single-process manifest cut for cut (same order, same Features fields) and feature for feature;
  * every rank wrote only its own shard (feats-r, cuts-r.jsonl.gz), the shards are disjoint and complete;
  * an interrupted run resumes per shard: what a rank's manifest already holds is not extracted again.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.reference

LENGTHS = [16000, 24000, 12345, 32000, 8000, 20000, 9000, 16001, 30000, 11111, 4000]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, workdir, first, balance, q, sub="sharded", workers=0):
    """One rank: a fresh process, as under torchrun."""
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), HIPFEAT_RUN_ID=f"test-{port}")
        if port:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        else:
            os.environ.pop("MASTER_ADDR", None), os.environ.pop("MASTER_PORT", None)
        from pathlib import Path

        from _dropin_support import import_lhotse, install_wave_backend, make_cpu_plan

        import_lhotse()
        import lhotse_amd as LA
        import lhotse_amd.extractors as E
        from lhotse import CutSet

        E._Plan = make_cpu_plan()
        install_wave_backend()
        work = Path(workdir)
        cuts = CutSet.from_jsonl_lazy(work / "cuts.jsonl.gz")
        if first:
            cuts = cuts.subset(first=first)
        out = LA.compute_and_store_features_sharded(cuts, LA.HipFbank(), storage_path=work / sub, manifest_path=work / sub / "cuts.jsonl.gz",
                                                    batch_duration=3.0, num_workers=workers, balance=balance, barrier_timeout=120.0)
        import torch.distributed as dist

        assert not dist.is_initialized()  # the group the driver created for its barrier is gone again
        # a second call in the same processes (everything is in the manifests already): the rendezvous must come up again
        again = LA.compute_and_store_features_sharded(cuts, LA.HipFbank(), storage_path=work / sub, manifest_path=work / sub / "cuts.jsonl.gz",
                                                      batch_duration=3.0, num_workers=workers, balance=balance, barrier_timeout=120.0)
        assert [c.id for c in again] == [c.id for c in out] and not dist.is_initialized()
        q.put((rank, [c.id for c in out], dict(LA.storage.TEMPLATE_STATS), None))
    except BaseException as e:  # noqa: BLE001 -- the parent must see the reason
        import traceback

        q.put((rank, None, None, traceback.format_exc()))
        raise


def _run_ranks(workdir, world=2, first=0, balance="round_robin", group=True, sub="sharded", workers=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port() if group else 0
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, str(workdir), first, balance, q, sub, workers)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, ids, stats, err = q.get(timeout=300)
        assert err is None, f"rank {rank} failed:\n{err}"
        res[rank] = (ids, stats)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    """The corpus on disk + the single-process answer."""
    from _dropin_support import import_lhotse, install_wave_backend, make_cpu_plan, write_cutset

    import_lhotse()
    import lhotse_amd as LA
    import lhotse_amd.extractors as E
    from lhotse.audio.backend import set_current_audio_backend

    work = tmp_path_factory.mktemp("sharded")
    prev_backend = install_wave_backend()
    prev_plan, E._Plan = E._Plan, make_cpu_plan()
    try:
        cuts = write_cutset(work, LENGTHS)
        cuts.to_file(work / "cuts.jsonl.gz")
        single = list(LA.compute_and_store_features_batch(cuts, LA.HipFbank(), storage_path=work / "single", manifest_path=work / "single.jsonl.gz",
                                                          batch_duration=3.0, num_workers=0))
        assert [c.id for c in single] == [f"cut{i}" for i in range(len(LENGTHS))]
        yield work, single
    finally:
        E._Plan = prev_plan
        set_current_audio_backend(prev_backend)


def _check_combined(work, single, sub="sharded", ranks=2):
    from lhotse import CutSet

    combined = list(CutSet.from_file(work / sub / "cuts.jsonl.gz"))
    assert [c.id for c in combined] == [c.id for c in single]
    for a, b in zip(combined, single):
        da, db = a.to_dict(), b.to_dict()
        assert da["features"]["storage_path"].endswith(f"feats-{int(a.id[3:]) % ranks}.hfa") or "duration" in sub
        for k in ("storage_path", "storage_key"):
            da["features"].pop(k), db["features"].pop(k)
        assert da == db
        assert np.array_equal(a.load_features(), b.load_features())
    return combined


def test_two_ranks_equal_one_process_and_resume(world):
    from lhotse import CutSet

    work, single = world
    # (1) an "interrupted" run: only the first 6 cuts exist yet
    res = _run_ranks(work, first=6)
    assert res[0][0] == [f"cut{i}" for i in range(6)]          # rank 0 returns the combined manifest, in input order
    assert res[1][0] == ["cut1", "cut3", "cut5"]               # the others return their own shard
    sizes = [os.path.getsize(work / "sharded" / f"feats-{r}.hfa") for r in range(2)]
    # (2) the full run resumes per shard: the bytes of the first three cuts of every shard stay, the rest is appended behind them
    res = _run_ranks(work)
    assert res[0][0] == [c.id for c in single]
    for r in range(2):
        shard = list(CutSet.from_file(work / "sharded" / f"cuts-{r}.jsonl.gz"))
        assert [c.id for c in shard] == [f"cut{i}" for i in range(r, len(LENGTHS), 2)]
        assert all(c.features.storage_path.endswith(f"feats-{r}.hfa") for c in shard)
        keys = [int(c.features.storage_key.split(":")[0]) for c in shard]
        assert keys == sorted(keys) and keys[3] == sizes[r]    # cut #4 of the shard starts where the interrupted run stopped
        assert os.path.getsize(work / "sharded" / f"feats-{r}.hfa") == sum(c.num_frames * 80 * 4 for c in shard)
    # the template path of the manifest writer really ran behind lhotse's sampler (it used to be dead code there)
    assert all(stats["template"] > 0 and stats["fallback"] == 0 for _, stats in res.values())
    _check_combined(work, single)
    # (3) running again changes nothing
    before = [os.path.getsize(work / "sharded" / f"feats-{r}.hfa") for r in range(2)]
    _run_ranks(work)
    assert before == [os.path.getsize(work / "sharded" / f"feats-{r}.hfa") for r in range(2)]
    _check_combined(work, single)


def test_eight_ranks_equal_one_process(world):
    """The node's real shape: EIGHT ranks over gloo on the same 11-cut corpus -- ranks 0-2 own two cuts, ranks 3-7 one; the
    combined manifest is still the single-process manifest cut for cut and feature for feature, every shard holds exactly its round-robin
    share, and a second call in the same eight processes (everything already extracted) meets again and changes nothing."""
    from lhotse import CutSet

    work, single = world
    res = _run_ranks(work, world=8, sub="sharded8")
    assert res[0][0] == [c.id for c in single]
    for r in range(8):
        shard = list(CutSet.from_file(work / "sharded8" / f"cuts-{r}.jsonl.gz"))
        assert [c.id for c in shard] == [f"cut{i}" for i in range(r, len(LENGTHS), 8)]
        assert res[r][0] == [c.id for c in (single if r == 0 else shard)]
        assert os.path.getsize(work / "sharded8" / f"feats-{r}.hfa") == sum(c.num_frames * 80 * 4 for c in shard)
    _check_combined(work, single, sub="sharded8", ranks=8)


def test_ranks_without_a_rendezvous_address_meet_through_marker_files(world, monkeypatch):
    work, single = world
    import shutil

    shutil.rmtree(work / "sharded", ignore_errors=True)
    res = _run_ranks(work, group=False, balance="duration")
    assert res[0][0] == [c.id for c in single]
    from lhotse import CutSet

    combined = list(CutSet.from_file(work / "sharded" / "cuts.jsonl.gz"))
    assert [c.id for c in combined] == [c.id for c in single]
    for a, b in zip(combined, single):
        assert np.array_equal(a.load_features(), b.load_features())
    # duration-balanced shards: both ranks carry about half of the audio
    loads = [sum(c.duration for c in CutSet.from_file(work / "sharded" / f"cuts-{r}.jsonl.gz")) for r in range(2)]
    assert abs(loads[0] - loads[1]) / sum(loads) < 0.1


def test_two_ranks_each_behind_its_own_ring_loader(world):
    """Round 6: every rank of the sharded driver loads its shard through its own shared-memory ring (the default loader with
    num_workers > 0): two spawned gloo ranks x one loader worker each equal the single-process run cut for cut, feature for feature."""
    work, single = world
    res = _run_ranks(work, world=2, sub="sharded_ring", workers=1)
    assert sorted(res) == [0, 1]
    assert sum(s.get("native", 0) for _, s in res.values()) >= len(single)  # (every line came out of the C splice, fed by the ring's workers)
    _check_combined(work, single, sub="sharded_ring")
