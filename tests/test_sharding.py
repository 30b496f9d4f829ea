"""CPU: the N>1 path.  Shard disjointness/coverage as the reference tests its samplers
(test/dataset/test_multinode_resume.py style: explicit rank/world), plus a REAL world_size-2
process group over gloo exercising rank discovery and the bookkeeping all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lhotse_amd import sharding as S


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("n", [0, 1, 7, 8, 100, 12501])
def test_round_robin_disjoint_and_complete(world, n):
    shards = [list(S.shard_indices(n, r, world)) for r in range(world)]
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(n))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    items = [f"cut-{i}" for i in range(n)]
    for r in range(world):
        assert S.shard(items, r, world) == [items[i] for i in shards[r]] == list(S.shard_iter(items, r, world))


def test_duration_balanced_sharding():
    rs = np.random.RandomState(0)
    dur = np.clip(np.exp(rs.randn(5000) * 0.5 + 2.4), 1, 35).tolist()  # LibriSpeech-like lengths
    parts = S.shard_by_duration(dur, 8)
    assert sorted(i for p in parts for i in p) == list(range(5000))
    loads = [sum(dur[i] for i in p) for p in parts]
    assert (max(loads) - min(loads)) / np.mean(loads) < 0.01
    assert parts == S.shard_by_duration(dur, 8)  # deterministic


def test_bad_rank():
    with pytest.raises(ValueError):
        S.shard_indices(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert S.rank_and_world() == (rank, world)
        mine = list(S.shard_indices(n, *S.rank_and_world()))
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        total, slowest = S.all_reduce_stats(len(mine), 1.0 + rank)
        q.put((rank, gathered, total, slowest))
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_group():
    world, n = 2, 101
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, total, slowest in res:
        assert sorted(i for g in gathered for i in g) == list(range(n))
        assert not set(gathered[0]) & set(gathered[1])
        assert total == n and slowest == 2.0


def test_rank_from_env(monkeypatch):
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert S.rank_and_world() == (3, 8)
    assert S.all_reduce_stats(5, 0.5) == (5, 0.5)


def test_bench_launches_itself_for_n_gt_1(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE re-execs under torch.distributed.run on 127.0.0.1 with its own argv."""
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_execv(exe, cmd):
        seen["exe"], seen["cmd"] = exe, list(cmd)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench.main()
    cmd = seen["cmd"]
    assert seen["exe"] == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    # the driver's own launch line (WORLD_SIZE already set) must NOT re-exec: it proceeds to the GPU check instead
    monkeypatch.setenv("WORLD_SIZE", "4")
    seen.clear()
    with pytest.raises(AssertionError, match="needs a GPU"):
        bench.main()
    assert not seen


def test_marker_file_rendezvous_meets_fails_fast_and_cleans_up(tmp_path, monkeypatch):
    """The rendezvous of the sharded driver for ranks that were only given RANK / WORLD_SIZE (lhotse_amd/sharding.py::_Rendezvous):
    both ranks pass once both markers exist; a failing rank releases the waiting one at once; markers of another run do not count."""
    import threading
    import time

    from lhotse_amd.sharding import _Rendezvous

    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("HIPFEAT_RUN_ID", "run-a")
    a, b = _Rendezvous(tmp_path, 0, 2, 5.0, seq=1), _Rendezvous(tmp_path, 1, 2, 5.0, seq=1)
    assert a.dist is None and b.dist is None and a.nonce and b.nonce is None
    t = threading.Thread(target=lambda: (time.sleep(0.2), b.barrier("extracted")))
    t.start()
    t0 = time.time()
    a.barrier("extracted")
    t.join()
    assert 0.15 < time.time() - t0 < 3.0 and b.nonce == a.nonce
    a.close(), b.close()
    assert not list(tmp_path.glob(".extracted-*"))  # every rank removed its marker of the first barrier
    # markers of an earlier run ("run-a") do not satisfy a new one
    monkeypatch.setenv("HIPFEAT_RUN_ID", "run-b")
    c = _Rendezvous(tmp_path, 0, 2, 0.3, seq=1)
    (tmp_path / f".extracted-run-a-1-{a.nonce}-1").write_text("done")
    with pytest.raises(TimeoutError, match=r"ranks \[1\] did not reach 'extracted'"):
        c.barrier("extracted")
    # a failing rank releases the others immediately
    d, e = _Rendezvous(tmp_path, 0, 2, 30.0, seq=2), _Rendezvous(tmp_path, 1, 2, 30.0, seq=2)
    threading.Thread(target=lambda: (time.sleep(0.2), e.failed())).start()
    t0 = time.time()
    with pytest.raises(RuntimeError, match=r"ranks \[1\] failed before 'combined'"):
        d.barrier("combined")
    assert time.time() - t0 < 5.0


def test_marker_file_rendezvous_survives_a_repeated_run_token(tmp_path, monkeypatch):
    """the token falls back to the parent pid, so a re-run from the same shell (the advertised per-shard resume) meets the
    leftovers of the run before it: a `failed` marker that used to fail every retry at once, `combined` markers that let a rank run
    through the second barrier on its own and strand the other one until the timeout.  With the per-launch nonce they are inert."""
    import threading
    import time

    from lhotse_amd.sharding import _Rendezvous

    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    monkeypatch.delenv("HIPFEAT_RUN_ID", raising=False)  # -> ppid, the same for both "launches" of this test

    def launch(fail_rank=None, late=(1,)):
        """One launch of two ranks (threads); the ranks in `late` start 0.3 s after the others, i.e. they find the other's fresh files and
        the other finds only THEIR stale ones.  Returns per-rank outcome."""
        out = {}

        def rank_main(r):
            try:
                if r in late:
                    time.sleep(0.3)
                m = _Rendezvous(tmp_path, r, 2, 10.0, seq=1)
                try:
                    if r == fail_rank:
                        raise ValueError("boom")
                    m.barrier("extracted")
                    m.barrier("combined")
                except BaseException:
                    m.failed()
                    raise
                finally:
                    m.close()
                out[r] = "ok"
            except BaseException as e:  # noqa: BLE001
                out[r] = type(e).__name__

        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
        t0 = time.time()
        [t.start() for t in ts]
        [t.join() for t in ts]
        return out, time.time() - t0

    out, dt = launch(fail_rank=1)           # launch 1 fails on rank 1: rank 0 is released at once ...
    assert out == {0: "RuntimeError", 1: "ValueError"} and dt < 5.0
    assert list(tmp_path.glob(".failed-*-1"))  # ... and the failed marker stays behind
    out, dt = launch()                      # launch 2, same token: the stale `failed` marker does not fail it
    assert out == {0: "ok", 1: "ok"} and dt < 5.0
    assert list(tmp_path.glob(".combined-*"))  # its `combined` markers stay behind (a slower rank may still be polling them)
    for late in ((1,), (0,)):               # launches 3 and 4: nobody runs through a barrier on the strength of stale markers, whoever is first
        out, dt = launch(late=late)
        assert out == {0: "ok", 1: "ok"} and 0.25 < dt < 5.0
    # the leftovers do not pile up: one hello per rank, one nonce, the `combined` markers of the last launch
    names = sorted(p.name.split("-")[0] for p in tmp_path.iterdir())
    assert names == [".combined", ".combined", ".hello", ".hello", ".nonce"], names


def test_shard_manifests_keep_the_whole_stem(tmp_path):
    """`cuts.train.jsonl.gz` and `cuts.dev.jsonl.gz` in one directory must not map to the same shard manifests."""
    a = S.shard_paths(tmp_path / "feats", tmp_path / "cuts.train.jsonl.gz", 1)
    b = S.shard_paths(tmp_path / "feats", tmp_path / "cuts.dev.jsonl.gz", 1)
    assert a[1].name == "cuts.train-1.jsonl.gz" and b[1].name == "cuts.dev-1.jsonl.gz" and a[0].name == "feats-1"
    assert S.shard_paths(tmp_path, tmp_path / "cuts.jsonl.gz", 0)[1].name == "cuts-0.jsonl.gz"  # (what the drivers' tests rely on)
    assert S.shard_paths(tmp_path, tmp_path / "cuts.jsonl", 3)[1].name == "cuts-3.jsonl.gz"


@pytest.mark.parametrize("mode,why", [("no-gpus", "fewer GPUs than ranks"), ("rccl-raises", "RCCL initialisation failed")])
def test_bench_group_falls_back_to_gloo_under_the_launcher(mode, why):
    """The driver's launch line (`python -m torch.distributed.run ... bench.py --gpus N`) on a box where RCCL cannot initialise: both
    ranks must end up in ONE gloo group (bench.py::init_dist) -- the launcher's agent store cannot serve a second initialisation, so the
    fallback brings its own TCPStore."""
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "_bench_dist_helper.py"), mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line and "world=2 sum=3.0" in line[0] and why in line[0], (out.stdout, out.stderr[-2000:])


def _fake_sysfs(root, gpus, node_cpus):
    """A miniature /sys: KFD topology (one CPU node + `gpus` GPU nodes at the given (domain, bus, dev, numa_node)), the PCI devices'
    numa_node files and the nodes' cpulists."""
    import os

    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)

    put(f"{root}/class/kfd/kfd/topology/nodes/0/properties", "cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (dom, bus, dev, node) in enumerate(gpus):
        put(f"{root}/class/kfd/kfd/topology/nodes/{i + 1}/properties", f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {(bus << 8) | (dev << 3)}\ndomain {dom}\n")
        put(f"{root}/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0/numa_node", f"{node}\n")
    for node, text in node_cpus.items():
        put(f"{root}/devices/system/node/node{node}/cpulist", text + "\n")


def test_numa_binding_reads_the_topology_and_never_widens(tmp_path, monkeypatch):
    """bind_to_gpu_numa_node on a fake /sys (no GPU here, so the PCI address comes from the KFD topology order): GPU k -> its PCI
    device's numa_node -> that node's cpulist, intersected with the CPUs the process may already use; numa_node = -1, an unknown GPU
    and the off switch are reported, not raised; a real application pins every thread of the process."""
    import os
    import threading

    from lhotse_amd import sharding as S

    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "HIPFEAT_NUMA_BIND"):
        monkeypatch.delenv(k, raising=False)
    allowed = sorted(os.sched_getaffinity(0))
    lo = ",".join(str(c) for c in allowed[: max(1, len(allowed) // 2)])
    root = str(tmp_path / "sys")
    _fake_sysfs(root, [(0, 0x05, 0, 0), (0, 0x85, 0, 1), (1, 0xC5, 0, -1)], {0: lo + ",100000-100003", 1: "100000-100007"})
    assert S._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and S._parse_cpulist("\n") == []
    assert S._gpu_pci_address(1, root) == "0000:85:00.0" and S._gpu_pci_address(2, root) == "0001:c5:00.0" and S._gpu_pci_address(3, root) is None
    dry = S.bind_to_gpu_numa_node(0, sysfs=root, apply=False)
    assert dry["node"] == 0 and dry["pci"] == "0000:05:00.0" and dry["cpus"] == max(1, len(allowed) // 2) and not dry["bound"]
    other = S.bind_to_gpu_numa_node(1, sysfs=root, apply=False)  # node 1's CPUs are not ours: nothing to bind to, and no widening
    assert other["node"] == 1 and other["cpus"] == 0 and not other["bound"] and "no CPU" in other["why"]
    none = S.bind_to_gpu_numa_node(2, sysfs=root)
    assert none["node"] == -1 and not none["bound"] and "-1" in none["why"]
    assert not S.bind_to_gpu_numa_node(7, sysfs=root)["bound"]
    monkeypatch.setenv("HIPFEAT_NUMA_BIND", "0")
    assert S.bind_to_gpu_numa_node(0, sysfs=root)["why"] == "HIPFEAT_NUMA_BIND=0"
    monkeypatch.delenv("HIPFEAT_NUMA_BIND")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")  # a masked index does not map onto the topology order
    assert S._gpu_pci_address(0, root) is None
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # the real thing, on a second thread that exists already: both end up on node 0's CPUs; restored afterwards
    seen, go, done = [], threading.Event(), threading.Event()

    def worker():
        go.wait(10)
        seen.append(sorted(os.sched_getaffinity(0)))
        done.set()

    t = threading.Thread(target=worker)
    t.start()
    try:
        got = S.bind_to_gpu_numa_node(0, sysfs=root)
        assert got["bound"] and sorted(os.sched_getaffinity(0)) == allowed[: max(1, len(allowed) // 2)]
        go.set()
        assert done.wait(10) and seen[0] == allowed[: max(1, len(allowed) // 2)]
    finally:
        go.set()
        t.join()
        os.sched_setaffinity(0, allowed)


def _gather_json_rank(rank, world, port, q):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = {"host_fed_cuts_per_s": {"batch_60": 100.0 * (rank + 1), "what": "zażółć " * (rank + 1)}, "rank": rank}
    got = bench.gather_extras(local, dist, world, torch.device("cpu"))
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gathers_the_host_fed_legs_of_all_ranks_with_plain_tensor_collectives():
    """bench.py's N > 1 host-fed report: dicts of different sizes (non-ASCII text included) from three gloo ranks, gathered with all_reduce /
    all_gather of byte tensors only -- every rank sees every rank's dict, and the numeric leaves are summed."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gather_json_rank, args=(r, 3, port, q)) for r in range(3)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(3))
    [p.join(timeout=60) for p in ps]
    for r in range(3):
        assert [o["rank"] for o in res[r]["per_rank"]] == [0, 1, 2]
        assert res[r]["per_rank"][2]["host_fed_cuts_per_s"]["what"] == "zażółć " * 3
        assert res[r]["aggregate_over_ranks"]["host_fed_cuts_per_s"]["batch_60"] == 600.0 and res[r]["aggregate_over_ranks"]["rank"] == 3.0
