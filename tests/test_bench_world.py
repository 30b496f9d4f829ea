"""CPU: bench.py's N > 1 plumbing at WORLD_SIZE 8, launched with the DRIVER'S OWN LINE (`python -m torch.distributed.run --nnodes=1
--nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W`), over gloo, with the device-less
self-test workload (BENCH_SELFTEST_STUB=1) -- group set-up, the contract's barriers, MAX over ranks, per-rank launch times, the parity
reduction, the host-fed gather, the ONE JSON line.  Only the driver can launch 8 GPUs (VERDICT r5 task 6); what it will run is this code
with `nccl` in place of `gloo` and a real workload in place of the stub."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world: int, steps: int = 4, warmup: int = 2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_SELFTEST_STUB="1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup)]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line, nobody else prints JSON
    assert [ln for ln in p.stdout.splitlines() if ln.strip() and "[Gloo]" in ln] == [], p.stdout[:500]  # (gloo's C++ connection chatter is kept off stdout)
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [8, 2])
def test_one_json_line_from_eight_ranks(world):
    steps, warmup = 4, 2
    r = _run(world, steps, warmup)
    cfg = r["config"]
    assert r["n_gpus"] == world and r["steps"] == steps and r["warmup"] == warmup and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert "SELFTEST" in cfg and cfg["world_size"] == world and cfg["dist_backend"].startswith("gloo")
    # value = units of ALL ranks x steps / the slowest rank's time
    assert abs(r["value"] - world * 10 * steps / (r["ms_per_step"] * steps * 1e-3)) / r["value"] < 1e-3
    assert len(cfg["rank_launch_ms"]) == world and all(x > 0 for x in cfg["rank_launch_ms"])
    assert max(cfg["rank_launch_ms"]) <= r["ms_per_step"] * 1.5
    # the group as every rank saw it
    g = cfg["group"]
    assert g["world_size"] == world and g["ranks_seen"] == world and sorted(x["rank"] for x in g["ranks"]) == list(range(world))
    assert len({x["pid"] for x in g["ranks"]}) == world
    assert len(cfg["numa_per_rank"]) == world and all(isinstance(n, dict) and "bound" in n for n in cfg["numa_per_rank"])
    # host-fed legs: every rank's dict + the sums
    e = r["extra"]
    assert [o["host_fed_cuts_per_s"]["batch_60"] for o in e["per_rank"]] == [100.0 * (k + 1) for k in range(world)]
    assert e["aggregate_over_ranks"]["host_fed_cuts_per_s"]["batch_60"] == 100.0 * world * (world + 1) / 2
    # parity: worst over all ranks, counts summed
    par = r["parity"]
    assert par["n"] == world and par["pass"] is True and par["statement_version"].startswith("r6")
    assert r["roofline"]["bound"] == "hbm" and "cpu_baseline" not in r
