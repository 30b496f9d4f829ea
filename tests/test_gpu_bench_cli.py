"""GPU: bench.py's command line as the driver runs it -- the process-group leg on RCCL (one rank: all a 1-GPU box can show) and the
strong-scaling form of BASELINE configs[2] (`--total-cuts`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-host-fed", *flags], capture_output=True, text=True,
                       env=e, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), r.stderr


def test_the_process_group_leg_runs_on_rccl():
    """BENCH_FORCE_DIST=1: the N > 1 code path (barrier, MAX of the elapsed time, gathered launch times, parity maxima) on a
    single-rank RCCL group.  A silent fall-back to gloo on a box whose RCCL works would hide a broken RCCL path until the 8-GPU run."""
    res, err = _bench("--steps", "3", "--cuts", "2000", "--no-other-configs", env={"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29417",
                                                              "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert res["config"]["dist_backend"] == "rccl", (res["config"], err[-2000:])
    assert res["n_gpus"] == 1 and res["scaling"] == "weak" and res["parity"]["pass"] is True
    assert len(res["config"]["rank_launch_ms"]) == 1
    assert "kaldi_torch" in res["parity"]["ref32"] and res["parity"]["K_allowed"] <= 3.0  # the reference's real float32 floor, K <= 3
    assert res["parity"]["numpy32_floor_of_rounds_1_to_4"]["numpy32_vs_f64_max_abs"] < res["parity"]["oracle_f32_vs_f64_max_abs"]
    assert "bound" in res["config"]["numa"]


def test_total_cuts_is_the_same_corpus_and_the_same_rate_as_the_default_form():
    """`--total-cuts T` at N = 1 holds exactly the cuts of the default form (one generator stream), reports strong scaling, and its rate
    agrees with the weak form's (the same launch over the same data)."""
    weak, _ = _bench("--steps", "20", "--cuts", "4000", "--no-other-configs")
    strong, _ = _bench("--steps", "20", "--total-cuts", "4000")
    assert strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert strong["config"]["cuts_per_gpu_per_step"] == weak["config"]["cuts_per_gpu_per_step"] == 4000
    assert "configs[2]" in strong["config"]["workload"]
    # identical inputs -> identical sampled outputs -> identical parity figures
    for k in ("rel_l2_max", "max_abs_max", "hip_vs_f64_max_abs"):
        assert strong["parity"][k] == weak["parity"][k], k
    assert abs(strong["value"] / weak["value"] - 1.0) < 0.08, (strong["value"], weak["value"])


@pytest.mark.parametrize("flags", [("--config", "mfcc40_libri", "--cuts", "600", "--steps", "3"), ("--config", "onthefly", "--cuts", "6", "--steps", "2", "--no-extra"),
                                   ("--config", "onthefly", "--cuts", "8", "--steps", "2", "--prefetch", "4", "--streams", "2", "--no-extra"),
                                   ("--config", "bulk_save", "--cuts", "2", "--steps", "1", "--no-extra"),
                                   ("--config", "plumbing", "--cuts", "4", "--steps", "1", "--no-extra")])
def test_the_other_configs_run_and_pass_their_in_run_parity(flags):
    """Small instances of every `--config`: one JSON line, in-run oracle parity `pass`, the contract's keys."""
    res, err = _bench(*flags)
    assert res["parity"]["pass"] is True, (res["parity"], err[-1500:])
    assert res["config"]["name"] == flags[1] and res["value"] > 0 and res["roofline"]["frac"] > 0 and res["unit"] == "cuts/s"


def test_the_default_line_carries_the_other_baseline_configs_and_both_regimes():
    """The driver's form of the command (`--steps 20 --warmup 5`, default config): ONE line whose `extra.configs` holds BASELINE configs[3]
    and [4] measured under the same contract with their own in-run parity, on-the-fly priced both ways (`frac` and `frac_end_to_end`),
    and the after-idle regime next to the sustained `value` (VERDICT r4 tasks 2 and 6)."""
    res, err = _bench("--steps", "20", "--warmup", "5")
    cfgs = dict(res["extra"]["configs"])
    assert set(cfgs) == {"mfcc40_libri", "onthefly", "plumbing"}
    plumb = cfgs.pop("plumbing")  # round 6: BASELINE configs[0] with the GPU in it (host-bound: WAV files -> loader workers -> features -> storage)
    assert "error" not in plumb, plumb
    assert plumb["value"] > 0 and plumb["parity"]["pass"] is True and plumb["last_pass"]["cuts_per_s"] > 0
    legs = plumb["legs"]
    # (leg A, the CPU per-cut driver, belongs to the CPU baseline: this test runs the line with --no-cpu-baseline)
    # (the fork-hazard legs and the staging-copy variant are part of `--config plumbing`, not of the default line)
    assert any(k.startswith("B hip_batch_numpy_files") for k in legs) and any(k.startswith("C hip_bulk") for k in legs)
    assert any(k.startswith("D hip_ring float32") and "PROCESSES" not in k for k in legs) and any(k.startswith("D hip_ring") and "2 PROCESSES sharing the GPU" in k for k in legs)
    ring = next(v for k, v in legs.items() if k.startswith("D hip_ring float32") and "PROCESSES" not in k)
    assert ring["ring_slots_page_locked"] > 0 and ring["batches_uploaded_straight_from_the_ring"] > 0
    assert ring["container_cpu_quota"] is None or ring["container_cpu_quota"] > 0
    assert all(v["cuts_per_s"] > 0 for k, v in legs.items() if isinstance(v, dict)), legs
    for name, c in cfgs.items():
        assert c["parity"]["pass_rel_l2"] is True and c["value"] > 0 and c["roofline"]["frac"] > 0 and c["steps"] > 0, (name, c)
        assert c["ms_per_step"] * c["steps"] < 5000  # a few seconds of GPU together
    otf = cfgs["onthefly"]["roofline"]
    assert 0 < otf["frac_end_to_end"] < otf["frac"] and otf["algorithmic_bytes_end_to_end"] < otf["algorithmic_bytes_per_launch"]
    ramp = res["extra"]["first_launches_after_idle_ms"]
    assert len(ramp) == 10 and all(x > 0 for x in ramp)
    assert 0.7 * res["value"] < res["extra"]["contract_only"]["value"] < 1.05 * res["value"]
    assert res["parity"]["pass_rel_l2"] and res["parity"]["pass_linear"]
