#!/usr/bin/env python3
"""
bench.py -- throughput of the hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cuts C] [--no-cpu-baseline] [--no-host-fed]

A "step" is ONE pass of the feature-extraction path over one batch of synthetic cuts that
is already resident in HBM: C cuts x 10 s @ 16 kHz float32 -> C x (1000, 80) float32 log-mel
(BASELINE.json configs[1]: "Synthetic 10k x 10 s 16 kHz mono cuts, 80-dim log-mel Fbank,
1xMI355X"; C defaults to 10 000 per GPU).  All C cuts hold distinct random data (6.4 GB of
input per GPU, far beyond the 256 MiB Infinity Cache), generated on the device.

N > 1: `python bench.py --gpus N` launches itself as one process per GPU through
torch.distributed.run (rendezvous on 127.0.0.1); when the driver has already done that
(WORLD_SIZE is set) the ranks just run.  Cuts are sharded with no data-path collective
(the reference shards the same way on CPU: LazySlicer(k, n) + per-shard storage,
lhotse/cut/set.py:2141-2160): every rank extracts its own C cuts, so the run is WEAK scaling and
`value` = N*C*K / max-over-ranks time.  RCCL carries only the barrier, the MAX reduction of the
elapsed time and the gather of the per-rank launch times and parity numbers.

The JSON line also carries
  parity        EVERY rank compares >= 64 cuts sampled from its TIMED output buffer with the oracle
                (oracle/kaldi_ref.py, float32 = the reference's arithmetic, float64 = truth): the
                worst rel_l2 / max_abs over all ranks, the fraction of values within rtol 1e-4 +
                atol 1e-3, and the oracle's own float32-vs-float64 floor (SURVEY section 8d
                "parity check in the same run");
  roofline      the dominant kernel against the HBM roofline: ALGORITHMIC bytes
                (960 000 B per 10 s cut: 640 000 read + 320 000 written)
                per launch / average launch duration measured here with HIP events on the
                launch stream; `traffic` = HBM bytes per launch from the committed rocprofv3
                PMC passes (`traffic_source`), or null; `secondary` = the f32 VALU issue
                roofline of the same kernel (instruction count per frame from the same PMC passes);
  cpu_baseline  the reference's CPU Fbank path restated with its own torch calls (oracle/kaldi_torch.py,
                kind "port": /root/reference cannot travel) timed on this host on a bounded sample of the
                same workload (rank 0, N == 1 only): B = one single-threaded process per core,
                A (`batched`) = extract_batch-style batches of 60 cuts with torch's default threads;
  extra         host_fed_cuts_per_s: HipFbank.extract_batch on pinned host tensors (PCIe-inclusive,
                never `value`), batches of 60 and 1024 cuts (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLES_PER_CUT = 160000  # 10 s @ 16 kHz
FRAMES_PER_CUT = 1000
NUM_MELS = 80
ALGO_BYTES_PER_CUT = SAMPLES_PER_CUT * 4 + FRAMES_PER_CUT * NUM_MELS * 4  # 960 000
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
NUM_SIMDS = 256 * 4
MAX_CLOCK = 2.4e9
PARITY_CUTS = 64


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_run(seconds: float, procs: int):
    """`procs` single-threaded worker processes for `seconds`; returns (cuts/s summed over workers, cuts, workers that answered)."""
    import subprocess

    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    ps = [subprocess.Popen([sys.executable, worker, str(seconds), str(100 * i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
          for i in range(procs)]
    res = []
    deadline = time.time() + seconds + 120
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            n, dt = out.split()
            res.append((int(n), float(dt)))
        except Exception:
            p.kill()
    return sum(n / dt for n, dt in res), sum(n for n, _ in res), len(res)


def cpu_baseline(seconds: float = 12.0, procs: int = 0):
    """Time lhotse's CPU Fbank path on this host.  /root/reference does not exist on the GPU box, so the path is
    restated in oracle/kaldi_torch.py with the reference's own sequence of torch (ATen) calls -- as_strided framing,
    rfft, matmul, log -- bit-identical to the reference on the golden vectors (tests/test_oracle.py).
    B (`value`): one cut per call as in CutSet.compute_and_store_features, N single-threaded processes in parallel,
    mirroring `num_jobs=N` with torch.set_num_threads(1) (lhotse/bin/modes/features.py:25-32).  The path is memory-bound
    on the host (each cut streams ~10 MB of intermediates), so more processes are not always faster: a short sweep over
    N = cores/8 .. cores/2 (or --cpu-procs) is timed and the BEST total is reported, with the whole sweep alongside.
    A (`batched`): batches of 60 cuts (600 s, the batch driver's default) through the batched forward with torch's
    default intra-op threads, as Fbank.extract_batch runs it.  Workers are plain subprocesses with a hard timeout."""
    import subprocess

    ncpu = os.cpu_count() or 1
    if procs:
        sweep = [procs]
    else:
        sweep = sorted({max(1, min(ncpu, n)) for n in (ncpu // 8, ncpu // 4, ncpu // 2)})
    per = max(4.0, seconds / len(sweep))
    runs = []
    for n in sweep:
        rate, cuts, ok = _cpu_run(per, n)
        if ok:
            runs.append({"processes": ok, "cuts_per_s": round(rate, 1), "cuts": cuts, "seconds": per})
    if not runs:
        return {"value": None, "unit": "cuts/s", "cores": 0, "kind": "port", "sample": "CPU baseline workers failed"}
    best = max(runs, key=lambda r: r["cuts_per_s"])
    out = {
        "value": best["cuts_per_s"],
        "unit": "cuts/s",
        "cores": best["processes"],
        "kind": "port",
        "cpu_model": cpu_model(),
        "logical_cores": ncpu,
        "sweep": runs,
        "sample": f"{best['cuts']} x 10 s cuts in {best['seconds']:.0f} s wall: {best['processes']} single-threaded processes of the reference's torch CPU Fbank "
        f"call sequence (oracle/kaldi_torch.py, bit-identical to the reference on the goldens; {best['cuts_per_s'] / best['processes']:.0f} cuts/s per process), "
        f"best of a sweep over {[r['processes'] for r in runs]} processes; host has {ncpu} logical cores ({cpu_model()})",
    }
    # baseline A: batched, default intra-op threads
    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    try:
        p = subprocess.run([sys.executable, worker, str(min(seconds, 6.0)), "7", "batched"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           text=True, timeout=seconds + 120)
        n, dt, threads = p.stdout.split()
        out["batched"] = {"value": round(int(n) / float(dt), 1), "unit": "cuts/s", "threads": int(threads),
                          "sample": f"{n} cuts as batches of 60 x 10 s through the batched forward, torch default intra-op threads"}
    except Exception as e:  # the line must still be printed
        out["batched"] = {"value": None, "error": repr(e)}
    return out


def load_profile_constants(kernel_name: str):
    """HBM bytes per cut and VALU instructions per frame from the committed PMC profile, if it matches the kernel."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if kernel_name.split(" ")[0] == t.get("kernel"):  # plan.kernel_name = "<kernel> lds=... blocks/CU=..."
            return t
    except Exception:
        pass
    return {}


def parity_check(wave, out, C, rank):
    """>= 64 cuts sampled from the timed output buffer against the oracle (float32 and float64)."""
    import numpy as np

    from oracle.kaldi_ref import RefConfig, RefExtractor

    rs = np.random.RandomState(4321 + rank)
    n = min(PARITY_CUTS, C)
    idx = np.sort(rs.choice(C, size=n, replace=False))
    o32, o64 = RefExtractor(RefConfig(kind="fbank"), np.float32), RefExtractor(RefConfig(kind="fbank"), np.float64)
    rel_max = abs_max = floor_max = 0.0
    within = total = 0
    for i in idx:
        x = wave[int(i)].cpu().numpy()
        got = out[int(i) * FRAMES_PER_CUT : (int(i) + 1) * FRAMES_PER_CUT].cpu().numpy()
        want, truth = o32.extract(x), o64.extract(x)
        assert got.shape == want.shape, (got.shape, want.shape)
        d = np.abs(got.astype(np.float64) - want)
        rel_max = max(rel_max, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
        abs_max = max(abs_max, float(d.max()))
        floor_max = max(floor_max, float(np.linalg.norm(want - truth) / np.linalg.norm(truth)))
        within += int((d <= 1e-3 + 1e-4 * np.abs(want)).sum())
        total += d.size
    return {"rel_l2_max": rel_max, "max_abs_max": abs_max, "frac_within": within / total, "n": int(n), "oracle_f32_vs_f64_rel_l2_max": floor_max}


def host_fed(ex, seconds: float = 2.0):
    """PCIe-inclusive rate of the drop-in API: extract_batch(padded host tensor + lengths) -> features back on the host
    (lhotse/cut/set.py:2393-2398 calls it exactly so), through the chunked H2D / kernel / D2H pipeline of lhotse_amd/extractors.py.
    Page-locked float32 (the bound is PCIe: 63 GB/s / 640 KB = 98 k cuts/s), page-locked int16 PCM (half the upload), and pageable
    float32 (what a DataLoader hands over: one extra host copy into pinned staging)."""
    import torch

    res = {}
    for tag, dtype, pin in (("", torch.float32, True), ("_int16", torch.int16, True), ("_pageable", torch.float32, False)):
        for B in (60, 1024):
            x = torch.rand(B, SAMPLES_PER_CUT) - 0.5
            if dtype == torch.int16:
                x = (x * 32767).to(torch.int16)
            if pin:
                x = x.pin_memory()
            lens = torch.full((B,), SAMPLES_PER_CUT, dtype=torch.int32)
            ex.extract_batch(x, 16000, lengths=lens)
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                r = ex.extract_batch(x, 16000, lengths=lens)
                assert r.shape == (B, FRAMES_PER_CUT, NUM_MELS)
                n += B
            torch.cuda.synchronize()
            res[f"batch_{B}{tag}"] = round(n / (time.perf_counter() - t0), 1)
            del x
    res["what"] = ("HipFbank.extract_batch((B, 160000) host tensor, lengths) -> host features; PCIe-inclusive, never `value`; "
                   "no suffix = page-locked float32, _int16 = page-locked int16 PCM, _pageable = pageable float32")
    return res


def self_launch(args) -> None:
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed launches (default: >= 1 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cuts", type=int, default=10000, help="cuts per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="host processes of the CPU baseline (default: best of a sweep over cores/8, cores/4, cores/2)")
    ap.add_argument("--input", default="uniform", choices=["uniform", "zeros", "sine"], help="synthetic input (the metric is defined on `uniform`; the others exist to expose power/DVFS effects)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (self-test of the N>1 path on one GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    if args.dist_backend == "gloo":  # self-test mode: all ranks may share one GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the RCCL path with a single rank (self-test)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
    cdev = dev if (dist is None or args.dist_backend == "nccl") else torch.device("cpu")  # where collective tensors live

    import lhotse_amd
    from lhotse_amd import _lib

    ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{local_rank}"))
    plan = ex.plan
    L = plan.lib
    C = args.cuts

    # ---- synthetic workload, resident in HBM: U(-1,1)*0.5, distinct per cut and per rank
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    wave = torch.empty((C, SAMPLES_PER_CUT), dtype=torch.float32, device=dev)
    chunk = 500
    for i in range(0, C, chunk):
        wave[i : i + chunk].uniform_(-0.5, 0.5, generator=g)
    if args.input == "zeros":
        wave.zero_()
    elif args.input == "sine":
        t = torch.arange(SAMPLES_PER_CUT, device=dev, dtype=torch.float32)
        wave[:] = 0.4 * torch.sin(2 * 3.14159265 * 440.0 / 16000.0 * t)
    out = torch.empty((C * FRAMES_PER_CUT, NUM_MELS), dtype=torch.float32, device=dev)
    offs = np.arange(C, dtype=np.int64) * SAMPLES_PER_CUT
    lens = np.full(C, SAMPLES_PER_CUT, dtype=np.int64)
    h = np.zeros(1, dtype=np.uint64)
    stream = torch.cuda.current_stream(dev).cuda_stream
    L.check("hipfeat_layout_create", plan.handle, C, _lib.addr(offs), _lib.addr(lens), None, None, NUM_MELS, stream, _lib.addr(h))
    layout = int(h[0])
    assert L.raw("hipfeat_layout_total_frames", layout) == C * FRAMES_PER_CUT

    def step():
        L.check("hipfeat_extract_layout", plan.handle, layout, wave.data_ptr(), out.data_ptr(), stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    out.zero_()  # the parity check below reads what the TIMED launches wrote
    barrier()
    # per-launch device time: HIP events on the launch stream (torch's current stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    rank_launch_ms = [launch_ms]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lm = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(lm, torch.tensor([launch_ms], dtype=torch.float64, device=cdev))
        rank_launch_ms = [float(x.item()) for x in lm]

    # ---- parity in the same run, on every rank, on the timed output buffer
    parity = None
    if not args.no_parity:
        chk = out[:FRAMES_PER_CUT].float()
        assert torch.isfinite(chk).all() and float(chk.std()) > 0.1
        par = parity_check(wave, out, C, rank)
        if dist is not None:
            mx = torch.tensor([par["rel_l2_max"], par["max_abs_max"], -par["frac_within"], par["oracle_f32_vs_f64_rel_l2_max"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            cnt = torch.tensor([float(par["n"])], dtype=torch.float64, device=cdev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            par = {"rel_l2_max": float(mx[0]), "max_abs_max": float(mx[1]), "frac_within": -float(mx[2]), "n": int(cnt.item()),
                   "oracle_f32_vs_f64_rel_l2_max": float(mx[3])}
        parity = {
            "rel_l2_max": float(f"{par['rel_l2_max']:.3e}"),
            "max_abs_max": float(f"{par['max_abs_max']:.3e}"),
            "frac_within_rtol1e-4_atol1e-3": round(par["frac_within"], 6),
            "n": par["n"],
            "oracle_f32_vs_f64_rel_l2_max": float(f"{par['oracle_f32_vs_f64_rel_l2_max']:.3e}"),
            "pass": bool(par["rel_l2_max"] <= 1e-4),
            "what": f"{PARITY_CUTS} cuts per rank sampled from the timed output buffer vs oracle/kaldi_ref.py (float32); worst over all ranks",
        }
        assert parity["pass"], parity

    if rank == 0:
        total_cuts = C * args.steps * world
        value = total_cuts / elapsed
        achieved = ALGO_BYTES_PER_CUT * C / (launch_ms * 1e-3)
        prof = load_profile_constants(plan.kernel_name)
        bytes_per_cut = prof.get("hbm_bytes_per_cut")
        res = {
            "metric": "cuts/sec (10 s @16 kHz -> 80-dim log-mel fbank)",
            "value": round(value, 1),
            "unit": "cuts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: {C} x 10 s 16 kHz mono cuts per GPU per step, 80-dim log-mel Fbank (25/10 ms, povey, no dither), device-resident float32 in / float32 out",
                "cuts_per_gpu_per_step": C,
                "sharding": "cuts sharded across ranks, no data-path collective",
                "kernel": plan.kernel_name,
                "world_size": world,
                "dist_backend": None if dist is None else ("rccl" if args.dist_backend == "nccl" else args.dist_backend),
                "rank_launch_ms": [round(x, 4) for x in rank_launch_ms],
            },
            "parity": parity,
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved / 1e9, 2),
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 4),
                "traffic": None if bytes_per_cut is None else round(float(bytes_per_cut) * C),
                "traffic_source": None if bytes_per_cut is None else "profiles/traffic.json (committed rocprofv3 PMC run of this kernel, not measured in this run)",
                "launch_ms": round(launch_ms, 4),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CUT * C,
            },
        }
        ipf = prof.get("valu_instr_per_frame")
        if ipf:
            # every wave64 VALU instruction occupies its SIMD's issue port for >= 2 clk (packed f32 ones 3, measured:
            # tools/ubench/valu_rate.hip); a wave instruction covers `frames_per_wave_instr` frames
            clk_per_instr = float(prof.get("valu_clk_per_instr", 2.0))
            frames_per_s = C * FRAMES_PER_CUT / (launch_ms * 1e-3)
            res["roofline"]["secondary"] = {
                "bound": "valu_f32",
                "instr_per_frame": ipf,
                "clk_per_instr": clk_per_instr,
                "achieved_frac": round(frames_per_s * float(ipf) * clk_per_instr / (NUM_SIMDS * MAX_CLOCK), 4),
                "what": "wave-level VALU instructions per frame (committed PMC run: SQ_INSTS_VALU / frames) x issue clocks per instruction "
                        "/ (1024 SIMDs x 2.4 GHz): the share of the chip's VALU issue slots this launch rate needs",
            }
        if world == 1:
            if not args.no_host_fed:
                res["extra"] = {"host_fed_cuts_per_s": host_fed(ex)}
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_procs)
        print(json.dumps(res), flush=True)
    L.check("hipfeat_layout_destroy", layout)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
