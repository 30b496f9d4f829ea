#!/usr/bin/env python3
"""
bench.py -- throughput of the hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cuts C] [--no-cpu-baseline]

A "step" is ONE pass of the feature-extraction path over one batch of synthetic cuts that
is already resident in HBM: C cuts x 10 s @ 16 kHz float32 -> C x (1000, 80) float32 log-mel
(BASELINE.json configs[1]: "Synthetic 10k x 10 s 16 kHz mono cuts, 80-dim log-mel Fbank,
1xMI355X"; C defaults to 10 000 per GPU).  All C cuts hold distinct random data (6.4 GB of
input per GPU, far beyond the 256 MiB Infinity Cache), generated on the device.

For N > 1 the driver launches one process per GPU (torch.distributed.run); cuts are sharded
with no data-path collective: every rank extracts its own C cuts, so the
run is WEAK scaling and `value` = N*C*K / max-over-ranks time.  RCCL is used only for the
barrier and the MAX reduction of the elapsed time.

The JSON line also carries
  roofline      the dominant kernel against the HBM roofline: ALGORITHMIC bytes
                (960 000 B per 10 s cut: 640 000 read + 320 000 written)
                per launch / average launch duration measured here with HIP events on the
                launch stream; `traffic` = measured HBM bytes per launch from the committed
                rocprofv3 PMC passes (profiles/traffic.json), or null;
  cpu_baseline  the reference's CPU Fbank path restated with its own torch calls (oracle/kaldi_torch.py,
                kind "port": /root/reference cannot travel) timed on this host on a bounded sample of the
                same workload (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
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


def cpu_baseline(seconds: float = 12.0, procs: int = 0):
    """Time lhotse's CPU Fbank path on this host.  /root/reference does not exist on the GPU box, so the path is
    restated in oracle/kaldi_torch.py with the reference's own sequence of torch (ATen) calls -- as_strided framing,
    rfft, matmul, log -- bit-identical to the reference on the golden vectors (tests/test_oracle.py): one cut per call
    as in CutSet.compute_and_store_features, `procs` single-threaded processes in parallel, mirroring `num_jobs=procs`
    with torch.set_num_threads(1) (lhotse/bin/modes/features.py:25-32).  Workers are plain subprocesses with a hard
    timeout, so a stuck worker can never hang the bench."""
    import subprocess

    ncpu = os.cpu_count() or 1
    procs = procs or min(ncpu, 32)
    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    ps = [subprocess.Popen([sys.executable, worker, str(seconds), str(100 * i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
          for i in range(procs)]
    res = []
    deadline = time.time() + seconds + 90
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            n, dt = out.split()
            res.append((int(n), float(dt)))
        except Exception:
            p.kill()
    if not res:
        return {"value": None, "unit": "cuts/s", "cores": 0, "kind": "port", "sample": "CPU baseline workers failed"}
    rate = sum(n / dt for n, dt in res)
    total = sum(n for n, _ in res)
    return {
        "value": round(rate, 1),
        "unit": "cuts/s",
        "cores": len(res),
        "kind": "port",
        "sample": f"{total} x 10 s cuts in {seconds:.0f} s wall: {len(res)} single-threaded processes of the reference's torch CPU Fbank "
        f"call sequence (oracle/kaldi_torch.py, bit-identical to the reference on the goldens; {rate / len(res):.0f} cuts/s per core); "
        f"host has {ncpu} logical cores",
    }


def load_traffic(kernel_name: str):
    """Measured HBM bytes per cut from the committed PMC profile, if it matches the kernel."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if kernel_name.split(" ")[0] == t.get("kernel"):  # plan.kernel_name = "<kernel> lds=... blocks/CU=..."
            return float(t["hbm_bytes_per_cut"])
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cuts", type=int, default=10000, help="cuts per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="host processes of the CPU baseline (default min(cores, 32))")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (self-test of the N>1 path on one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node {args.gpus} ...")
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
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

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
    # per-launch device time: HIP events on the launch stream (torch's current stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    # sanity: the timed buffer holds real features
    chk = out[:FRAMES_PER_CUT].float()
    assert torch.isfinite(chk).all() and float(chk.std()) > 0.1

    if rank == 0:
        total_cuts = C * args.steps * world
        value = total_cuts / elapsed
        achieved = ALGO_BYTES_PER_CUT * C / (launch_ms * 1e-3)
        bytes_per_cut = load_traffic(plan.kernel_name)
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
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved / 1e9, 2),
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 4),
                "traffic": None if bytes_per_cut is None else round(bytes_per_cut * C),
                "launch_ms": round(launch_ms, 4),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CUT * C,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_procs)
        print(json.dumps(res), flush=True)
    L.check("hipfeat_layout_destroy", layout)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
