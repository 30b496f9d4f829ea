#!/usr/bin/env python3
"""
bench.py -- throughput of the hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config fbank16k|mfcc40_libri|onthefly] [--cuts C]
                    [--no-cpu-baseline] [--no-host-fed] [--no-parity]

A "step" is ONE pass of the feature-extraction path over one batch of synthetic cuts that is already resident in HBM.

  --config fbank16k (default) BASELINE.json configs[1]: C (default 10 000) cuts x 10 s @ 16 kHz float32 -> C x (1000, 80) float32 log-mel,
                              one launch.  All cuts hold distinct random data (6.4 GB of input per GPU, far beyond the 256 MiB Infinity Cache).
  --config mfcc40_libri       configs[3]: C (default 8 000) cuts with LibriSpeech-like lengths (seeded log-normal clipped to 1-35 s, mean
                              ~12.3 s; the corpus itself is not available offline) -> 40-dim MFCC (40 filters, 40 cepstra, lifter 22),
                              one launch over the packed ragged batch.
  --config onthefly           configs[4]: a pool of mini-batches of 600 s of audio each (cuts U(1, 30) s, seed 0), every cut speed-perturbed by
                              a factor from {0.9, 1.0, 1.1} on the device, then 80-dim Fbank collated to a padded (B, Tmax, 80) tensor with
                              LOG_EPSILON -- what K2SpeechRecognitionDataset's OnTheFlyFeatures + PerturbSpeed produce per batch
                              (lhotse/dataset/input_strategies.py:351-476).  A step is one pass over the whole pool (default 64 mini-batches);
                              the waveforms are resident in HBM, as for the other configs (the PCIe-inclusive rate is `extra`).

N > 1: `python bench.py --gpus N` launches itself as one process per GPU through torch.distributed.run (rendezvous on 127.0.0.1); when
the driver has already done that (WORLD_SIZE is set) the ranks just run.  Cuts are sharded with no data-path collective (SURVEY section 8e;
the reference shards the same way on CPU: LazySlicer(k, n) + per-shard storage, lhotse/cut/set.py:2141-2160): every rank extracts its own
cuts, so the run is WEAK scaling and `value` = N * units * K / max-over-ranks time.  The process group (RCCL; gloo if RCCL cannot be
initialised -- logged in `config.dist_backend`) carries only the barrier, the MAX reduction of the elapsed time and the gather of the
per-rank launch times and parity numbers.

The JSON line also carries
  parity        EVERY rank compares >= 64 cuts sampled from its TIMED output buffer with the oracle (float32 = the reference's arithmetic,
                float64 = truth): worst rel_l2 / max_abs over all ranks, the fraction of values within rtol 1e-4 + atol 1e-3, and the
                errors of BOTH float32 implementations against the float64 oracle (max and rms).  pass = rel_l2(hip, ref32) <= 1e-4 (the north
                star's tolerance; asserted) AND max|hip - f64| <= max(2e-3, 3 x max|ref32 - f64|) (reported: a tail statistic);
  roofline      the step against the HBM roofline: ALGORITHMIC bytes per step (samples read once as float32, features
                written once as float32) / average step duration measured here with HIP events on the launch stream; `traffic` = HBM bytes
                per launch from the committed rocprofv3 PMC passes (only while profiles/traffic.json matches the kernel SOURCE it was
                measured on), else null; `secondary` = the f32 VALU issue roofline of the same kernel from the same PMC passes;
  cpu_baseline  the reference's CPU path for the same workload restated with its own torch calls (oracle/kaldi_torch.py, kind "port":
                /root/reference cannot travel) timed on this host on a bounded sample (rank 0, N == 1 only);
  extra         PCIe-inclusive rates of the drop-in API (never `value`).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 16000
SAMPLES_PER_CUT = 160000  # 10 s @ 16 kHz
FRAMES_PER_CUT = 1000
NUM_MELS = 80
ALGO_BYTES_PER_CUT = SAMPLES_PER_CUT * 4 + FRAMES_PER_CUT * NUM_MELS * 4  # 960 000
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
NUM_SIMDS = 256 * 4
MAX_CLOCK = 2.4e9
PARITY_CUTS = 64
LOG_EPSILON = -23.025850929940457  # lhotse/utils.py:50-51


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N == 1): single-threaded worker processes of the reference's torch call sequence
# ---------------------------------------------------------------------------------------------------------------------------------
def _cpu_run(seconds: float, procs: int, mode: str = ""):
    """`procs` single-threaded worker processes for `seconds`; returns (cuts/s summed over workers, cuts, workers that answered,
    audio seconds/s)."""
    import subprocess

    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    extra = [mode] if mode else []
    ps = [subprocess.Popen([sys.executable, worker, str(seconds), str(100 * i), *extra], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
          for i in range(procs)]
    res = []
    deadline = time.time() + seconds + 120
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            f = out.split()
            res.append((int(f[0]), float(f[1]), float(f[2]) if len(f) > 2 and mode else 10.0 * int(f[0])))
        except Exception:
            p.kill()
    return sum(n / dt for n, dt, _ in res), sum(n for n, _, _ in res), len(res), sum(a / dt for _, dt, a in res)


def cpu_baseline(seconds: float = 12.0, procs: int = 0, mode: str = "", what: str = "Fbank"):
    """Time lhotse's CPU path for the workload on this host.  /root/reference does not exist on the GPU box, so the path is restated in
    oracle/kaldi_torch.py with the reference's own sequence of torch (ATen) calls -- as_strided framing, rfft, matmul, log (+ DCT / lifter
    for MFCC, + F.pad / conv1d(stride) for Speed) -- pinned to the reference's outputs on the golden vectors (tests/test_oracle.py).
    B (`value`): one cut per call as in CutSet.compute_and_store_features, N single-threaded processes in parallel, mirroring
    `num_jobs=N` with torch.set_num_threads(1) (lhotse/bin/modes/features.py:25-32).  The path is memory-bound on the host, so more
    processes are not always faster: a short sweep over N = cores/8 .. cores/2 (or --cpu-procs) is timed and the BEST total is reported.
    A (`batched`, fbank16k only): batches of 60 cuts through the batched forward with torch's default intra-op threads."""
    import subprocess

    ncpu = os.cpu_count() or 1
    quota = cpu_quota()  # the container's cgroup CPU quota (the MI355X boxes of round 6: 16 CPUs' worth on a 256-thread host)
    counts = (ncpu // 8, ncpu // 4, ncpu // 2) if quota is None else (int(quota), int(1.5 * quota), int(2 * quota), ncpu // 8)
    sweep = [procs] if procs else sorted({max(1, min(ncpu, n)) for n in counts})
    per = max(4.0, seconds / len(sweep))
    runs = []
    for n in sweep:
        rate, cuts, ok, asps = _cpu_run(per, n, mode)
        if ok:
            runs.append({"processes": ok, "cuts_per_s": round(rate, 1), "audio_seconds_per_s": round(asps, 1), "cuts": cuts, "seconds": per})
    if not runs:
        return {"value": None, "unit": "cuts/s", "cores": 0, "kind": "port", "sample": "CPU baseline workers failed"}
    best = max(runs, key=lambda r: r["cuts_per_s"])
    out = {
        "value": best["cuts_per_s"],
        "unit": "cuts/s",
        "cores": best["processes"],
        "kind": "port",
        "audio_seconds_per_s": best["audio_seconds_per_s"],
        "cpu_model": cpu_model(),
        "logical_cores": ncpu,
        "container_cpu_quota": quota,
        "sweep": runs,
        "sample": f"{best['cuts']} cuts in {best['seconds']:.0f} s wall: {best['processes']} single-threaded processes of the reference's torch CPU {what} "
        f"call sequence (oracle/kaldi_torch.py, pinned to the reference's outputs on the goldens; {best['cuts_per_s'] / best['processes']:.0f} cuts/s per process), "
        f"best of a sweep over {[r['processes'] for r in runs]} processes; host has {ncpu} logical cores ({cpu_model()})"
        + ("" if quota is None else f", of which this container may use {quota:g} CPUs' worth of time (cgroup cpu.max)"),
    }
    if not mode:  # baseline A: batched, default intra-op threads
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


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), None = unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        return None


def kernel_source_hash(files) -> str:
    h = hashlib.sha256()
    for rel in files:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_profile_constants(kernel_name: str, config_name: str = "fbank16k"):
    """HBM bytes per cut and VALU instructions per frame from the committed PMC profile -- only if it was measured on THIS kernel source
    (profiles/traffic.json carries the sha256 of the files it names; a changed kernel body invalidates the numbers instead of re-labelling them).
    The top level of the file is the headline kernel; `configs[<name>]` holds the entries of the other BASELINE configs (same hash rule)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if config_name != "fbank16k":
            t = (t.get("configs") or {}).get(config_name) or {}
            if not t:
                return {}
        if kernel_name.split(" ")[0] != t.get("kernel"):  # plan.kernel_name = "<kernel> lds=... blocks/CU=..."
            return {}
        if t.get("source_sha256_16") != kernel_source_hash(t.get("source_files", [])):
            return {"stale": True}
        return t
    except Exception:
        return {}


def compare(got, want, truth, log_mel=True, alt32=None):
    """Error figures of one cut (oracle/parity_bar.py::figures -- the checker's module, imported by the parity leg only)."""
    from oracle import parity_bar

    return parity_bar.figures(got, want, truth, log_mel=log_mel, alt32=alt32)


def fold(stats):
    from oracle import parity_bar

    return parity_bar.fold(stats)


# ---------------------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------------------
SETTLE_LAUNCHES = int(os.environ.get("BENCH_SETTLE", "150"))  # (BENCH_SETTLE=0: counter passes, tools/collect.sh traffic) untimed launches at construction of the headline workload (~0.5 s): the package's steady power state
FILL_CHUNK = 500  # cuts per uniform_() call: part of the definition of the synthetic input (the generator's stream position)


def fbank16k_fill(wave, seed: int, first: int = 0) -> None:
    """The synthetic input of the headline workload: rows of U(-0.5, 0.5) from ONE device generator stream, FILL_CHUNK cuts per draw.
    `first` = global index of wave[0] (a multiple of FILL_CHUNK): the stream is advanced past the rows in front of it, so that any
    window of the 10 000-cut input can be regenerated (tests/test_gpu_parity.py re-creates the cuts the in-run parity leg samples)."""
    import torch

    assert first % FILL_CHUNK == 0
    g = torch.Generator(device=wave.device).manual_seed(seed)
    if first:
        scratch = torch.empty((FILL_CHUNK, wave.shape[1]), dtype=torch.float32, device=wave.device)
        for _ in range(first // FILL_CHUNK):
            scratch.uniform_(-0.5, 0.5, generator=g)
    for i in range(0, wave.shape[0], FILL_CHUNK):
        wave[i : i + FILL_CHUNK].uniform_(-0.5, 0.5, generator=g)


def fbank16k_fill_shard(wave, seed: int, total: int, rank: int, world: int) -> None:
    """Rank `rank`'s rows of ONE global corpus of `total` cuts (global cut i = row i of the `fbank16k_fill` stream of `seed`): the cuts
    rank, rank + world, ...  Every rank draws the whole stream chunk by chunk and keeps its own rows, so N = 1 and N = 8 extract the
    same `total` cuts (and N = 1 holds exactly the rows of `fbank16k_fill`)."""
    import torch

    g = torch.Generator(device=wave.device).manual_seed(seed)
    scratch = torch.empty((FILL_CHUNK, wave.shape[1]), dtype=torch.float32, device=wave.device)
    row = 0
    for base in range(0, total, FILL_CHUNK):
        n = min(FILL_CHUNK, total - base)
        scratch.uniform_(-0.5, 0.5, generator=g)  # always a whole chunk: the stream does not depend on where the corpus ends
        sel = scratch[(rank - base) % world : n : world]
        wave[row : row + sel.shape[0]] = sel
        row += sel.shape[0]
    assert row == wave.shape[0], (row, wave.shape)


def fbank16k_parity_indices(cuts: int, rank: int):
    """The cuts of the timed output buffer that the in-run parity leg compares with the oracle."""
    import numpy as np

    rs = np.random.RandomState(4321 + rank)
    return np.sort(rs.choice(cuts, size=min(PARITY_CUTS, cuts), replace=False))


class Fbank16k:
    """BASELINE configs[1]."""

    name = "fbank16k"
    metric = "cuts/sec (10 s @16 kHz -> 80-dim log-mel fbank)"
    default_cuts = 10000
    cpu_mode, cpu_what = "", "Fbank"

    def __init__(self, dev, rank, args):
        import numpy as np
        import torch

        import lhotse_amd
        from lhotse_amd import _lib

        self.torch, self.np, self.rank = torch, np, rank
        world = int(os.environ.get("WORLD_SIZE", "1"))
        total = int(getattr(args, "total_cuts", 0) or 0)
        # --total-cuts T (BASELINE configs[2] as written: T cuts SHARDED over the ranks): rank r takes the cuts r, r + W, ... of ONE global
        # synthetic corpus, as CutSet.compute_and_store_features shards with LazySlicer(k=r, n=W) (lhotse/cut/set.py:2158-2160)
        C = self.C = len(range(rank, total, world)) if total else (args.cuts or self.default_cuts)
        assert C > 0, "fewer cuts than ranks"
        self.ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        L = self.L = self.plan.lib
        self.wave = torch.empty((C, SAMPLES_PER_CUT), dtype=torch.float32, device=dev)
        if total:
            fbank16k_fill_shard(self.wave, 1234, total, rank, world)
        else:
            fbank16k_fill(self.wave, 1234 + rank)
        if args.input == "zeros":
            self.wave.zero_()
        elif args.input == "sine":
            t = torch.arange(SAMPLES_PER_CUT, device=dev, dtype=torch.float32)
            self.wave[:] = 0.4 * torch.sin(2 * 3.14159265 * 440.0 / 16000.0 * t)
        self.out = torch.empty((C * FRAMES_PER_CUT, NUM_MELS), dtype=torch.float32, device=dev)
        # a second output buffer, zeroed HERE, for the timed steps: the parity leg must read what the TIMED steps wrote, and any clearing
        # kernel between the warm-up and the timed steps (a 3.2 GB memset, but also a 65-row index_fill_) takes the package out of its steady
        # power state -- the first ~6 timed launches then run up to 20 % slower, a quarter of the driver's 20 steps (tools/launch_ramp2.py and
        # BENCH_SHOW_LAUNCHES=1).  `clear()` therefore only switches buffers.
        self.out_warm, self.out_timed = self.out, torch.zeros((C * FRAMES_PER_CUT, NUM_MELS), dtype=torch.float32, device=dev)
        offs = np.arange(C, dtype=np.int64) * SAMPLES_PER_CUT
        lens = np.full(C, SAMPLES_PER_CUT, dtype=np.int64)
        h = np.zeros(1, dtype=np.uint64)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        L.check("hipfeat_layout_create", self.plan.handle, C, _lib.addr(offs), _lib.addr(lens), None, None, NUM_MELS, self.stream, _lib.addr(h))
        self.layout = int(h[0])
        assert L.raw("hipfeat_layout_total_frames", self.layout) == C * FRAMES_PER_CUT
        self.units = C
        self.audio_seconds = 10.0 * C
        self.algo_bytes = ALGO_BYTES_PER_CUT * C
        # the package takes its time to reach the steady power state after the idle phases of start-up (plan creation, RNG fill): the first
        # launches run 1-4 % slower, for 10 to >100 launches depending on the box (tools/launch_ramp.py; the driver's 5 warm-ups + 20 steps would
        # sit entirely inside that ramp).  The metric is SUSTAINED extraction throughput, so the device is brought there before the contract's
        # own warm-up / timed steps begin (`settle_device`, called by main() directly in front of the warm-up); `config.settle` says so.
        self.settle = SETTLE_LAUNCHES
        self.kernel = self.plan.kernel_name
        self.workload = (f"BASELINE configs[1]: {C} x 10 s 16 kHz mono cuts per GPU per step, 80-dim log-mel Fbank (25/10 ms, povey, no dither), "
                         "device-resident float32 in / float32 out")
        if total:
            self.workload = (f"BASELINE configs[2]: {total} x 10 s 16 kHz mono cuts in total per step, sharded round-robin over {world} GPU(s) "
                             f"(rank r takes cuts r, r + {world}, ...: {C} on rank {rank}), 80-dim log-mel Fbank (25/10 ms, povey, no dither), "
                             "device-resident float32 in / float32 out")

    def step(self):
        self.L.check("hipfeat_extract_layout", self.plan.handle, self.layout, self.wave.data_ptr(), self.out.data_ptr(), self.stream)

    def settle_device(self):
        """Untimed launches right in front of the contract's warm-up (no synchronisation in between): see `settle` in __init__."""
        for _ in range(self.settle):
            self.step()

    def clear(self):
        self.out = self.out_timed  # (no device work: see __init__)

    def parity(self, rank):
        from oracle.kaldi_ref import RefConfig, RefExtractor
        from oracle.kaldi_torch import reference_f32

        np = self.np
        chk = self.out[:FRAMES_PER_CUT].float()
        assert self.torch.isfinite(chk).all() and float(chk.std()) > 0.1
        idx = fbank16k_parity_indices(self.C, rank)
        # ref32 = the reference's own float32 torch call sequence; numpy32 = kaldi_ref's float32 mode (float64 FFT rounded down: the
        # ref32 of rounds 1-4, carried side by side); f64 = truth
        r32, n32, o64 = reference_f32(RefConfig(kind="fbank")), RefExtractor(RefConfig(kind="fbank"), np.float32), RefExtractor(RefConfig(kind="fbank"), np.float64)
        stats = []
        for i in idx:
            x = self.wave[int(i)].cpu().numpy()
            got = self.out[int(i) * FRAMES_PER_CUT : (int(i) + 1) * FRAMES_PER_CUT].cpu().numpy()
            want, truth = r32.extract(x), o64.extract(x)
            assert got.shape == want.shape, (got.shape, want.shape)
            stats.append(compare(got, want, truth, alt32=n32.extract(x)))
        return fold(stats)

    def extra(self, args):
        return {} if args.no_host_fed else {"host_fed_cuts_per_s": host_fed(self.ex)}

    def close(self):
        self.L.check("hipfeat_layout_destroy", self.layout)


def libri_like_lengths(n: int, seed: int):
    """LibriSpeech-960-like utterance lengths in samples (seeded log-normal clipped to [1, 35] s, mean ~12.3 s;
    corpus statistics from general knowledge -- the corpus is not available offline)."""
    import numpy as np

    rs = np.random.RandomState(seed)
    dur = np.clip(np.exp(rs.randn(n) * 0.45 + 2.42), 1.0, 35.0)
    return np.round(dur * SR).astype(np.int64)


class Mfcc40Libri:
    """BASELINE configs[3]."""

    name = "mfcc40_libri"
    metric = "cuts/sec (LibriSpeech-like 1-35 s @16 kHz -> 40-dim MFCC)"
    default_cuts = 8000
    cpu_mode, cpu_what = "mfcc40", "Mfcc(40 filters, 40 cepstra)"
    F = 40

    def __init__(self, dev, rank, args):
        import numpy as np
        import torch

        import lhotse_amd
        from lhotse_amd import _lib

        self.torch, self.np = torch, np
        C = self.C = args.cuts or self.default_cuts
        self.ex = lhotse_amd.HipMfcc(lhotse_amd.HipMfccConfig(num_filters=40, num_ceps=40, cepstral_lifter=22, device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        L = self.L = self.plan.lib
        self.lens = libri_like_lengths(C, 1000 + rank)
        step = (self.lens + 3) & ~3  # every cut starts on a 16-byte boundary (as pack_to_device lays batches out)
        self.offs = np.concatenate([[0], np.cumsum(step)[:-1]]).astype(np.int64)
        total = int(self.offs[-1] + self.lens[-1])
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        self.wave = torch.empty(total, dtype=torch.float32, device=dev)
        for i in range(0, total, 1 << 26):
            self.wave[i : i + (1 << 26)].uniform_(-0.5, 0.5, generator=g)
        self.frames = (self.lens + 80) // 160
        self.rows = np.concatenate([[0], np.cumsum(self.frames)]).astype(np.int64)
        self.out = torch.empty((int(self.rows[-1]), self.F), dtype=torch.float32, device=dev)
        h = np.zeros(1, dtype=np.uint64)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        L.check("hipfeat_layout_create", self.plan.handle, C, _lib.addr(self.offs), _lib.addr(self.lens), None, None, self.F, self.stream, _lib.addr(h))
        self.layout = int(h[0])
        assert L.raw("hipfeat_layout_total_frames", self.layout) == int(self.rows[-1])
        self.units = C
        self.audio_seconds = float(self.lens.sum()) / SR
        self.algo_bytes = int(self.lens.sum()) * 4 + int(self.rows[-1]) * self.F * 4
        self.settle = SETTLE_LAUNCHES  # as for the headline workload (settle_device)
        self.kernel = self.plan.kernel_name
        self.workload = (f"BASELINE configs[3] stand-in: {C} cuts per GPU per step with LibriSpeech-like lengths (log-normal, 1-35 s, mean "
                         f"{self.audio_seconds / C:.1f} s; the corpus is not available offline), 40-dim MFCC (40 mel filters, 40 cepstra, lifter 22), "
                         "packed ragged batch, device-resident float32 in / float32 out")

    def step(self):
        self.L.check("hipfeat_extract_layout", self.plan.handle, self.layout, self.wave.data_ptr(), self.out.data_ptr(), self.stream)

    def settle_device(self):
        """Untimed launches right in front of the contract's warm-up (no synchronisation in between): see `settle` in __init__."""
        for _ in range(self.settle):
            self.step()

    def clear(self):
        self.out.zero_()

    def parity(self, rank):
        from oracle.kaldi_ref import RefConfig, RefExtractor
        from oracle.kaldi_torch import reference_f32

        np = self.np
        rs = np.random.RandomState(4321 + rank)
        idx = np.sort(rs.choice(self.C, size=min(PARITY_CUTS, self.C), replace=False))
        rc = RefConfig(kind="mfcc", num_filters=40, num_ceps=40, cepstral_lifter=22)
        o32, o64 = reference_f32(rc), RefExtractor(rc, np.float64)  # ref32 = Wav2MFCC's own float32 torch calls (oracle/kaldi_torch.TorchMfcc)
        stats = []
        for i in idx:
            o, n = int(self.offs[i]), int(self.lens[i])
            x = self.wave[o : o + n].cpu().numpy()
            got = self.out[int(self.rows[i]) : int(self.rows[i + 1])].cpu().numpy()
            want, truth = o32.extract(x), o64.extract(x)
            assert got.shape == want.shape, (got.shape, want.shape)
            stats.append(compare(got, want, truth, log_mel=False))  # cepstra: no linear-domain reading
        return fold(stats)

    def extra(self, args):
        return {}

    def close(self):
        self.L.check("hipfeat_layout_destroy", self.layout)


class OnTheFly:
    """BASELINE configs[4]: speed perturbation + Fbank + collation per 600 s mini-batch."""

    name = "onthefly"
    metric = "cuts/sec (1-30 s @16 kHz, speed-perturb 0.9/1.0/1.1 -> 80-dim log-mel fbank, collated per 600 s mini-batch)"
    default_cuts = 64  # mini-batches in the pool
    cpu_mode, cpu_what = "onthefly", "Speed + Fbank"

    def __init__(self, dev, rank, args):
        import numpy as np
        import torch

        import lhotse_amd
        from lhotse_amd import augmentation as A

        self.torch, self.np, self.A, self.dev, self.rank = torch, np, A, dev, rank
        NB = self.NB = args.cuts or self.default_cuts
        self.ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        rng = np.random.RandomState(rank)  # rank 0 = seed 0
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        K = self.K = max(1, int(getattr(args, "prefetch", 1) or 1))
        assert NB % K == 0, "--cuts (mini-batches per step) must be a multiple of --prefetch"
        minis = []
        ncuts = 0
        for b in range(NB):
            lens, tot = [], 0.0
            while True:
                d = rng.uniform(1.0, 30.0)
                if tot + d > 600.0:
                    break
                lens.append(int(d * SR))
                tot += d
            minis.append((np.asarray(lens, dtype=np.int64), rng.choice([0.9, 1.0, 1.1], size=len(lens))))
            ncuts += len(lens)
        # `batches`: what ONE call serves -- a mini-batch, or (--prefetch K) the K mini-batches a prefetching loader has packed into one arena
        self.batches = []
        for b0 in range(0, NB, K):
            lens = np.concatenate([m[0] for m in minis[b0 : b0 + K]])
            fac = np.concatenate([m[1] for m in minis[b0 : b0 + K]])
            sizes = np.asarray([len(m[0]) for m in minis[b0 : b0 + K]], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum((lens + 3) & ~3)[:-1]]).astype(np.int64)
            front = int(offs[-1] + lens[-1])
            arena = torch.empty(((front + 3) & ~3) + A.perturbed_tail_floats(lens, fac, SR), dtype=torch.float32, device=dev)
            arena[:front].uniform_(-0.5, 0.5, generator=g)
            self.batches.append({"arena": arena, "offs": offs, "lens": lens, "fac": fac, "front": front, "idx": None, "sizes": sizes if K > 1 else None,
                                 "first": np.concatenate([[0], np.cumsum(sizes)])})
        self.units = ncuts
        self.feats = [None] * len(self.batches)
        # the resamplers of the three factors resident in one bank: ONE launch perturbs a whole mini-batch, whatever its mix of factors,
        # fills the padding rows and carries the descriptor tables of the feature launch in its kernel arguments (hipfeat_minibatch_*)
        self.bank = A.HipSpeedBank([0.9, 1.0, 1.1], SR, dev)
        for bt in self.batches:
            bt["idx"] = self.bank.index_of(bt["fac"])
        self.nstreams = max(1, int(getattr(args, "streams", 3) or 3))
        # mini-batches alternate between a few streams, as a prefetching loader's would: the ramp and tail of one mini-batch's launches
        # overlap with the next one's (a 600 s mini-batch is too small to keep 256 CUs busy from its first workgroup to its last)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.nstreams)] if self.nstreams > 1 else []
        self.route = getattr(args, "route", "pair")
        self.step()  # sizes of the perturbed batch (for the byte count) and the first outputs
        in_samples = out_samples = frames = audio_in = all_out = pad_rows = 0
        for bt, (f, fl, po, pl) in zip(self.batches, self.feats):
            pad_rows += sum(int(t.shape[0] * t.shape[1]) for t in (f if isinstance(f, list) else [f])) - int(fl.sum())
            pert = bt["fac"] != 1.0
            in_samples += int(bt["lens"][pert].sum())
            out_samples += int(pl[pert].sum())
            frames += int(fl.sum())
            audio_in += int(bt["lens"].sum())
            all_out += int(pl.sum())
        self.audio_seconds = audio_in / SR
        # resampler: reads the perturbed cuts' inputs, writes their outputs; fbank: reads every (perturbed) cut once, writes its rows once
        self.algo_bytes = 4 * (in_samples + out_samples) + 4 * all_out + 4 * NUM_MELS * frames
        # ... and priced as ONE pass: every input sample read once, every feature written once (the perturbed waveforms in between
        # are this implementation's intermediate, not algorithmic traffic) -- roofline.frac_end_to_end
        self.algo_bytes_end_to_end = 4 * audio_in + 4 * NUM_MELS * frames
        self.algo_parts = {"resampler_read": 4 * in_samples, "resampler_write": 4 * out_samples, "feature_read": 4 * all_out, "feature_write": 4 * NUM_MELS * frames,
                           # the LOG_EPSILON rows behind every cut of the padded (B, Tmax, 80) tensors: written by the prep launch; part of what the
                           # API returns, NOT counted in `algorithmic_bytes` (the conservative reading: features once)
                           "padding_rows_write_not_in_algorithmic_bytes": 4 * NUM_MELS * pad_rows,
                           "feature_launches_per_step": len(self.batches)}
        self.kernel = self.plan.kernel_name + " + minibatch_prep_" + ("inline_" if K == 1 else "") + "kernel (mixed-factor resample_fast_block + padding rows + descriptor tables)"
        self.workload = (f"BASELINE configs[4]: {NB} mini-batches of 600 s per GPU per step ({ncuts} cuts U(1,30) s, {self.audio_seconds:.0f} s of audio), "
                         "each cut speed-perturbed by 0.9 / 1.0 / 1.1 on the device, then 80-dim log-mel Fbank written straight into the padded "
                         "(B, Tmax, 80) batch tensor (LOG_EPSILON padding); waveforms resident in HBM, features stay on the device; two launches "
                         + (f"per mini-batch" if K == 1 else f"per {K} mini-batches (a loader that prefetches {K}: every mini-batch its own dense tensor)")
                         + f", calls alternating over {max(1, self.nstreams)} stream(s)")

    def step(self):
        if self.route == "per_factor":  # round 3's route: one resample launch per distinct factor, then hipfeat_extract_collated
            A = self.A
            assert self.K == 1
            for k, bt in enumerate(self.batches):
                po, pl = A.perturb_speed_in_arena(bt["arena"], bt["offs"], bt["lens"], bt["fac"], SR, bt["front"])
                f, fl = self.plan.run_collated(bt["arena"], po, pl, None, LOG_EPSILON)
                self.feats[k] = (f, fl, po, pl)
            return
        torch, bank, plan = self.torch, self.bank, self.plan
        if not self.streams:
            for k, bt in enumerate(self.batches):
                self.feats[k] = bank.extract_collated(plan, bt["arena"], bt["offs"], bt["lens"], bt["idx"], bt["front"], LOG_EPSILON, group_sizes=bt["sizes"])
            return
        main = torch.cuda.current_stream(self.dev)
        fork = torch.cuda.Event()
        fork.record(main)
        for s in self.streams:
            s.wait_event(fork)
        try:
            for k, bt in enumerate(self.batches):
                torch.cuda.set_stream(self.streams[k % self.nstreams])  # (allocations of the outputs belong to the stream that fills them)
                self.feats[k] = bank.extract_collated(plan, bt["arena"], bt["offs"], bt["lens"], bt["idx"], bt["front"], LOG_EPSILON, group_sizes=bt["sizes"])
        finally:
            torch.cuda.set_stream(main)
        for s in self.streams:  # join: the step ends on the launch stream, where bench.py's events are recorded
            e = torch.cuda.Event()
            e.record(s)
            main.wait_event(e)

    def clear(self):
        for ft in self.feats:
            if ft is not None:
                for t in (ft[0] if isinstance(ft[0], list) else [ft[0]]):
                    t.zero_()

    def parity(self, rank):
        from oracle import resample_ref as R
        from oracle.kaldi_ref import RefConfig, RefExtractor

        from oracle.kaldi_torch import TorchSpeed, reference_f32

        np = self.np
        rs = np.random.RandomState(4321 + rank)
        # ref32 = the reference's own float32 torch calls end to end: Speed (F.pad + conv1d(stride), resample.py:284-315) then Fbank
        o32, o64 = reference_f32(RefConfig(kind="fbank")), RefExtractor(RefConfig(kind="fbank"), np.float64)
        sp32 = {f: TorchSpeed(SR, f) for f in (0.9, 1.1)}
        stats = []
        for _ in range(PARITY_CUTS):
            b = int(rs.randint(len(self.batches)))
            bt = self.batches[b]
            f, fl, po, pl = self.feats[b]
            i = int(rs.randint(len(bt["lens"])))
            if isinstance(f, list):  # --prefetch K: the mini-batch of cut i, and the cut's row in that mini-batch's tensor
                j = int(np.searchsorted(bt["first"], i, side="right")) - 1
                f, row = f[j], i - int(bt["first"][j])
            else:
                row = i
            x = bt["arena"][int(bt["offs"][i]) : int(bt["offs"][i]) + int(bt["lens"][i])].cpu().numpy()
            fac = float(bt["fac"][i])
            y32 = sp32[fac](x) if fac != 1.0 else x
            y64 = R.speed(x.astype(np.float64), SR, fac, np.float64) if fac != 1.0 else x.astype(np.float64)
            assert len(y32) == int(pl[i]), (len(y32), int(pl[i]))
            want, truth = o32.extract(y32), o64.extract(y64)
            got = f[row, : int(fl[i])].cpu().numpy()
            assert got.shape == want.shape, (got.shape, want.shape)
            assert bool((f[row, int(fl[i]) :] == LOG_EPSILON).all()), "padding rows of the collated batch"
            stats.append(compare(got, want, truth))
        return fold(stats)

    def _rate(self, seconds: float = 1.0):
        """(cuts/s, host microseconds per mini-batch) of the current route.  The host figure is the wall time of enqueueing a burst of
        8 mini-batches right after the device has drained (nothing to wait for: fewer calls than the library has table slots), i.e.
        Python + ctypes + the HIP launch calls themselves."""
        torch = self.torch
        self.step()
        torch.cuda.synchronize(self.dev)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            self.step()
            n += 1
            if n % 4 == 0:
                torch.cuda.synchronize(self.dev)
        torch.cuda.synchronize(self.dev)
        dt = time.perf_counter() - t0
        keep = self.batches
        burst = max(1, 8 // self.K)
        host = []
        for i in range(0, len(keep) - burst + 1, burst):
            self.batches = keep[i : i + burst]
            torch.cuda.synchronize(self.dev)
            h0 = time.perf_counter()
            self.step()
            host.append((time.perf_counter() - h0) / (burst * self.K))
        self.batches = keep
        self.step()  # (self.feats back in step with self.batches)
        torch.cuda.synchronize(self.dev)
        host.sort()
        return round(self.units * n / dt, 1), round(host[len(host) // 2] * 1e6, 2)  # (per MINI-BATCH, whatever the prefetch depth)

    def extra(self, args):
        """Same-call A/B of the routes (device-resident), the host's share per mini-batch, and the host-fed rate."""
        import copy

        out = {}
        keep = (self.route, self.streams, self.nstreams)
        ab = {}
        for name, route, ns, K in (("pair_3_streams", "pair", 3, 1), ("pair_2_streams", "pair", 2, 1), ("pair_1_stream", "pair", 1, 1),
                                   ("pair_4_minibatches_per_call_2_streams", "pair", 2, 4), ("per_factor_route_of_round_3", "per_factor", 1, 1)):
            w = self
            if K != self.K:
                a2 = copy.copy(args)
                a2.prefetch, a2.cuts = K, self.NB
                w = OnTheFly(self.dev, self.rank, a2)
            w.route, w.nstreams = route, ns
            w.streams = [self.torch.cuda.Stream(device=self.dev) for _ in range(ns)] if ns > 1 else []
            r, h = w._rate()
            ab[name] = {"cuts_per_s": r, "host_us_per_minibatch": h}
            if w is not self:
                del w
        self.route, self.streams, self.nstreams = keep
        ab["what"] = ("device-resident, ~1 s each in this run: `pair` = hipfeat_minibatch_plan + _run (two launches, tables in the kernel arguments), calls "
                      "alternating over 3 / 2 / 1 streams, one mini-batch per call or four (a loader that prefetches: every mini-batch still its own dense "
                      "tensor); `per_factor` = one hipfeat_resample launch per factor + hipfeat_extract_collated; host_us = median wall time per mini-batch of "
                      "enqueueing a burst of 8 mini-batches on a drained device (Python + ctypes + the HIP launch calls; includes the stream fork / join of a step)")
        out["routes"] = ab
        if not args.no_host_fed:
            out["host_fed"] = onthefly_host_fed(self)
        return out

    def close(self):
        pass


class BulkSave:
    """The offline path end to end on the GPU box (SURVEY 8d timing method iii; lhotse/cut/set.py:2307-2404): 600 s batches of HOST
    waveforms + the halves of every cut's manifest line (what the loader's worker processes hand over: `storage.manifest_fragments`)
    -> the H2D / kernel / D2H pipeline (`_batch_features_on_host`) on the calling thread -> background thread 1: the batch appended to
    the flat archive on tmpfs (libhipfeat's hipfeat_archive_append, striped over `--stripes` files) -> background thread 2: the batch's
    manifest lines spliced in libhipfeat (hipfeat_manifest_lines, which also enforces validate_features' frame-count contract), gzip,
    write, flush -- with the back-pressure of `storage.pump_batches`: the loop `compute_and_store_features_batch` runs, minus lhotse's
    loader and objects (lhotse cannot travel to the GPU box; the manifest half is tested against the reference driver, byte for byte
    against the per-cut path, in tests/test_lhotse_dropin.py).  PCIe-, host- and file-system-inclusive: NOT a device-resident rate."""

    name = "bulk_save"
    host_bound = True
    metric = "cuts/sec (10 s @16 kHz host waveforms -> 80-dim log-mel fbank -> hip_archive on tmpfs + manifests; PCIe- and host-inclusive)"
    default_cuts = 64  # batches of 60 cuts (600 s) per step
    cpu_mode, cpu_what = "", "Fbank"

    def __init__(self, dev, rank, args):
        import dataclasses
        import tempfile

        import numpy as np
        import torch

        import lhotse_amd
        from lhotse_amd import storage as S

        self.torch, self.np, self.S, self.dev, self.rank = torch, np, S, dev, rank
        NB = self.NB = args.cuts or self.default_cuts
        self.stripes = max(1, int(getattr(args, "stripes", 8) or 8))
        self.ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self.tmp = tempfile.TemporaryDirectory(prefix=f"hipfeat_bulk_r{rank}_", dir=base)
        self.fs = "tmpfs (/dev/shm)" if base else "the default temporary directory"

        @dataclasses.dataclass
        class Sup:  # the JSON-scalar fields of a SupervisionSegment (lhotse/supervision.py:44-120)
            id: str
            recording_id: str
            start: float
            duration: float
            channel: int = 0
            text: str = None
            language: str = None
            speaker: str = None
            gender: str = None
            custom: dict = None
            alignment: dict = None

        @dataclasses.dataclass
        class Src:
            type: str
            channels: list
            source: str

        @dataclasses.dataclass
        class Rec:  # Recording (lhotse/audio/recording.py:59-120)
            id: str
            sources: list
            sampling_rate: int
            num_samples: int
            duration: float
            channel_ids: list = None
            transforms: list = None

        class Cut:  # what storage._mono_cut_dict reads off a MonoCut
            __slots__ = ("id", "start", "duration", "channel", "recording_id", "supervisions", "custom", "recording", "sampling_rate")

        g = torch.Generator().manual_seed(99 + rank)
        pool = [(torch.rand(SAMPLES_PER_CUT, generator=g) - 0.5) for _ in range(64)]  # pageable float32, as a DataLoader hands them over
        self.pool16 = [(x * 32767).to(torch.int16) for x in pool]
        self.pool32 = pool
        self.cut_batches = []
        n = 0
        for b in range(NB):
            cuts = []
            for i in range(60):
                c = Cut()
                c.id, c.start, c.duration, c.channel, c.recording_id, c.sampling_rate = f"cut-{rank}-{n:07d}", 0.0, SAMPLES_PER_CUT / SR, 0, f"rec-{rank}-{n:07d}", SR
                c.supervisions = [Sup(id=c.id, recording_id=c.recording_id, start=0.0, duration=c.duration, text="SYNTHETIC UTTERANCE " * 4, language="English", speaker=f"spk{n % 251}")]
                c.custom = {"dataloading_info": {"rank": 0, "world_size": 1, "worker_id": None}}  # what lhotse's sampler attaches (sampling/base.py:473-487)
                c.recording = Rec(id=c.recording_id, sources=[Src("file", [0], f"/data/corpus/{c.recording_id}.flac")], sampling_rate=SR,
                                  num_samples=SAMPLES_PER_CUT, duration=c.duration, channel_ids=[0])
                cuts.append(c)
                n += 1
            self.cut_batches.append((cuts, [(b * 60 + i) % 64 for i in range(60)]))
        # the loader's side of the manifests: the two halves of every cut's line (in the product they are made in the DataLoader's worker
        # processes, next to audio decoding; here once, like the waveforms -- `fragments_per_s_per_process` in `extra` says what one costs)
        self.template = {"type": self.ex.name, "num_features": NUM_MELS, "frame_shift": self.ex.frame_shift, "sampling_rate": SR, "storage_type": "hip_archive",
                         "storage_path": ""}
        rc = {}
        t0 = time.perf_counter()
        self.batches = [(cuts, idx, [S.manifest_fragments(c, self.template, self.ex.frame_shift, rc) for c in cuts]) for cuts, idx in self.cut_batches]
        self.fragments_per_s = n / (time.perf_counter() - t0)
        assert all(f is not None for _, _, fr in self.batches for f in fr)
        self.units = n
        self.audio_seconds = SAMPLES_PER_CUT / SR * n
        self.algo_bytes = ALGO_BYTES_PER_CUT * n
        self.kernel = self.plan.kernel_name
        self.stats = {}
        self.variant = ("float32", "hip_archive", "native")
        self.workload = (f"SURVEY 8d(iii) offline path: {NB} batches of 60 x 10 s (600 s) pageable float32 host waveforms per GPU per step -> chunked H2D / "
                         f"fft512c / D2H pipeline -> background thread 1: hipfeat_archive_append into a 'hip_archive' of {self.stripes} file(s) on {self.fs} -> "
                         "background thread 2: hipfeat_manifest_lines (key splice + frame-count contract) + gzip JSONL + flush per batch; back-pressure of 8 batches "
                         "-- compute_and_store_features_batch's loop without lhotse's loader (which hands over waveforms AND line halves)")
        self._run_id = 0

    def _one_pass(self, dtype: str, storage: str, stats: dict, route: str = "native"):
        """One pass over the batches.  route "native": NativeArchive + spliced lines (the product's path for the hip_archive storages);
        "per_cut": round 4's path -- write_packed by Python + one template dict + json.dumps per cut -- kept for the A/B."""
        import gzip
        import json

        S, np = self.S, self.np
        half = storage == "hip_archive_f16"
        pool = self.pool16 if dtype == "int16" else self.pool32
        self._run_id += 1
        root = os.path.join(self.tmp.name, f"run{self._run_id}")
        os.makedirs(root)
        busy = {"save": 0.0, "lines": 0.0, "n": 0}
        ex = self.ex
        pipe0 = ex._native_pipe().stats() if route == "native" and hasattr(getattr(ex, "plan", None), "handle") else None

        def extract(batch):
            cuts, idx, frags = batch
            if route == "native":  # the library's host pipeline: packed and ENQUEUED here, waited for on the archive thread
                pending, frames = S._batch_features_pending(ex, [pool[i] for i in idx], SR, None, half=half)
                return cuts, frags, pending, frames
            host, frames = S._batch_features_on_host(ex, [pool[i] for i in idx], SR, None, half=half)  # round 4: the Python pipeline, synchronous
            return cuts, frags, host, frames

        with gzip.open(os.path.join(root, "cuts.jsonl.gz"), "wb") as manifest:
            if route == "native":
                with S.NativeArchive(os.path.join(root, "feats"), mode="w", np_dtype="<f2" if half else "<f4", stripes=self.stripes, name=storage) as ar:

                    def save(cuts, frags, pending, frames):
                        t0 = time.perf_counter()
                        host = pending.wait()
                        t1 = time.perf_counter()
                        fr = np.ascontiguousarray(frames, dtype=np.int64)
                        file_of, byte_off = ar.append(host, fr)
                        del host
                        pending.release()
                        busy["wait"] = busy.get("wait", 0.0) + t1 - t0
                        busy["save"] += time.perf_counter() - t1
                        return frags, fr, file_of, byte_off

                    def lines(frags, fr, file_of, byte_off):
                        t0 = time.perf_counter()
                        blob = ar.lines([f[0] for f in frags], [f[1] for f in frags], fr, np.fromiter((f[2] for f in frags), dtype=np.int64, count=len(frags)),
                                        file_of, byte_off, NUM_MELS)
                        manifest.write(blob)
                        manifest.flush()
                        busy["lines"] += time.perf_counter() - t0
                        busy["n"] += len(frags)

                    S.pump_batches(self.batches, extract, save, stats=stats, finish=lines)
                    stats["archive_bytes"] = stats.get("archive_bytes", 0) + sum(ar.size(k) for k in range(self.stripes))
            else:
                wcls = S.HipArchiveF16Writer if half else S.HipArchiveWriter
                rec_cache = {}
                with wcls(os.path.join(root, "feats"), mode="w") as writer:
                    template = dict(self.template, storage_type=writer.name, storage_path=str(writer.storage_path))

                    def save(cuts, frags, host, frames):
                        t0 = time.perf_counter()
                        for c, t in zip(cuts, frames):  # the frame-count contract of validate_features (lhotse/qa.py:286-301)
                            if S.expected_num_frames(c.duration, ex.frame_shift, SR) != t:
                                raise AssertionError(f"cut {c.id}: {t} frames")
                        keys = writer.write_packed(host, frames)
                        writer.flush()
                        busy["save"] += time.perf_counter() - t0
                        return cuts, frames, keys

                    def lines(cuts, frames, keys):
                        t0 = time.perf_counter()
                        for c, t, k in zip(cuts, frames, keys):
                            manifest.write((json.dumps(S._mono_cut_dict(c, S._features_dict(template, c, t, k), rec_cache), ensure_ascii=False) + "\n").encode())
                        manifest.flush()
                        busy["lines"] += time.perf_counter() - t0
                        busy["n"] += len(cuts)

                    S.pump_batches(self.batches, extract, save, stats=stats, finish=lines)
                    stats["archive_bytes"] = stats.get("archive_bytes", 0) + os.path.getsize(writer.storage_path)
        if pipe0 is not None:  # the library's pipeline thread: its own clock over this pass
            pipe1 = ex._native_pipe().stats()
            for k in ("busy_s", "pack_s", "device_backpressure_s"):
                stats["pipe_" + k] = stats.get("pipe_" + k, 0.0) + pipe1[k] - pipe0[k]
        stats["device_wait_s"] = stats.get("device_wait_s", 0.0) + busy.get("wait", 0.0)
        stats["manifest_s"] = stats.get("manifest_s", 0.0) + busy["lines"]
        stats["save_s"] = stats.get("save_s", 0.0) + busy["save"]
        stats["manifest_lines"] = stats.get("manifest_lines", 0) + busy["n"]
        stats["manifest_bytes"] = stats.get("manifest_bytes", 0) + os.path.getsize(os.path.join(root, "cuts.jsonl.gz"))
        self.last_root = root
        return root

    def _drop(self, root):
        """Delete a finished run on a helper thread: freeing gigabytes of tmpfs pages takes a few 100 ms, which the product's loop never
        spends (it keeps its archive) -- it must not sit between two timed passes."""
        import shutil
        from concurrent.futures import ThreadPoolExecutor

        if getattr(self, "_dropper", None) is None:
            self._dropper = ThreadPoolExecutor(max_workers=1, thread_name_prefix="bulk-drop")
        self._dropper.submit(shutil.rmtree, root, True)

    def step(self):
        prev = getattr(self, "last_root", None)
        self._one_pass(self.variant[0], self.variant[1], self.stats, self.variant[2])
        if prev:
            self._drop(prev)

    def clear(self):
        self.stats.clear()

    def parity(self, rank):
        """What the last timed pass stored, read back through the archive reader(s), against the oracle."""
        import gzip
        import json

        from oracle.kaldi_ref import RefConfig, RefExtractor
        from oracle.kaldi_torch import reference_f32

        np, S = self.np, self.S
        o32, o64 = reference_f32(RefConfig(kind="fbank")), RefExtractor(RefConfig(kind="fbank"), np.float64)
        with gzip.open(os.path.join(self.last_root, "cuts.jsonl.gz"), "rt") as f:
            lines = [json.loads(ln) for ln in f]
        assert len(lines) == self.units and [d["id"] for d in lines[:3]] == [c.id for c in self.batches[0][0][:3]]
        readers = {}
        rs = np.random.RandomState(4321 + rank)
        stats = []
        flat = [i for _, idx, _ in self.batches for i in idx]
        for j in rs.choice(self.units, size=min(PARITY_CUTS, self.units), replace=False):
            d = lines[int(j)]
            assert d["features"]["num_frames"] == FRAMES_PER_CUT and d["features"]["storage_type"] == "hip_archive" and d["type"] == "MonoCut"
            path = d["features"]["storage_path"]
            reader = readers.get(path) or readers.setdefault(path, S.HipArchiveReader(path))
            got = reader.read(d["features"]["storage_key"])
            x = self.pool32[flat[int(j)]].numpy()
            stats.append(compare(got, o32.extract(x), o64.extract(x)))
        return fold(stats)

    def extra(self, args):
        """The other entry forms / storages and round 4's per-cut route, a few passes each, with the stage split of every variant."""
        out = {}
        keep_stripes = self.stripes
        for dtype, storage, route, stripes in (("float32", "hip_archive", "native", keep_stripes), ("int16", "hip_archive", "native", keep_stripes),
                                               ("float32", "hip_archive_f16", "native", keep_stripes), ("int16", "hip_archive_f16", "native", keep_stripes),
                                               ("float32", "hip_archive", "native", 1), ("int16", "hip_archive_f16", "native", 1),
                                               ("float32", "hip_archive", "per_cut", 1), ("int16", "hip_archive_f16", "per_cut", 1)):
            if route == "native" and stripes == 1 and keep_stripes == 1:
                continue
            self.stripes = stripes
            self._drop(self._one_pass(dtype, storage, {}, route))  # warm
            st = {}
            t0 = time.perf_counter()
            for _ in range(3):
                self._drop(self._one_pass(dtype, storage, st, route))
            dt = time.perf_counter() - t0
            cuts = 3 * self.units
            out[f"{dtype}->{storage}" + (f" ({stripes} file{'s' if stripes > 1 else ''})" if route == "native" else " (round 4's per-cut Python route, 1 file)")] = {
                "cuts_per_s": round(cuts / dt, 1), "archive_MB_per_s": round(st["archive_bytes"] / dt / 1e6, 1),
                "h2d_MB_per_s": round(cuts * SAMPLES_PER_CUT * (2 if dtype == "int16" else 4) / dt / 1e6, 1),
                "manifest_bytes_per_cut": round(st["manifest_bytes"] / cuts, 1),
                "main_thread_extract_share": round(st["extract_s"] / dt, 3), "main_thread_blocked_share": round(st["wait_s"] / dt, 3),
                "archive_thread_busy_share": round(st["save_s"] / dt, 3), "archive_thread_waiting_for_the_device_share": round(st.get("device_wait_s", 0.0) / dt, 3),
                "manifest_thread_busy_share": round(st["manifest_s"] / dt, 3),
                "pipeline_thread_busy_share": round(st.get("pipe_busy_s", 0.0) / dt, 3), "pipeline_thread_packing_share": round(st.get("pipe_pack_s", 0.0) / dt, 3),
                "pipeline_thread_waiting_for_pcie_or_device_share": round(st.get("pipe_device_backpressure_s", 0.0) / dt, 3),
                "binds": max((st["extract_s"], "the calling thread (packing into page-locked staging + enqueueing)"), (st["save_s"], "archive thread"),
                             (st.get("device_wait_s", 0.0), "PCIe / device (the archive thread waits for the batch's download)"), (st["manifest_s"], "manifest thread"),
                             (st.get("pipe_busy_s", 0.0), "the library's pipeline thread (packing into page-locked staging + enqueueing; its PCIe / device back-pressure included)"))[1],
            }
        self.stripes = keep_stripes
        out["fragments_per_s_per_process"] = round(self.fragments_per_s, 1)
        out["stripes"] = self.stripes
        out["what"] = ("3 passes per variant after one warm-up; shares are of wall time: the calling thread extracts (pack to pinned + H2D + kernel + D2H), "
                       "one background thread appends to the archive, a second one writes the manifest lines (storage.pump_batches), the library's pipeline thread packs and enqueues "
                       "(its own clock: hipfeat_host_pipeline_stats); the largest share binds; "
                       "fragments_per_s_per_process = manifest line halves one (loader) process serialises per second -- that work rides on the loader's workers, "
                       "next to audio decoding, not on this process")
        return {"bulk_save": out}

    def close(self):
        if getattr(self, "_dropper", None) is not None:
            self._dropper.shutdown(wait=True)
        self.tmp.cleanup()


class Plumbing:
    """BASELINE configs[0] / SURVEY 8d baseline C on the GPU box: 64 int16 WAV files on tmpfs -> decoding DataLoader worker processes ->
    features -> storage + gzip JSONL manifest (tools/plumbing.py, which cites the lhotse driver each leg keeps the structure of).  A step =
    ONE pass of the product's bulk driver over `repeat` x 64 cuts with the product's default loader, the shared-memory ring (leg D);
    `extra` carries leg C (the same driver behind a torch DataLoader), leg B (lhotse's batch-driver structure with lhotse's own per-cut
    .npy save path around HipFbank), several processes sharing the GPU and, as `cpu_baseline`, leg A (the reference's per-cut CPU
    driver restated: num_jobs = 1 and num_jobs = the container's CPU quota).  lhotse itself cannot run here (not installed on the box; a Python reference cannot travel):
    the REAL drivers are timed next to legs A in the authoring container, profiles/r06_plumbing_container.json."""

    name = "plumbing"
    host_bound = True
    metric = "cuts/sec (64 x 10 s 16 kHz int16 WAV files on tmpfs -> decode -> 80-dim log-mel fbank -> storage + manifest; decode-, PCIe- and host-inclusive)"
    default_cuts = 400  # passes over the 64 files per step (25 600 cuts: ~1 s at the ring loader's rate)
    cpu_mode, cpu_what = "", "Fbank"

    def __init__(self, dev, rank, args):
        import tempfile

        import numpy as np
        import torch

        import lhotse_amd

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import plumbing as P

        self.P, self.np, self.torch, self.rank = P, np, torch, rank
        self.repeat = args.cuts or self.default_cuts
        self.workers = int(os.environ.get("BENCH_LOADER_WORKERS", "0")) or P.default_workers()
        self.stripes = max(1, int(getattr(args, "stripes", 8) or 8))
        self.ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self.tmp = tempfile.TemporaryDirectory(prefix=f"hipfeat_plumb_r{rank}_", dir=base)
        self.fs = "tmpfs (/dev/shm)" if base else "the default temporary directory"
        self.paths = P.write_corpus(os.path.join(self.tmp.name, "wav"), 64, seed=rank)
        self.cuts = P.make_cuts(self.paths, self.repeat)
        self.units = len(self.cuts)
        self.audio_seconds = 10.0 * self.units
        self.algo_bytes = ALGO_BYTES_PER_CUT * self.units
        self.kernel = self.plan.kernel_name
        self.settle = 0
        self._run = 0
        self.last = None
        self.quota = cpu_quota()
        self.workload = (f"BASELINE configs[0] with the GPU in it: 64 x 10 s 16 kHz mono int16 WAV files on {self.fs}, visited {self.repeat} x per step "
                         f"({self.units} cuts, 600 s batches) -> {self.workers} loader worker processes decode (stdlib wave, int16 / 32768) straight into the slots of "
                         "one shared-memory ring and serialise the manifest line halves -> hipfeat_host_pipeline uploads out of the (page-locked) slots -> "
                         f"hip_archive striped over {self.stripes} file(s) + gzip JSONL manifest flushed per batch (the product's bulk driver with its default loader; "
                         f"lhotse's own drivers cannot run on this box, see extra.plumbing.what); container CPU quota: {self.quota} of {os.cpu_count()} logical CPUs")

    def _dir(self, tag):
        self._run += 1
        return os.path.join(self.tmp.name, f"{tag}{self._run}")

    def step(self):
        import shutil

        prev = self.last
        # this process holds a live HIP context: workers FORKED off it would slow every device round trip by ~30 ms while they live
        # (lhotse_amd/_lib.py, profiles/r06_loader_pipeline_probe.txt) -- the product's driver starts them through a fork server in that
        # situation, and so does this step
        self.last = self.P.hip_ring(self.ex, self.cuts, self._dir("ring"), self.workers, stripes=self.stripes, context="forkserver")
        if prev:
            shutil.rmtree(os.path.dirname(prev["manifest"]), ignore_errors=True)

    def clear(self):
        pass

    def parity(self, rank):
        """What the last timed pass stored, read back through the manifest + archive reader, against the oracle on the decoded files."""
        from oracle.kaldi_ref import RefConfig, RefExtractor
        from oracle.kaldi_torch import reference_f32

        np, P = self.np, self.P
        o32, o64 = reference_f32(RefConfig(kind="fbank")), RefExtractor(RefConfig(kind="fbank"), np.float64)
        rs = np.random.RandomState(4321 + rank)
        stats = []
        for j in rs.choice(self.units, size=min(16, self.units), replace=False):
            x = P.read_wav(self.cuts[int(j)].path)[0]
            stats.append(compare(P.read_back(self.last, int(j)), o32.extract(x), o64.extract(x)))
        return fold(stats)

    def _fresh(self, *flags, repeat=None):
        """One leg in a FRESH process (tools/plumbing.py main): the GPU is first touched at the first batch, after the loader's workers were
        forked -- the order lhotse's own driver produces.  -> the JSON lines of its passes."""
        import subprocess

        cmd = [sys.executable, os.path.join(ROOT, "tools", "plumbing.py"), "--wav-dir", os.path.dirname(self.paths[0]), "--repeat", str(repeat or self.repeat),
               "--stripes", str(self.stripes), *[str(f) for f in flags]]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HIPFEAT_NO_FORK_WARNING="1"))
        rows = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not rows:
            return {"error": (p.stderr or p.stdout)[-600:]}
        return rows

    def _shared_gpu(self, procs: int, workers: int, leg: str = "D", flags=(), label: str = "float32 -> hip_archive"):
        """Leg `leg` in `procs` FRESH processes that share this GPU (each with its own plan, pipeline, archive and `workers` loader workers),
        started together: what one GPU takes when the loader-bound driver is run as several processes -- the sharded driver accepts any
        number of ranks per device.  Aggregate = cuts of all processes behind their first batches / the span of their steady regions."""
        import subprocess

        start = time.time() + 12.0  # (imports and plan creation of all processes are over by then)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "plumbing.py"), "--leg", leg, "--wav-dir", os.path.dirname(self.paths[0]), "--repeat", str(getattr(self, "long", self.repeat) if leg == "D" else self.repeat),
               "--stripes", str(self.stripes), "--workers", str(workers), "--passes", "1", "--start-at", str(start), *flags]
        ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, HIPFEAT_NO_FORK_WARNING="1")) for _ in range(procs)]
        rows = []
        for p in ps:
            o, _ = p.communicate(timeout=900)
            rows += [json.loads(ln) for ln in o.splitlines() if ln.startswith("{")]
        key = f"{leg} {'hip_ring' if leg == 'D' else 'hip_bulk'} {label}, {procs} PROCESSES sharing the GPU x {workers} loader workers each"
        if len(rows) != procs:
            return {key: {"error": f"{len(rows)} of {procs} processes answered"}}
        t0, t1 = min(r["steady_region_epoch"][0] for r in rows), max(r["steady_region_epoch"][1] for r in rows)
        return {key: {"cuts_per_s": round(sum(r["steady_cuts"] for r in rows) / (t1 - t0), 1), "per_process_cuts_per_s": [r["cuts_per_s"] for r in rows],
                      "processes": procs, "num_workers": workers, "cuts": sum(r["cuts"] for r in rows)}}

    def extra(self, args, full: bool = True):
        import shutil

        P, out = self.P, {}
        keep = ("cuts_per_s", "cuts_per_s_incl_worker_start", "seconds_to_first_batch", "cuts", "num_workers", "worker_start", "transport", "input", "storage", "save_thread_busy_share",
                "ring_slots_page_locked", "batches_uploaded_straight_from_the_ring", "batches", "cpus_busy_by_thread_name", "container_cpus_busy",
                "container_cpus_busy_user_system", "container_cpu_quota", "quota_periods_throttled")

        def brief(rows, k=0):
            if isinstance(rows, dict):
                return rows
            r = rows[min(k, len(rows) - 1)]
            return {**{a: r[a] for a in keep if a in r}, **{a: v for a, v in r.items() if a.endswith("_share")}}

        W = self.workers
        small = max(4, self.repeat // 8)   # (legs that move 1-6 k cuts/s)
        # fresh processes, workers forked before the GPU is touched (pass 0 of each) -----------------------------------------------------
        for wk in sorted({4, W} if full else {W}):
            out[f"B hip_batch_numpy_files, {wk} loader workers"] = brief(self._fresh("--leg", "B", "--workers", wk, "--passes", 1, repeat=small))
        out[f"C hip_bulk float32 -> hip_archive, {W} loader workers (torch DataLoader, one packed tensor per batch)"] = \
            brief(self._fresh("--leg", "C", "--workers", W, "--passes", 1, repeat=max(4, self.repeat // 4)))
        # D legs: 64 000 cuts, ~2 s (every slot's first batch still goes through staging while it is being page-locked) -- 20 GB of archive per
        # process in the temporary directory: only where there is room for it
        try:
            st = os.statvfs(self.tmp.name)
            roomy = st.f_bavail * st.f_frsize > (120 << 30)
        except OSError:
            roomy = False
        self.long = long = max(self.repeat, 1000) if roomy else self.repeat
        out[f"D hip_ring float32 -> hip_archive, {W} loader workers (shared-memory ring, slots page-locked for the GPU)"] = brief(self._fresh("--leg", "D", "--workers", W, "--passes", 1, repeat=long))
        out[f"D hip_ring int16 -> hip_archive_f16, {W} loader workers"] = brief(self._fresh("--leg", "D", "--workers", W, "--passes", 1, "--pcm16", "--half", repeat=long))
        out[f"F hip_ring float32 -> lhotse's NumpyFilesWriter layout (one .npy + one manifest dict per cut), {W} loader workers"] = \
            brief(self._fresh("--leg", "F", "--workers", W, "--passes", 1, repeat=max(4, self.repeat // 2)))
        half_w = max(2, W * 3 // 4)
        out.update(self._shared_gpu(2, half_w, "D"))
        # the per-cut driver (leg A's loop) with HipFbank as a drop-in of Fbank: what changes for a user who changes only the extractor object
        for jobs in sorted({1, max(1, int(self.quota or 16) // 2)} if full else {max(1, int(self.quota or 16) // 2)}):
            rows = self._fresh("--leg", "E", "--jobs", jobs, repeat=max(4, self.repeat // (2 if jobs > 1 else 16)))
            out[f"E per-cut driver (compute_and_store_features structure) around HipFbank, num_jobs={jobs} processes sharing the GPU"] = \
                rows if isinstance(rows, dict) else {k: rows[0][k] for k in ("cuts_per_s", "per_process_cuts_per_s", "cuts", "num_jobs", "errors")}
        if full:
            out.update(self._shared_gpu(2, half_w, "D", ("--pcm16", "--half"), "int16 -> hip_archive_f16"))
            out[f"D hip_ring float32 -> hip_archive, {W} loader workers, staging copy kept (slots not page-locked)"] = \
                brief(self._fresh("--leg", "D", "--workers", W, "--passes", 1, "--no-pin", repeat=long))
            out[f"D hip_ring float32 -> hip_archive, {W} loader workers, fork server, GPU touched before"] = \
                brief(self._fresh("--leg", "D", "--workers", W, "--passes", 1, "--gpu-first", "--context", "forkserver", repeat=long))
            out[f"C hip_bulk int16 -> hip_archive_f16, {W} loader workers"] = brief(self._fresh("--leg", "C", "--workers", W, "--passes", 1, "--pcm16", "--half", repeat=max(4, self.repeat // 4)))
            out[f"C hip_bulk float32 -> hip_archive, {W} loader workers, one array per cut through the worker queue (lhotse's transport)"] = \
                brief(self._fresh("--leg", "C", "--workers", W, "--passes", 1, "--per-cut-transport", repeat=small))
            # the hazard: the same leg with the GPU touched BEFORE the workers are forked, and its remedy (fork server) ---------------------
            out[f"C hip_bulk float32 -> hip_archive, {W} loader workers, FORKED AFTER the GPU was touched"] = \
                brief(self._fresh("--leg", "C", "--workers", W, "--passes", 1, "--gpu-first", repeat=small))
            out[f"C hip_bulk float32 -> hip_archive, {W} loader workers, fork server, GPU touched before"] = \
                brief(self._fresh("--leg", "C", "--workers", W, "--passes", 1, "--gpu-first", "--context", "forkserver", repeat=max(4, self.repeat // 4)))
            out.update(self._shared_gpu(4, max(2, W // 2), "D"))
        if not args.no_cpu_baseline:
            ncpu = len(os.sched_getaffinity(0))
            for jobs in sorted({1, max(1, min(64, ncpu // 4 if self.quota is None else int(self.quota)))}):
                n = 64 * (2 if jobs == 1 else max(2, min(40, jobs)))
                d = self._dir("a")
                out[f"A cpu_per_cut (compute_and_store_features(Fbank(), NumpyFilesWriter, num_jobs={jobs}) restated; kind = port)"] = P.cpu_per_cut(P.make_cuts(self.paths, n // 64), d, jobs)
                shutil.rmtree(d, ignore_errors=True)
        out["what"] = ("cuts/s BEHIND the first batch (the start of the worker processes is reported next to it) of whole passes incl. WAV decode, storage and manifest; "
                       "shares are of wall time.  A = the reference's per-cut CPU driver restated with the reference's torch call sequence as the extractor "
                       "(oracle/kaldi_torch.py: the checker / baseline, never the product path); B = the structure of CutSet.compute_and_store_features_batch "
                       "(lhotse/cut/set.py:2296-2408) around HipFbank with lhotse's own save path (one .npy per cut, one json.dumps + flush per cut on ONE save "
                       "thread) and lhotse's transport (one array per cut through the worker queue); C = the product's bulk driver behind a torch DataLoader (one packed "
                       "tensor per batch); F = D's loader and pipeline in front of lhotse's own per-cut storage (lhotse_amd.compute_and_store_features_batch(storage_type=NumpyFilesWriter)); E = leg A's per-cut loop with HipFbank in place of the reference extractor (one cut per call, .npy + manifest line per cut, every job process its own plan); D = the product's bulk driver with its default loader, lhotse_amd/ring_loader.py (workers decode into slots of one shared "
                       "ring, the slots are page-locked for the GPU as they come into use and the host pipeline uploads straight out of them: ABI v5).  "
                       "B, C and D run in FRESH processes (tools/plumbing.py): the GPU is first touched at the first batch, after the workers were forked, as under "
                       "lhotse's driver.  container_cpus_busy / quota_periods_throttled: the whole container's CPU time over the leg against its cgroup quota "
                       "(the host-bound legs end at that quota, not at the GPU).  The C legs marked so show the fork hazard (lhotse_amd/_lib.py) and its remedy.  lhotse itself cannot run on this box; "
                       "the real drivers are timed next to leg A in the authoring container: profiles/r06_plumbing_container.json")
        return {"plumbing": out}

    def close(self):
        self.tmp.cleanup()


class _HostEvent:
    """Stand-in for torch.cuda.Event in the CPU self-test of the N > 1 plumbing (BENCH_SELFTEST_STUB=1): host clock."""

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other) -> float:
        return (other.t - self.t) * 1e3


def _sync(dev) -> None:
    if dev.type == "cuda":
        import torch

        torch.cuda.synchronize(dev)


def _event_pairs(dev, n: int):
    if dev.type != "cuda":
        return [(_HostEvent(), _HostEvent()) for _ in range(n)]
    import torch

    return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]


class SelfTestStub:
    """NOT a measurement: a workload without a device, selected ONLY by BENCH_SELFTEST_STUB=1, so that everything main() does around a
    workload at N > 1 -- group set-up, the contract's barriers, MAX / gather of the per-rank figures, the parity reduction, the host-fed
    gather, the assembly of the one JSON line -- can run at world_size 8 over gloo on a machine without GPUs (tests/test_bench_world.py;
    VERDICT r5 task 6: only the driver can launch 8 GPUs, so the path has to be right by construction)."""

    name = "selftest_stub"
    host_bound = True
    metric = "SELF-TEST (no device work): the N > 1 plumbing of bench.py"
    default_cuts = 10
    cpu_mode, cpu_what = "", "none"

    def __init__(self, dev, rank, args):
        self.rank, self.units, self.audio_seconds, self.algo_bytes = rank, 10, 100.0, ALGO_BYTES_PER_CUT * 10
        self.kernel, self.settle, self.workload = "selftest_stub", 0, "SELF-TEST stub (BENCH_SELFTEST_STUB=1): no device work"

    def step(self):
        time.sleep(0.001 * (1 + self.rank % 3))

    def clear(self):
        pass

    def parity(self, rank):
        import numpy as np

        rs = np.random.RandomState(rank)
        truth = rs.randn(50, 80) - 5.0
        want = (truth + 1e-6 * rs.randn(50, 80)).astype(np.float32)
        got = (want + np.float32(1e-6) * rs.randn(50, 80).astype(np.float32)).astype(np.float32)
        return fold([compare(got, want, truth)])

    def extra(self, args):
        return {"host_fed_cuts_per_s": {"batch_60": 100.0 * (self.rank + 1), "what": "self-test"}}

    def close(self):
        pass


WORKLOADS = {w.name: w for w in (Fbank16k, Mfcc40Libri, OnTheFly, BulkSave, Plumbing)}


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


def onthefly_host_fed(w, seconds: float = 3.0):
    """The same mini-batches starting as float32 numpy arrays in HOST memory (as decoded audio would): pack + one H2D into the arena ->
    speed perturbation -> fbank collated on the device.  PCIe-inclusive; never `value`."""
    import torch

    from lhotse_amd.extractors import pack_to_device

    A = w.A
    host = []
    for bt in w.batches[:16]:
        a = bt["arena"][: bt["front"]].cpu().numpy()
        host.append([a[int(o) : int(o) + int(n)].copy() for o, n in zip(bt["offs"], bt["lens"])])

    def one(k):
        bt = w.batches[k]
        packed, offs, lens = pack_to_device(host[k], w.dev)
        arena = torch.empty(((packed.numel() + 3) & ~3) + A.perturbed_tail_floats(lens, bt["fac"], SR), dtype=torch.float32, device=w.dev)
        arena[: packed.numel()].copy_(packed, non_blocking=True)
        return w.bank.extract_collated(w.plan, arena, offs, lens, bt["idx"], packed.numel(), LOG_EPSILON, group_sizes=bt["sizes"])

    one(0)
    torch.cuda.synchronize()
    n = cuts = 0
    secs = 0.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        k = n % len(host)
        one(k)
        cuts += len(host[k])
        secs += float(w.batches[k]["lens"].sum()) / SR
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"batches_per_s": round(n * w.K / dt, 1), "cuts_per_s": round(cuts / dt, 1), "audio_seconds_per_s": round(secs / dt, 1),
            "what": "600 s mini-batches as host float32 arrays -> pack + H2D -> speed perturbation -> fbank collated on the device; PCIe-inclusive, never `value`"}


def self_launch(args) -> None:
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def init_dist(backend: str, dev):
    """(dist module, backend actually used).  RCCL first; if its initialisation (or its first collective) fails, the same ranks fall back
    to gloo -- the group carries three scalars per run, no data, so the measurement does not depend on which one it is.  The fallback
    group gets its OWN TCPStore (rank 0 hosts it one port above MASTER_PORT): under torchrun the env:// rendezvous is a client of the
    launcher's agent store, which a second initialisation cannot reuse safely (stale keys of the failed group)."""
    import datetime

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")  # (BENCH_FORCE_DIST without a launcher: a group of one)
    os.environ.setdefault("WORLD_SIZE", "1")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

    def quiet_stdout(fn):
        """gloo announces its connections on STDOUT from C++ ("[Gloo] Rank 0 is connected to ..."): the contract is ONE JSON line there, so
        file descriptor 1 points at stderr while a gloo group is being set up."""
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            return fn()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def gloo_group(why: str):
        def make():
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1, world, is_master=(rank == 0),
                                  timeout=datetime.timedelta(seconds=120), wait_for_workers=False)
            dist.init_process_group(backend="gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            dist.barrier()  # (the full mesh is connected here at the latest)

        quiet_stdout(make)
        return dist, why

    if backend == "nccl":
        if dev.type != "cuda" or torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            # ranks that share a GPU (a self-test of the N > 1 path on a smaller box): RCCL refuses duplicate devices -- do not even try
            return gloo_group("gloo (fewer GPUs than ranks on this node)")
        try:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1, device=dev)
            dist.all_reduce(t)
            torch.cuda.synchronize(dev)
            return dist, "nccl"
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: RCCL initialisation failed ({e!r}); falling back to gloo", file=sys.stderr, flush=True)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            return gloo_group("gloo (RCCL initialisation failed)")
    if backend == "gloo":
        quiet_stdout(lambda: (dist.init_process_group(backend=backend), dist.barrier()))
    else:
        dist.init_process_group(backend=backend)
    return dist, backend


def group_report(dist, backend_used, world: int, rank: int, local_rank: int, dev, cdev, numa):
    """`config.group`: what the process group consists of, as seen by EVERY rank (gather_json: all_reduce / all_gather of byte tensors)."""
    import torch

    me = {"rank": rank, "local_rank": local_rank, "device": str(dev), "visible_devices": torch.cuda.device_count() if torch.cuda.is_available() else 0,
          "pid": os.getpid(), "numa": numa}
    if dev.type == "cuda":
        try:
            props = torch.cuda.get_device_properties(dev)
            me["device_name"], me["gcn_arch"] = props.name, getattr(props, "gcnArchName", None)
            me["pci_bus_id"] = getattr(props, "pci_bus_id", None)
        except Exception as e:  # noqa: BLE001
            me["device_name"] = repr(e)
    out = {"world_size": world, "backend": None if dist is None else ("rccl" if backend_used == "nccl" else backend_used)}
    if backend_used == "nccl":
        try:
            out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            out["rccl_version"] = repr(e)
    if dist is None:
        out["ranks"] = [me]
        return out
    out["ranks"] = gather_json(me, dist, world, cdev)
    out["ranks_seen"] = len(out["ranks"])
    out["distinct_devices"] = len({(r.get("pci_bus_id"), r["device"]) if r.get("pci_bus_id") is not None else r["device"] for r in out["ranks"]})
    return out


def bind_numa(local_rank: int, mode: str):
    """Bind this rank (and every thread it starts later: packing threads, the save thread, pinned-staging first touch) to the CPUs of
    the NUMA node its GPU hangs off -- lhotse_amd.sharding.bind_to_gpu_numa_node, the helper compute_and_store_features_sharded uses.
    `mode`: "auto" = bind when N > 1 (8 ranks that all run on node 0's cores would share one socket's memory controllers and cross the
    inter-socket link for half the GPUs' H2D copies), "on", "off".  Returns what was done, for `config.numa`."""
    from lhotse_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if mode == "off" or (mode == "auto" and world == 1):
        return {"bound": False, "why": f"--numa {mode}" + (" and N = 1" if mode == "auto" else "")}
    return sharding.bind_to_gpu_numa_node(local_rank)


def timed_region(w, steps: int, warmup: int, dev, dist, cdev, world: int):
    """The contract's timed region for workload `w`: (settle,) `warmup` untimed steps, barrier + synchronize, exactly `steps` steps each
    bracketed by HIP events on the launch stream, barrier + synchronize; elapsed = MAX over ranks."""
    import numpy as np
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()
        _sync(dev)

    # everything the timed region needs is prepared BEFORE the warm-up: a device that idles for a millisecond between the warm-up and the
    # timed steps (host-side set-up, a big memset) drops out of its steady power state, and the first ~5 timed launches then run ~20 % slower
    # -- a quarter of the driver's 20 steps (tools/launch_ramp2.py).  Between the two there is only the contract's barrier.
    evs = _event_pairs(dev, steps)
    if hasattr(w, "settle_device"):
        w.settle_device()
    for _ in range(warmup):
        w.step()
    barrier()
    if not os.environ.get("BENCH_NO_CLEAR"):
        w.clear()  # the parity check reads what the TIMED steps wrote
    barrier()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        w.step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_launch = [a.elapsed_time(b) for a, b in evs]
    launch_ms = float(np.mean(per_launch))
    if os.environ.get("BENCH_SHOW_LAUNCHES"):
        print("[bench] per-launch ms:", " ".join(f"{x:.3f}" for x in per_launch[:40]), file=sys.stderr, flush=True)
    if getattr(w, "host_bound", False):  # the step runs on side streams and host threads: the events on this stream see none of it
        launch_ms = elapsed / steps * 1e3
    rank_launch_ms = [launch_ms]
    units_total, audio_total = float(w.units), float(w.audio_seconds)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lm = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(lm, torch.tensor([launch_ms], dtype=torch.float64, device=cdev))
        rank_launch_ms = [float(x.item()) for x in lm]
        s = torch.tensor([units_total, audio_total], dtype=torch.float64, device=cdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        units_total, audio_total = float(s[0]), float(s[1])
    return {"elapsed": elapsed, "launch_ms": launch_ms, "rank_launch_ms": rank_launch_ms, "units_total": units_total, "audio_total": audio_total}


PARITY_MAX_KEYS = ["rel_l2_max", "max_abs_max", "oracle_f32_vs_f64_rel_l2_max", "oracle_f32_vs_f64_max_abs", "hip_vs_f64_max_abs",
                   "oracle_f32_vs_f64_rms", "hip_vs_f64_rms", "lin_margin_max", "lin_own_max", "lin_floor_max"]
PARITY_ALT_KEYS = ["numpy32_vs_f64_max_abs", "hip_vs_numpy32_max_abs", "numpy32_vs_ref32_max_abs"]


def parity_leg(w, rank: int, dist, cdev, log_mel: bool = True):
    """The in-run parity leg on every rank's TIMED output buffer, worst over all ranks -> (JSON block, verdict)."""
    import torch

    from oracle import parity_bar
    from oracle.kaldi_torch import REF32_NAME

    par = w.parity(rank)
    if dist is not None:
        keys = PARITY_MAX_KEYS + [k for k in PARITY_ALT_KEYS if k in par]
        mx = torch.tensor([par[k] for k in keys] + [-par["frac_within"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        cnt = torch.tensor([float(par["n"]), float(par["lin_bad"]), float(par["n_over_2e-3"]), float(par["n_values"]), float(par["lin_own_over1"]),
                            float(par["lin_floor_over1"])], dtype=torch.float64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        local = par
        par = {k: float(v) for k, v in zip(keys, mx[:-1])}
        par.update(frac_within=-float(mx[-1]), n=int(cnt[0].item()), lin_bad=int(cnt[1].item()), lin_own_over1=int(cnt[4].item()), lin_floor_over1=int(cnt[5].item()))
        par.update({"n_over_2e-3": int(cnt[2].item()), "n_values": int(cnt[3].item()), "over_ref_value_max": local["over_ref_value_max"]})  # (the last one: rank 0's own sample)
    # the ONE parity statement of the repository (oracle/parity_bar.py; the GPU suite enforces the same three clauses on the same inputs)
    v = parity_bar.verdict(par)
    sig = lambda x: float(f"{x:.3e}")  # noqa: E731
    parity = {
        "ref32": REF32_NAME,
        "rel_l2_max": sig(par["rel_l2_max"]),
        "max_abs_max": sig(par["max_abs_max"]),  # max|hip - reference32|
        "hip_vs_f64_max_abs": sig(par["hip_vs_f64_max_abs"]),
        "oracle_f32_vs_f64_max_abs": sig(par["oracle_f32_vs_f64_max_abs"]),  # max|reference32 - f64|: the reference's own floor
        "elementwise_bar": sig(v["elementwise_bar"]),
        "K_measured": round(v["K_measured"], 2),
        "K_allowed": v["K_allowed"],
        "hip_vs_f64_rms": sig(par["hip_vs_f64_rms"]),
        "oracle_f32_vs_f64_rms": sig(par["oracle_f32_vs_f64_rms"]),
        "frac_within_rtol1e-4_atol1e-3": par["frac_within"],
        "values_over_2e-3": {"count": par["n_over_2e-3"], "of": par["n_values"], "largest_reference_value_among_them": par["over_ref_value_max"],
                             "log_mel_floor": -15.942385},  # elements over 2e-3 sit within a few nats of the log(eps) clamp: DESIGN section 2
        # clause 2, against float64 truth: share = |exp(x) - exp(f64)| / (1e-4 exp(f64) + eps), eps = the reference's own mel floor
        "linear_domain_vs_f64": {"hip_worst_share": round(par["lin_own_max"], 4), "reference32_worst_share": round(par["lin_floor_max"], 4),
                                 "bar": round(v["linear_bar_share_of_tolerance"], 4), "hip_values_over_1": par["lin_own_over1"],
                                 "reference32_values_over_1": par["lin_floor_over1"], "of": par["n_values"],
                                 "K_linear_measured": round(v["K_linear_measured"], 3) if log_mel else None, "K_linear_allowed": v["K_linear_allowed"]},
        "statement_version": v["statement_version"],
        # ... and the two float32 pipelines against each other (reported; rounds 4's form of the clause: the reference fails it against float64 itself)
        "linear_domain_outside_rtol1e-4_atol_eps": par["lin_bad"],  # values with |exp(hip) - exp(ref32)| > 1e-4 exp(ref32) + eps
        "linear_domain_worst_share_of_tolerance": round(par["lin_margin_max"], 4),  # max |exp(hip) - exp(ref32)| / (1e-4 exp(ref32) + eps)
        "n": par["n"],
        "oracle_f32_vs_f64_rel_l2_max": sig(par["oracle_f32_vs_f64_rel_l2_max"]),
        "pass_rel_l2": v["pass_rel_l2"],
        "pass_linear": v["pass_linear"] if log_mel else None,
        "pass_elementwise": v["pass_elementwise"],
        "pass": v["pass"],
        "what": f"{PARITY_CUTS} cuts per rank sampled from the timed output buffer vs the oracle, worst over all ranks; max_abs_max = max|hip - ref32|; "
                + parity_bar.STATEMENT,
    }
    if "numpy32_vs_f64_max_abs" in par:
        parity["numpy32_floor_of_rounds_1_to_4"] = {
            "numpy32_vs_f64_max_abs": sig(par["numpy32_vs_f64_max_abs"]), "hip_vs_numpy32_max_abs": sig(par["hip_vs_numpy32_max_abs"]),
            "numpy32_vs_ref32_max_abs": sig(par["numpy32_vs_ref32_max_abs"]), "K_against_numpy32_floor": round(v["K_against_numpy32_floor"], 2),
            "what": "oracle/kaldi_ref.py's float32 mode (numpy's float64 rfft rounded to complex64), the ref32 of rounds 1-4: NOT the reference's "
                    "arithmetic, ~4x closer to float64 than the reference is; side by side for the record, not part of `pass`"}
    # clause (1) -- north_star's tolerance -- stops the run; clauses (2) and (3) are tail statistics of maxima over millions of values:
    # they are reported in `pass` (and enforced, on the samples of all eight ranks of the driver's run, by the GPU suite) rather than
    # allowed to cost the driver its bench line
    assert v["pass_rel_l2"], parity
    return parity, v


def roofline_block(w, launch_ms: float, config_name: str):
    achieved = w.algo_bytes / (launch_ms * 1e-3)
    prof = load_profile_constants(w.kernel, config_name)
    traffic, src = None, None
    if prof.get("stale"):
        src = "profiles/traffic.json is stale: the kernel source changed since the PMC run"
    elif prof.get("hbm_bytes_per_cut") is not None:  # fixed-size cuts: bytes per cut x cuts
        traffic = round(float(prof["hbm_bytes_per_cut"]) * w.units)
    elif prof.get("hbm_bytes_per_algorithmic_byte") is not None:  # ragged workloads: the profiled launch's counter bytes / its algorithmic bytes
        traffic = round(float(prof["hbm_bytes_per_algorithmic_byte"]) * w.algo_bytes)
    if traffic is not None:
        src = "profiles/traffic.json (committed rocprofv3 PMC run of this kernel source, not measured in this run)"
    r = {
        "bound": "hbm",
        "achieved": round(achieved / 1e9, 2),
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK, 4),
        "traffic": traffic,
        "traffic_source": src,
        "launch_ms": round(launch_ms, 4),
        "algorithmic_bytes_per_launch": w.algo_bytes,
    }
    if getattr(w, "algo_parts", None):
        r["algorithmic_bytes_parts"] = w.algo_parts
    for k in ("per_kernel", "infinity_cache_note"):
        if prof.get(k) and traffic is not None:
            r["traffic_" + k] = prof[k]
    e2e = getattr(w, "algo_bytes_end_to_end", None)
    if e2e:
        # the same step priced as ONE pass: every input sample read once, every feature written once -- the intermediate (perturbed
        # waveforms written by the resampler and read back by the feature kernel) is an implementation cost, not algorithmic traffic
        r["frac_end_to_end"] = round(e2e / (launch_ms * 1e-3) / HBM_PEAK, 4)
        r["algorithmic_bytes_end_to_end"] = e2e
        r["frac_note"] = ("`frac` prices the two launches separately (resampler in + out, feature kernel in + out: the unfused intermediate counts); "
                          "`frac_end_to_end` prices the step as one pass (input samples once + features once)")
    return r, prof


def sub_config(name: str, args, dev, rank: int, dist, cdev, world: int):
    """One of the other BASELINE configs inside the default run (after the headline's timed region and parity, so that it cannot perturb
    them): the same contract -- settle, warm-up, barrier, K event-bracketed steps, barrier, MAX over ranks -- and the same parity leg."""
    import copy

    a = copy.copy(args)
    a.cuts, a.total_cuts, a.prefetch, a.streams, a.route, a.input = 0, 0, 1, 3, "pair", "uniform"
    steps = {"mfcc40_libri": 60, "onthefly": 30}[name]
    w = WORKLOADS[name](dev, rank, a)
    try:
        tr = timed_region(w, steps, args.warmup, dev, dist, cdev, world)
        out = {
            "metric": w.metric,
            "value": round(tr["units_total"] * steps / tr["elapsed"], 1),
            "unit": "cuts/s",
            "audio_seconds_per_s": round(tr["audio_total"] * steps / tr["elapsed"], 1),
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(tr["elapsed"] / steps * 1e3, 4),
            "workload": w.workload,
            "kernel": w.kernel,
            "cuts_per_gpu_per_step": w.units,
            "rank_launch_ms": [round(x, 4) for x in tr["rank_launch_ms"]],
        }
        out["roofline"], _ = roofline_block(w, tr["launch_ms"], name)
        if not args.no_parity:
            p, _ = parity_leg(w, rank, dist, cdev, log_mel=(name != "mfcc40_libri"))
            out["parity"] = {k: p[k] for k in ("pass", "pass_rel_l2", "pass_linear", "pass_elementwise", "rel_l2_max", "max_abs_max", "hip_vs_f64_max_abs",
                                               "oracle_f32_vs_f64_max_abs", "K_measured", "K_allowed", "n")}
        return out
    finally:
        w.close()


def sub_plumbing(args, dev, rank: int):
    """`extra.configs.plumbing`: one warm + two timed passes of the product's bulk driver over 64 WAV files x 400 (leg D: the ring loader), the other legs
    and the CPU per-cut baseline as its `legs` (tools/plumbing.py) -- BASELINE configs[0] / SURVEY 8d baseline C."""
    import copy

    a = copy.copy(args)
    a.cuts = 400
    w = Plumbing(dev, rank, a)
    try:
        w.step()
        t0 = time.perf_counter()
        for _ in range(2):
            w.step()
        dt = time.perf_counter() - t0
        out = {"metric": w.metric, "value": round(2 * w.units / dt, 1), "unit": "cuts/s", "steps": 2, "warmup": 1, "ms_per_step": round(dt / 2 * 1e3, 2),
               "workload": w.workload, "cuts_per_step": w.units,
               "last_pass": {k: v for k, v in w.last.items() if k.endswith("_share") or k in ("cuts_per_s", "cuts_per_s_incl_worker_start", "seconds_to_first_batch", "transport",
                                                                                                "ring_slots_page_locked", "batches_uploaded_straight_from_the_ring", "batches",
                                                                                                "cpus_busy_by_thread_name", "container_cpus_busy", "container_cpu_quota", "quota_periods_throttled")},
               "what": "value = whole passes incl. the start of the loader's worker processes (a pass is ~1 s: a corpus-sized run amortises that start); "
                       "last_pass.cuts_per_s = the rate behind the first batch"}
        if not args.no_parity:
            from oracle import parity_bar

            f = w.parity(rank)
            v = parity_bar.verdict(f)
            out["parity"] = {"pass": v["pass"], "rel_l2_max": float(f"{f['rel_l2_max']:.3e}"), "max_abs_max": float(f"{f['max_abs_max']:.3e}"), "n": f["n"],
                             "what": "cuts read back through the manifest + archive reader of the last timed pass vs the oracle on the decoded files"}
        out["legs"] = w.extra(a, full=False)["plumbing"]  # (the full set of legs: bench.py --config plumbing)
        return out
    finally:
        w.close()


def after_idle(w, dev, idle_s: float = 2.0, launches: int = 25):
    """The OTHER regime (VERDICT r4 / ADVICE r4): the package idles for `idle_s`, then `launches` back-to-back launches with no settle --
    the per-launch times of the ramp, and the rate the contract's 5 warm-ups + 20 steps see on their own."""
    import torch

    torch.cuda.synchronize(dev)
    time.sleep(idle_s)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in evs:
        a.record()
        w.step()
        b.record()
    torch.cuda.synchronize(dev)
    ms = [a.elapsed_time(b) for a, b in evs]
    tail = ms[5:]
    return {"first_launches_after_idle_ms": [round(x, 4) for x in ms[:10]],
            "contract_only": {"value": round(w.units / (sum(tail) / len(tail) * 1e-3), 1), "unit": "cuts/s", "launch_ms": round(sum(tail) / len(tail), 4),
                              "what": f"no settle launches: {idle_s:.0f} s of idle, then 5 untimed + {len(tail)} event-timed launches back to back (device time, rank 0) -- what a "
                                      "caller that launches after an idle gap sees; `value` of the line is the sustained rate"}}


def gather_json(local, dist, world: int, cdev):
    """Every rank's JSON-serialisable object on every rank, with the same primitives the rest of the run uses (all_reduce / all_gather of
    plain tensors on `cdev`: no pickling, no object collectives -- nothing new for RCCL to trip over on the driver's 8-GPU run)."""
    import torch

    raw = json.dumps(local).encode("utf-8")
    n = torch.tensor([len(raw)], dtype=torch.int64, device=cdev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    size = int(n.item())
    buf = torch.zeros(size + 8, dtype=torch.uint8, device=cdev)
    buf[:8] = torch.tensor(list(len(raw).to_bytes(8, "little")), dtype=torch.uint8, device=cdev)
    buf[8 : 8 + len(raw)] = torch.tensor(list(raw), dtype=torch.uint8, device=cdev)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for t in parts:
        b = bytes(t.cpu().tolist())
        out.append(json.loads(b[8 : 8 + int.from_bytes(b[:8], "little")].decode("utf-8")))
    return out


def gather_extras(local, dist, world: int, cdev=None):
    """Host-fed legs run on ALL ranks at once (barrier in, barrier out): rank 0 gets every rank's dict and the sums of the numeric leaves."""
    if dist is None or world == 1:
        return local
    if hasattr(dist, "all_gather_object") and cdev is None:  # (tests hand in a stand-in group)
        objs = [None] * world
        dist.all_gather_object(objs, local)
    else:
        objs = gather_json(local, dist, world, cdev)

    def total(path):
        vals = []
        for o in objs:
            for k in path:
                o = o.get(k) if isinstance(o, dict) else None
            if isinstance(o, (int, float)):
                vals.append(float(o))
        return round(sum(vals), 1) if len(vals) == world else None

    agg = {}
    def walk(d, path):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, path + [k])
            elif isinstance(v, (int, float)) and not isinstance(v, bool):
                node = agg
                for q in path:
                    node = node.setdefault(q, {})
                node[k] = total(path + [k])
    walk(objs[0] or {}, [])
    return {"aggregate_over_ranks": agg, "per_rank": objs,
            "what": f"host-fed legs of all {world} ranks running CONCURRENTLY (barrier in, barrier out): aggregate = sum over ranks of each rank's own rate -- "
                    "the curve that can bend with N (PCIe root complexes, host memory bandwidth, NUMA), unlike the device-resident `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default per config: >= 1 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="fbank16k", choices=sorted(WORKLOADS), help="fbank16k = BASELINE configs[1] (default), mfcc40_libri = configs[3], onthefly = configs[4], bulk_save = the offline path end to end (SURVEY 8d iii), plumbing = configs[0] with the GPU in it (WAV files -> decode -> features -> storage + manifest)")
    ap.add_argument("--cuts", type=int, default=0, help="cuts per GPU per step (onthefly: mini-batches per step); default per config")
    ap.add_argument("--prefetch", type=int, default=1, help="onthefly: mini-batches per call (a loader that prefetches K packs them into one arena and gets K dense tensors "
                    "from ONE pair of launches); default 1")
    ap.add_argument("--streams", type=int, default=3, help="onthefly: streams the calls alternate over (default 3)")
    ap.add_argument("--stripes", type=int, default=8, help="bulk_save: files the archive is striped over (one writer thread each; default 8 -- "
                    "one page-cache file takes ONE writer's copy rate, ~4 GB/s on the box of profiles/r05_tmpfs_write_probe.json; the single-file rate is in `extra`)")
    ap.add_argument("--route", default="pair", choices=["pair", "per_factor"], help="onthefly: `pair` = the two-launch mini-batch (default), `per_factor` = round 3's route")
    ap.add_argument("--total-cuts", type=int, default=0, help="fbank16k only: STRONG scaling (BASELINE configs[2]: 100000): this many cuts in total per step, "
                    "sharded round-robin over the ranks (same global corpus for every N); default 0 = weak scaling, --cuts per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (`extra`) altogether: A/B runs")
    ap.add_argument("--no-other-configs", action="store_true", help="fbank16k: do not append BASELINE configs[3] / [4] (`extra.configs`)")
    ap.add_argument("--numa", default="auto", choices=["auto", "on", "off"], help="bind every rank to the CPUs of its GPU's NUMA node before any pinned allocation "
                    "(auto = when N > 1); logged in config.numa")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="host processes of the CPU baseline (default: best of a sweep over cores/8, cores/4, cores/2)")
    ap.add_argument("--input", default="uniform", choices=["uniform", "zeros", "sine"], help="fbank16k: synthetic input (the metric is defined on `uniform`; the others exist to expose power/DVFS effects)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default; falls back to gloo if it cannot be initialised) or gloo (self-test of the N>1 path on one GPU)")
    args = ap.parse_args()
    if not args.steps:
        args.steps = {"fbank16k": 250, "mfcc40_libri": 200, "onthefly": 60, "bulk_save": 5, "plumbing": 3}[args.config]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return

    import numpy as np  # noqa: F401
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    stub = bool(os.environ.get("BENCH_SELFTEST_STUB"))  # the N > 1 plumbing without a device (SelfTestStub): tests only, says so in the line
    all_cpus = sorted(os.sched_getaffinity(0))
    if stub:
        dev = torch.device("cpu")
        numa = {"bound": False, "why": "BENCH_SELFTEST_STUB: no device", "rank": rank}
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
        # one GPU per rank; ranks wrap around when fewer devices are visible (a launcher that narrows *_VISIBLE_DEVICES per rank, or the
        # gloo self-test where all ranks share the one GPU of the box)
        local_rank = local_rank % torch.cuda.device_count()
        numa = bind_numa(local_rank, args.numa)  # before the first pinned allocation and before any worker thread exists
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist, backend_used = None, None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the RCCL path with a single rank (self-test)
        dist, backend_used = init_dist("gloo" if stub else args.dist_backend, dev)
    cdev = dev if (dist is not None and backend_used == "nccl") else torch.device("cpu")  # where collective tensors live
    # who is in the group: every rank's device, visible-device count, NUMA binding and (on RCCL) the library version, gathered with the same
    # plain tensor collectives as everything else -- the driver's first 8-GPU record then PROVES that RCCL saw 8 ranks on 8 devices
    group = group_report(dist, backend_used, world, rank, local_rank, dev, cdev, numa)

    if args.total_cuts and args.config != "fbank16k":
        ap.error("--total-cuts is defined for --config fbank16k")
    w = (SelfTestStub if stub else WORKLOADS[args.config])(dev, rank, args)

    tr = timed_region(w, args.steps, args.warmup, dev, dist, cdev, world)
    elapsed, launch_ms = tr["elapsed"], tr["launch_ms"]

    # ---- parity in the same run, on every rank, on the timed output buffer
    parity = None
    if not args.no_parity:
        parity, _ = parity_leg(w, rank, dist, cdev, log_mel=(args.config != "mfcc40_libri"))

    # ---- the other BASELINE configs under the same (driver) clock: configs[3] and configs[4], every rank, same contract
    other = {}
    if args.config == "fbank16k" and not args.no_other_configs and not args.no_extra and not args.total_cuts and args.input == "uniform" and not stub:
        for name in ("mfcc40_libri", "onthefly"):
            other[name] = sub_config(name, args, dev, rank, dist, cdev, world)

    # ---- PCIe-inclusive legs: all ranks at once
    extra = {}
    if not args.no_extra:
        if dist is not None:
            dist.barrier()
        try:
            local = w.extra(args) or {}
        except Exception as e:  # noqa: BLE001 -- every rank must still take part in the collectives below: an error travels as data
            local = {"error": repr(e)}
        if dist is not None:
            dist.barrier()
        extra = gather_extras(local, dist, world, cdev)  # unconditionally, on every rank
        if args.config == "fbank16k" and hasattr(w, "settle_device") and not stub:
            extra.update(after_idle(w, dev))
    # ---- BASELINE configs[0] with the GPU in it (WAV files -> decode -> features -> storage + manifest): N = 1 only (a host-bound leg per
    # rank would only measure the ranks' competition for the host), after everything else, and never at the cost of the line
    if (args.config == "fbank16k" and world == 1 and not args.no_other_configs and not args.no_extra and not args.total_cuts and args.input == "uniform"
            and not stub and not os.environ.get("BENCH_NO_PLUMBING")):
        try:
            other["plumbing"] = sub_plumbing(args, dev, rank)
        except Exception as e:  # noqa: BLE001
            other["plumbing"] = {"error": repr(e)}
    if other:
        extra["configs"] = other

    if rank == 0:
        value = tr["units_total"] * args.steps / elapsed
        roof, prof = roofline_block(w, launch_ms, args.config)
        res = {
            "metric": w.metric,
            "value": round(value, 1),
            "unit": "cuts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.total_cuts else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": w.workload,
                "name": args.config,
                "cuts_per_gpu_per_step": w.units,
                "audio_seconds_per_s": round(tr["audio_total"] * args.steps / elapsed, 1),
                "sharding": "cuts sharded across ranks, no data-path collective",
                "scaling": (f"strong: {args.total_cuts} cuts in total per step, ceil(total / world) per rank" if args.total_cuts
                            else "weak: the same number of cuts per GPU per step for every N"),
                "kernel": w.kernel,
                "settle": (f"{w.settle} untimed launches directly in front of the contract's warm-up: sustained-throughput metric, the first launches "
                           "after start-up run 1-4 % slower while the package reaches its steady power state (tools/launch_ramp.py); the rate WITHOUT them "
                           "is extra.contract_only, the ramp itself extra.first_launches_after_idle_ms") if getattr(w, "settle", 0) else None,
                "world_size": world,
                "dist_backend": None if dist is None else ("rccl" if backend_used == "nccl" else backend_used),
                "rank_launch_ms": [round(x, 4) for x in tr["rank_launch_ms"]],
                "numa": numa,
                "numa_per_rank": [r.get("numa") for r in group["ranks"]],
                "group": group,
                **({"SELFTEST": "BENCH_SELFTEST_STUB=1: no device work was done; this line is NOT a measurement"} if stub else {}),
            },
            "parity": parity,
            "roofline": roof,
        }
        if getattr(w, "host_bound", False):
            res["roofline"]["note"] = ("host-, PCIe- and file-system-bound configuration: `achieved` is algorithmic feature-kernel bytes per wall second of the "
                                       "whole step, not a kernel rate; the stage split is in extra")
            res["data"] = "synthetic (host-resident waveforms: PCIe-inclusive)"
        ipf = prof.get("valu_instr_per_frame") if args.config == "fbank16k" else None
        if ipf:
            # every wave64 VALU instruction occupies its SIMD's issue port for >= 2 clk (packed f32 ones 3, measured:
            # tools/ubench/valu_rate.hip); a wave instruction covers `frames_per_wave_instr` frames
            clk_per_instr = float(prof.get("valu_clk_per_instr", 2.0))
            frames_per_s = w.units * FRAMES_PER_CUT / (launch_ms * 1e-3)
            mhz = prof.get("sclk_MHz_under_load")
            res["roofline"]["secondary"] = {
                "bound": "valu_f32",
                "instr_per_frame": ipf,
                "clk_per_instr": clk_per_instr,
                "achieved_frac": round(frames_per_s * float(ipf) * clk_per_instr / (NUM_SIMDS * MAX_CLOCK), 4),
                "achieved_frac_at_measured_clock": None if not mhz else round(frames_per_s * float(ipf) * clk_per_instr / (NUM_SIMDS * mhz * 1e6), 4),
                "shader_clock": {"median_MHz": mhz, "source": f"profiles/traffic.json: rocm-smi samples over a 7 s run of this kernel on the box of the committed PMC run ({prof.get('power_probe')})"} if mhz else None,
                "what": "wave-level VALU instructions per frame (committed PMC run: SQ_INSTS_VALU / frames) x issue clocks per instruction "
                        "/ (1024 SIMDs x clock): the share of the chip's VALU issue slots this launch rate needs -- at the nominal 2.4 GHz "
                        "(achieved_frac) and at the shader clock the chip holds under its 1.4 kW power cap while this kernel runs for seconds (`shader_clock`)",
            }
        if extra:
            res["extra"] = extra
        if world == 1 and not args.no_cpu_baseline and not stub:
            os.sched_setaffinity(0, all_cpus)  # the CPU baseline's worker processes get the whole host, whatever --numa did to this rank
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_procs, w.cpu_mode, w.cpu_what)
        print(json.dumps(res), flush=True)
    w.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
