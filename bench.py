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
    sweep = [procs] if procs else sorted({max(1, min(ncpu, n)) for n in (ncpu // 8, ncpu // 4, ncpu // 2)})
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
        "sweep": runs,
        "sample": f"{best['cuts']} cuts in {best['seconds']:.0f} s wall: {best['processes']} single-threaded processes of the reference's torch CPU {what} "
        f"call sequence (oracle/kaldi_torch.py, pinned to the reference's outputs on the goldens; {best['cuts_per_s'] / best['processes']:.0f} cuts/s per process), "
        f"best of a sweep over {[r['processes'] for r in runs]} processes; host has {ncpu} logical cores ({cpu_model()})",
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


def kernel_source_hash(files) -> str:
    h = hashlib.sha256()
    for rel in files:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_profile_constants(kernel_name: str):
    """HBM bytes per cut and VALU instructions per frame from the committed PMC profile -- only if it was measured on THIS kernel source
    (profiles/traffic.json carries the sha256 of the files it names; a changed kernel body invalidates the numbers instead of re-labelling them)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if kernel_name.split(" ")[0] != t.get("kernel"):  # plan.kernel_name = "<kernel> lds=... blocks/CU=..."
            return {}
        if t.get("source_sha256_16") != kernel_source_hash(t.get("source_files", [])):
            return {"stale": True}
        return t
    except Exception:
        return {}


def compare(got, want, truth, log_mel=True):
    """Error figures of one cut: rel_l2 / max_abs vs the float32 oracle, values inside rtol 1e-4 + atol 1e-3, and BOTH float32
    implementations against the float64 oracle (max and mean square), so that an element-wise number can be read in context."""
    import numpy as np

    d = np.abs(got.astype(np.float64) - want)
    fl = np.abs(want.astype(np.float64) - truth)
    own = np.abs(got.astype(np.float64) - truth)
    return {
        "rel": float(np.linalg.norm(got - want) / np.linalg.norm(want)),
        "abs": float(d.max()),
        "within": int((d <= 1e-3 + 1e-4 * np.abs(want)).sum()),
        "total": int(d.size),
        "floor_rel": float(np.linalg.norm(want - truth) / np.linalg.norm(truth)),
        "floor_abs": float(fl.max()),
        "own_abs": float(own.max()),
        "floor_sq": float((fl ** 2).sum()),
        "own_sq": float((own ** 2).sum()),
        # the elements behind a max_abs above the suite's bar: how many, and how far above log(mel floor) the largest of them sits
        # the same comparison in the LINEAR domain with the reference's own floor constant as absolute tolerance: the reference clamps
        # every mel energy at eps = 1.19e-7 (layers.py:536-538, 577), i.e. treats differences below eps as nothing
        "lin_bad": (int((np.abs(np.exp(got.astype(np.float64)) - np.exp(want.astype(np.float64))) > 1e-4 * np.exp(want.astype(np.float64)) + 1.1920929e-07).sum())
                    if log_mel else 0),
        "over": int((d > 2e-3).sum()),
        "over_ref_max": float(want[d > 2e-3].max()) if bool((d > 2e-3).any()) else None,
    }


def fold(stats):
    n = max(1, sum(s["total"] for s in stats))
    return {
        "rel_l2_max": max(s["rel"] for s in stats),
        "max_abs_max": max(s["abs"] for s in stats),
        "frac_within": sum(s["within"] for s in stats) / n,
        "n": len(stats),
        "oracle_f32_vs_f64_rel_l2_max": max(s["floor_rel"] for s in stats),
        "oracle_f32_vs_f64_max_abs": max(s["floor_abs"] for s in stats),
        "hip_vs_f64_max_abs": max(s["own_abs"] for s in stats),
        "oracle_f32_vs_f64_rms": (sum(s["floor_sq"] for s in stats) / n) ** 0.5,
        "hip_vs_f64_rms": (sum(s["own_sq"] for s in stats) / n) ** 0.5,
        "lin_bad": sum(max(s["lin_bad"], 0) for s in stats),
        "n_over_2e-3": sum(s["over"] for s in stats),
        "over_ref_value_max": max([s["over_ref_max"] for s in stats if s["over_ref_max"] is not None], default=None),
        "n_values": n,
    }


# ---------------------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------------------
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

        self.torch, self.np = torch, np
        C = self.C = args.cuts or self.default_cuts
        self.ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        L = self.L = self.plan.lib
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.wave = torch.empty((C, SAMPLES_PER_CUT), dtype=torch.float32, device=dev)
        for i in range(0, C, 500):
            self.wave[i : i + 500].uniform_(-0.5, 0.5, generator=g)
        if args.input == "zeros":
            self.wave.zero_()
        elif args.input == "sine":
            t = torch.arange(SAMPLES_PER_CUT, device=dev, dtype=torch.float32)
            self.wave[:] = 0.4 * torch.sin(2 * 3.14159265 * 440.0 / 16000.0 * t)
        self.out = torch.empty((C * FRAMES_PER_CUT, NUM_MELS), dtype=torch.float32, device=dev)
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
        self.kernel = self.plan.kernel_name
        self.workload = (f"BASELINE configs[1]: {C} x 10 s 16 kHz mono cuts per GPU per step, 80-dim log-mel Fbank (25/10 ms, povey, no dither), "
                         "device-resident float32 in / float32 out")

    def step(self):
        self.L.check("hipfeat_extract_layout", self.plan.handle, self.layout, self.wave.data_ptr(), self.out.data_ptr(), self.stream)

    def clear(self):
        self.out.zero_()

    def parity(self, rank):
        from oracle.kaldi_ref import RefConfig, RefExtractor

        np = self.np
        chk = self.out[:FRAMES_PER_CUT].float()
        assert self.torch.isfinite(chk).all() and float(chk.std()) > 0.1
        rs = np.random.RandomState(4321 + rank)
        idx = np.sort(rs.choice(self.C, size=min(PARITY_CUTS, self.C), replace=False))
        o32, o64 = RefExtractor(RefConfig(kind="fbank"), np.float32), RefExtractor(RefConfig(kind="fbank"), np.float64)
        stats = []
        for i in idx:
            x = self.wave[int(i)].cpu().numpy()
            got = self.out[int(i) * FRAMES_PER_CUT : (int(i) + 1) * FRAMES_PER_CUT].cpu().numpy()
            want, truth = o32.extract(x), o64.extract(x)
            assert got.shape == want.shape, (got.shape, want.shape)
            stats.append(compare(got, want, truth))
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
        self.kernel = self.plan.kernel_name
        self.workload = (f"BASELINE configs[3] stand-in: {C} cuts per GPU per step with LibriSpeech-like lengths (log-normal, 1-35 s, mean "
                         f"{self.audio_seconds / C:.1f} s; the corpus is not available offline), 40-dim MFCC (40 mel filters, 40 cepstra, lifter 22), "
                         "packed ragged batch, device-resident float32 in / float32 out")

    def step(self):
        self.L.check("hipfeat_extract_layout", self.plan.handle, self.layout, self.wave.data_ptr(), self.out.data_ptr(), self.stream)

    def clear(self):
        self.out.zero_()

    def parity(self, rank):
        from oracle.kaldi_ref import RefConfig, RefExtractor

        np = self.np
        rs = np.random.RandomState(4321 + rank)
        idx = np.sort(rs.choice(self.C, size=min(PARITY_CUTS, self.C), replace=False))
        rc = RefConfig(kind="mfcc", num_filters=40, num_ceps=40, cepstral_lifter=22)
        o32, o64 = RefExtractor(rc, np.float32), RefExtractor(rc, np.float64)
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

        self.torch, self.np, self.A, self.dev = torch, np, A, dev
        NB = self.NB = args.cuts or self.default_cuts
        self.ex = lhotse_amd.HipFbank(lhotse_amd.HipFbankConfig(device=f"cuda:{dev.index}"))
        self.plan = self.ex.plan
        rng = np.random.RandomState(rank)  # rank 0 = seed 0
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        self.batches = []
        ncuts = 0
        for b in range(NB):
            lens, tot = [], 0.0
            while True:
                d = rng.uniform(1.0, 30.0)
                if tot + d > 600.0:
                    break
                lens.append(int(d * SR))
                tot += d
            lens = np.asarray(lens, dtype=np.int64)
            fac = rng.choice([0.9, 1.0, 1.1], size=len(lens))
            offs = np.concatenate([[0], np.cumsum((lens + 3) & ~3)[:-1]]).astype(np.int64)
            front = int(offs[-1] + lens[-1])
            arena = torch.empty(((front + 3) & ~3) + A.perturbed_tail_floats(lens, fac, SR), dtype=torch.float32, device=dev)
            arena[:front].uniform_(-0.5, 0.5, generator=g)
            self.batches.append({"arena": arena, "offs": offs, "lens": lens, "fac": fac, "front": front})
            ncuts += len(lens)
        self.units = ncuts
        self.feats = [None] * NB
        self.step()  # sizes of the perturbed batch (for the byte count) and the first outputs
        in_samples = out_samples = frames = audio_in = all_out = 0
        for bt, (f, fl, po, pl) in zip(self.batches, self.feats):
            pert = bt["fac"] != 1.0
            in_samples += int(bt["lens"][pert].sum())
            out_samples += int(pl[pert].sum())
            frames += int(fl.sum())
            audio_in += int(bt["lens"].sum())
            all_out += int(pl.sum())
        self.audio_seconds = audio_in / SR
        # resampler: reads the perturbed cuts' inputs, writes their outputs; fbank: reads every (perturbed) cut once, writes its rows once
        self.algo_bytes = 4 * (in_samples + out_samples) + 4 * all_out + 4 * NUM_MELS * frames
        self.kernel = self.plan.kernel_name + " + resample_fast_kernel"
        self.workload = (f"BASELINE configs[4]: {NB} mini-batches of 600 s per GPU per step ({ncuts} cuts U(1,30) s, {self.audio_seconds:.0f} s of audio), "
                         "each cut speed-perturbed by 0.9 / 1.0 / 1.1 on the device, then 80-dim log-mel Fbank written straight into the padded "
                         "(B, Tmax, 80) batch tensor (LOG_EPSILON padding); waveforms resident in HBM, features stay on the device")

    def step(self):
        A = self.A
        for k, bt in enumerate(self.batches):
            po, pl = A.perturb_speed_in_arena(bt["arena"], bt["offs"], bt["lens"], bt["fac"], SR, bt["front"])
            f, fl = self.plan.run_collated(bt["arena"], po, pl, None, LOG_EPSILON)
            self.feats[k] = (f, fl, po, pl)

    def clear(self):
        for k in range(self.NB):
            if self.feats[k] is not None:
                self.feats[k][0].zero_()

    def parity(self, rank):
        from oracle import resample_ref as R
        from oracle.kaldi_ref import RefConfig, RefExtractor

        np = self.np
        rs = np.random.RandomState(4321 + rank)
        o32, o64 = RefExtractor(RefConfig(kind="fbank"), np.float32), RefExtractor(RefConfig(kind="fbank"), np.float64)
        stats = []
        for _ in range(PARITY_CUTS):
            b = int(rs.randint(self.NB))
            bt = self.batches[b]
            f, fl, po, pl = self.feats[b]
            i = int(rs.randint(len(bt["lens"])))
            x = bt["arena"][int(bt["offs"][i]) : int(bt["offs"][i]) + int(bt["lens"][i])].cpu().numpy()
            fac = float(bt["fac"][i])
            y32 = R.speed(x, SR, fac, np.float32) if fac != 1.0 else x
            y64 = R.speed(x.astype(np.float64), SR, fac, np.float64) if fac != 1.0 else x.astype(np.float64)
            assert len(y32) == int(pl[i]), (len(y32), int(pl[i]))
            want, truth = o32.extract(y32), o64.extract(y64)
            got = f[i, : int(fl[i])].cpu().numpy()
            assert got.shape == want.shape, (got.shape, want.shape)
            assert bool((f[i, int(fl[i]) :] == LOG_EPSILON).all()), "padding rows of the collated batch"
            stats.append(compare(got, want, truth))
        return fold(stats)

    def extra(self, args):
        if args.no_host_fed:
            return {}
        return {"host_fed": onthefly_host_fed(self)}

    def close(self):
        pass


WORKLOADS = {w.name: w for w in (Fbank16k, Mfcc40Libri, OnTheFly)}


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
        po, pl = A.perturb_speed_in_arena(arena, offs, lens, bt["fac"], SR, packed.numel())
        return w.plan.run_collated(arena, po, pl, None, LOG_EPSILON)

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
    return {"batches_per_s": round(n / dt, 1), "cuts_per_s": round(cuts / dt, 1), "audio_seconds_per_s": round(secs / dt, 1),
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

    def gloo_group(why: str):
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1, world, is_master=(rank == 0),
                              timeout=datetime.timedelta(seconds=120), wait_for_workers=False)
        dist.init_process_group(backend="gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
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
    dist.init_process_group(backend=backend)
    return dist, backend


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default per config: >= 1 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="fbank16k", choices=sorted(WORKLOADS), help="fbank16k = BASELINE configs[1] (default), mfcc40_libri = configs[3], onthefly = configs[4]")
    ap.add_argument("--cuts", type=int, default=0, help="cuts per GPU per step (onthefly: mini-batches per step); default per config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="host processes of the CPU baseline (default: best of a sweep over cores/8, cores/4, cores/2)")
    ap.add_argument("--input", default="uniform", choices=["uniform", "zeros", "sine"], help="fbank16k: synthetic input (the metric is defined on `uniform`; the others exist to expose power/DVFS effects)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default; falls back to gloo if it cannot be initialised) or gloo (self-test of the N>1 path on one GPU)")
    args = ap.parse_args()
    if not args.steps:
        args.steps = {"fbank16k": 250, "mfcc40_libri": 200, "onthefly": 60}[args.config]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    # one GPU per rank; ranks wrap around when fewer devices are visible (a launcher that narrows *_VISIBLE_DEVICES per rank, or the
    # gloo self-test where all ranks share the one GPU of the box)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, backend_used = None, None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the RCCL path with a single rank (self-test)
        dist, backend_used = init_dist(args.dist_backend, dev)
    cdev = dev if (dist is not None and backend_used == "nccl") else torch.device("cpu")  # where collective tensors live

    w = WORKLOADS[args.config](dev, rank, args)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        w.step()
    barrier()
    w.clear()  # the parity check below reads what the TIMED steps wrote
    barrier()
    # per-step device time: HIP events on the launch stream (torch's current stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        w.step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
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

    # ---- parity in the same run, on every rank, on the timed output buffer
    parity = None
    if not args.no_parity:
        par = w.parity(rank)
        if dist is not None:
            keys = ["rel_l2_max", "max_abs_max", "oracle_f32_vs_f64_rel_l2_max", "oracle_f32_vs_f64_max_abs", "hip_vs_f64_max_abs",
                    "oracle_f32_vs_f64_rms", "hip_vs_f64_rms"]
            mx = torch.tensor([par[k] for k in keys] + [-par["frac_within"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            cnt = torch.tensor([float(par["n"])], dtype=torch.float64, device=cdev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            local = par
            par = {k: float(v) for k, v in zip(keys, mx[:-1])}
            par.update(frac_within=-float(mx[-1]), n=int(cnt.item()))
            par.update({k: local[k] for k in ("n_over_2e-3", "over_ref_value_max", "n_values", "lin_bad")})  # (rank 0's own sample)
        # element-wise bar: the suite's 2e-3 (log units), or 3 x the reference arithmetic's OWN float32 error against float64 on the same
        # cuts where that is larger -- measured the same way for both (each against the float64 oracle), DESIGN section 2
        abs_bar = max(2e-3, 3.0 * par["oracle_f32_vs_f64_max_abs"])
        ok_rel = bool(par["rel_l2_max"] <= 1e-4)
        ok_abs = bool(par["hip_vs_f64_max_abs"] <= abs_bar)
        parity = {
            "rel_l2_max": float(f"{par['rel_l2_max']:.3e}"),
            "max_abs_max": float(f"{par['max_abs_max']:.3e}"),
            "hip_vs_f64_max_abs": float(f"{par['hip_vs_f64_max_abs']:.3e}"),
            "oracle_f32_vs_f64_max_abs": float(f"{par['oracle_f32_vs_f64_max_abs']:.3e}"),
            "max_abs_bar": float(f"{abs_bar:.3e}"),
            "hip_vs_f64_rms": float(f"{par['hip_vs_f64_rms']:.3e}"),
            "oracle_f32_vs_f64_rms": float(f"{par['oracle_f32_vs_f64_rms']:.3e}"),
            "frac_within_rtol1e-4_atol1e-3": par["frac_within"],
            "values_over_2e-3": {"count": par["n_over_2e-3"], "of": par["n_values"], "largest_reference_value_among_them": par["over_ref_value_max"],
                                 "log_mel_floor": -15.942385},  # elements over the bar sit within a few nats of the log(eps) clamp: DESIGN section 2
            "linear_domain_outside_rtol1e-4_atol_eps": par["lin_bad"],  # values with |exp(hip) - exp(ref32)| > 1e-4 exp(ref32) + eps (the reference's own mel floor)
            "n": par["n"],
            "oracle_f32_vs_f64_rel_l2_max": float(f"{par['oracle_f32_vs_f64_rel_l2_max']:.3e}"),
            "pass_rel_l2": ok_rel,
            "pass_max_abs": ok_abs,
            "pass": bool(ok_rel and ok_abs),
            "what": f"{PARITY_CUTS} cuts per rank sampled from the timed output buffer vs the oracle (float32 = the reference's arithmetic, float64 = truth); worst over "
                    "all ranks; pass = rel_l2(hip, ref32) <= 1e-4 and max|hip - f64| <= max(2e-3, 3 x max|ref32 - f64|); max_abs_max = max|hip - ref32|",
        }
        assert ok_rel, parity  # the north star's tolerance; the element-wise verdict is reported, not asserted (it is a tail statistic)

    if rank == 0:
        value = units_total * args.steps / elapsed
        achieved = w.algo_bytes / (launch_ms * 1e-3)
        prof = load_profile_constants(w.kernel) if args.config == "fbank16k" else {}
        bytes_per_cut = prof.get("hbm_bytes_per_cut")
        res = {
            "metric": w.metric,
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
                "workload": w.workload,
                "name": args.config,
                "cuts_per_gpu_per_step": w.units,
                "audio_seconds_per_s": round(audio_total * args.steps / elapsed, 1),
                "sharding": "cuts sharded across ranks, no data-path collective",
                "kernel": w.kernel,
                "world_size": world,
                "dist_backend": None if dist is None else ("rccl" if backend_used == "nccl" else backend_used),
                "rank_launch_ms": [round(x, 4) for x in rank_launch_ms],
            },
            "parity": parity,
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved / 1e9, 2),
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 4),
                "traffic": None if bytes_per_cut is None else round(float(bytes_per_cut) * w.units),
                "traffic_source": ("profiles/traffic.json is stale: the kernel source changed since the PMC run" if prof.get("stale") else None) if bytes_per_cut is None
                else "profiles/traffic.json (committed rocprofv3 PMC run of this kernel source, not measured in this run)",
                "launch_ms": round(launch_ms, 4),
                "algorithmic_bytes_per_launch": w.algo_bytes,
            },
        }
        ipf = prof.get("valu_instr_per_frame")
        if ipf:
            # every wave64 VALU instruction occupies its SIMD's issue port for >= 2 clk (packed f32 ones 3, measured:
            # tools/ubench/valu_rate.hip); a wave instruction covers `frames_per_wave_instr` frames
            clk_per_instr = float(prof.get("valu_clk_per_instr", 2.0))
            frames_per_s = w.units * FRAMES_PER_CUT / (launch_ms * 1e-3)
            res["roofline"]["secondary"] = {
                "bound": "valu_f32",
                "instr_per_frame": ipf,
                "clk_per_instr": clk_per_instr,
                "achieved_frac": round(frames_per_s * float(ipf) * clk_per_instr / (NUM_SIMDS * MAX_CLOCK), 4),
                "what": "wave-level VALU instructions per frame (committed PMC run: SQ_INSTS_VALU / frames) x issue clocks per instruction "
                        "/ (1024 SIMDs x 2.4 GHz): the share of the chip's VALU issue slots this launch rate needs",
            }
        if world == 1:
            extra = w.extra(args)
            if extra:
                res["extra"] = extra
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_procs, w.cpu_mode, w.cpu_what)
        print(json.dumps(res), flush=True)
    w.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
