"""
BASELINE configs[0] / SURVEY 8d "Config 1" + baseline C: the PLUMBING around the extractor -- int16 WAV files on tmpfs -> decode ->
features -> one .npy per cut (or the product's archive) + a gzip JSONL manifest -- as cuts/s, with the stage split.

Why this module restates lhotse's driver loops instead of calling them: lhotse is not installed on the GPU box and a Python reference
cannot travel there (task rules).  Each leg keeps the STRUCTURE of the driver it stands for (processes, threads, argument forms, what is
written, when it is flushed) and cites it; tools/plumbing_reference.py runs the REAL drivers next to legs A1 / A2 in the authoring container
(8 vCPU, no GPU) so that the restated loops can be read against the real ones on the same machine (profiles/r06_plumbing_container.json).

  A  cpu_per_cut(num_jobs)    CutSet.compute_and_store_features(Fbank(), NumpyFilesWriter, num_jobs)         lhotse/cut/set.py:1981-2195
                              N processes x torch.set_num_threads(1) (bin/modes/features.py:25-32): decode -> Fbank.extract -> np.save
                              -> manifest line.  The extractor is the reference's own torch call sequence (oracle/kaldi_torch.py, bit-equal
                              to the live reference): `cpu_baseline.kind == "port"`.  CHECKER / BASELINE ONLY (the one leg that touches oracle/).
  B  hip_batch_numpy_files    CutSet.compute_and_store_features_batch(HipFbank(), NumpyFilesWriter, num_workers=W)   lhotse/cut/set.py:2197-2408
                              DataLoader worker processes decode -> extract_batch(list of (1, T)) on the main thread under no_grad ->
                              ONE save thread: per cut .npy + json.dumps(manifest dict) into a gzip JSONL, flushed per cut.
  C  hip_bulk                 lhotse_amd.compute_and_store_features_batch's native route (storage.py): workers decode AND serialise the
                              line halves -> hipfeat_host_pipeline (pack / H2D / kernel / D2H) -> archive thread (hipfeat_archive_append,
                              striped) -> manifest thread (hipfeat_manifest_lines + gzip + flush per batch).

The corpus: 64 x 10 s 16 kHz mono int16 WAVs of seeded uniform noise (the content does not matter to any stage); a pass visits every file
`repeat` times (cut ids differ, as several cuts of one recording would) so that a timed leg lasts about a second.
"""
from __future__ import annotations

import dataclasses
import gzip
import json
import os
import sys
import time
import wave
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional

import numpy as np

SR = 16000
SAMPLES = 160000
FRAMES = 1000
NUM_MELS = 80
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------------------------------------------------------------
# corpus
# ----------------------------------------------------------------------------------------------------------------------------------
def write_corpus(directory: str, n_files: int = 64, seed: int = 0) -> List[str]:
    os.makedirs(directory, exist_ok=True)
    rs = np.random.RandomState(seed)
    paths = []
    for i in range(n_files):
        pcm = (rs.uniform(-0.5, 0.5, size=SAMPLES) * 32767.0).astype(np.int16)
        p = os.path.join(directory, f"rec{i:03d}.wav")
        with wave.open(p, "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(SR)
            f.writeframes(pcm.tobytes())
        paths.append(p)
    return paths


def read_wav(path: str, pcm16: bool = False) -> np.ndarray:
    """(1, T): float32 = int16 / 32768 (what lhotse's audio backends return for a PCM_16 file), or the int16 samples themselves."""
    with wave.open(path, "rb") as f:
        raw = f.readframes(f.getnframes())
    x = np.frombuffer(raw, dtype=np.int16)
    return x.reshape(1, -1).copy() if pcm16 else (x.astype(np.float32) / 32768.0).reshape(1, -1)


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


class Cut:  # what lhotse_amd.storage._mono_cut_dict reads off a MonoCut
    __slots__ = ("id", "start", "duration", "channel", "recording_id", "supervisions", "custom", "recording", "sampling_rate", "path")


def make_cuts(paths: List[str], repeat: int) -> List[Cut]:
    cuts = []
    for r in range(repeat):
        for i, p in enumerate(paths):
            c = Cut()
            n = r * len(paths) + i
            c.id, c.start, c.duration, c.channel, c.recording_id, c.sampling_rate, c.path = f"cut-{n:07d}", 0.0, SAMPLES / SR, 0, f"rec{i:03d}", SR, p
            c.supervisions = [Sup(id=c.id, recording_id=c.recording_id, start=0.0, duration=c.duration, text="SYNTHETIC UTTERANCE " * 4, language="English",
                                  speaker=f"spk{n % 251}")]
            c.custom = None
            c.recording = Rec(id=c.recording_id, sources=[Src("file", [0], p)], sampling_rate=SR, num_samples=SAMPLES, duration=c.duration, channel_ids=[0])
            cuts.append(c)
    return cuts


def batches_of(cuts: List[Cut], batch_cuts: int = 60) -> List[List[int]]:
    """600 s batches (lhotse's default batch_duration): 60 x 10 s."""
    return [list(range(i, min(i + batch_cuts, len(cuts)))) for i in range(0, len(cuts), batch_cuts)]


def cut_manifest_dict(c: Cut, feat: Dict) -> Dict:
    """MonoCut.to_dict() with features attached (lhotse/cut/mono.py, features/base.py:556-600), None fields dropped as asdict_nonull does."""
    nn = lambda d: {k: v for k, v in d.items() if v is not None}  # noqa: E731
    return {"id": c.id, "start": c.start, "duration": c.duration, "channel": c.channel,
            "supervisions": [nn(dataclasses.asdict(s)) for s in c.supervisions], "features": feat,
            "recording": nn({**dataclasses.asdict(c.recording), "sources": [dataclasses.asdict(s) for s in c.recording.sources]}), "type": "MonoCut"}


def features_dict(c: Cut, type_name: str, num_frames: int, storage_type: str, storage_path: str, storage_key: str) -> Dict:
    return {"type": type_name, "num_frames": num_frames, "num_features": NUM_MELS, "frame_shift": 0.01, "sampling_rate": SR, "start": c.start,
            "duration": c.duration, "storage_type": storage_type, "storage_path": storage_path, "storage_key": storage_key,
            "recording_id": c.recording_id, "channels": c.channel}


def numpy_files_write(storage_dir: str, key: str, value: np.ndarray) -> str:
    """NumpyFilesWriter.write (lhotse/features/io.py:499-525)."""
    sub = os.path.join(storage_dir, key[:3])
    os.makedirs(sub, exist_ok=True)
    np.save(os.path.join(sub, key + ".npy"), value, allow_pickle=False)
    return os.path.join(key[:3], key + ".npy")


# ----------------------------------------------------------------------------------------------------------------------------------
# leg A: the CPU per-cut driver (baseline C of SURVEY 8d), restated; oracle/ is touched HERE ONLY
# ----------------------------------------------------------------------------------------------------------------------------------
_JOB_EXTRACTOR: Dict = {}


def _cpu_job(job: int, cuts: List[Cut], out_dir: str, extractor: str = "cpu") -> int:
    """One job of the per-cut driver (lhotse/cut/set.py:2141-2195, lhotse/features/base.py:160-282): per cut load_audio -> extract ->
    writer.write -> manifest line.  extractor "cpu" = the reference's torch call sequence (oracle: leg A, the baseline), "hip" = the
    product's HipFbank as a drop-in of it (leg E: every job process builds its own plan the first time it extracts)."""
    import torch

    torch.set_num_threads(1)  # lhotse/bin/modes/features.py:25-32
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    ex = _JOB_EXTRACTOR.get(extractor)
    if ex is None:
        if extractor == "hip":
            import lhotse_amd

            hip = lhotse_amd.HipFbank()
            name, fn = hip.name, (lambda x: hip.extract(x, SR))
        else:
            from oracle.kaldi_torch import TorchFbank

            name, fn = "kaldi-fbank", TorchFbank().extract
        ex = _JOB_EXTRACTOR[extractor] = (name, fn)
    name, fn = ex
    store = os.path.join(out_dir, f"feats-{job}")
    with gzip.open(os.path.join(out_dir, f"cuts-{job}.jsonl.gz"), "wt") as man:
        for c in cuts:
            feats = fn(read_wav(c.path)[0])
            key = numpy_files_write(store, c.id, feats)
            assert feats.shape == (FRAMES, NUM_MELS)
            man.write(json.dumps(cut_manifest_dict(c, features_dict(c, name, feats.shape[0], "numpy_files", store, key))) + "\n")
    return len(cuts)


def cpu_per_cut(cuts: List[Cut], out_dir: str, num_jobs: int, extractor: str = "cpu") -> Dict:
    """Leg A (extractor "cpu") / leg E ("hip": the same per-cut driver with HipFbank in place of Fbank -- what a lhotse user gets who changes
    nothing but the extractor object; call it from a process that has not touched the GPU, the jobs are forked).  Job i takes cuts i, i + N, ... (LazySlicer, cut/set.py:2158-2160); worker processes are forked HERE (the reference spawns:
    process start-up is outside the timed loop of a corpus-sized run either way, so it is excluded: workers are started, warmed with one
    cut, then released together)."""
    import multiprocessing as mp

    os.makedirs(out_dir, exist_ok=True)
    ctx = mp.get_context("fork")
    go, ready = ctx.Event(), ctx.Queue()
    res = ctx.Queue()

    def body(j):
        try:
            _cpu_job(j, cuts[:1], os.path.join(out_dir, f"warm{j}"), extractor)  # imports, first-call costs (leg E: the job's plan)
            ready.put(j)
            go.wait()
            t0 = time.perf_counter()
            n = _cpu_job(j, cuts[j::num_jobs], out_dir, extractor)
            res.put((n, time.perf_counter() - t0))
        except BaseException as e:  # noqa: BLE001
            ready.put(-1)
            res.put((0, repr(e)))

    ps = [ctx.Process(target=body, args=(j,)) for j in range(num_jobs)]
    for p in ps:
        os.makedirs(os.path.join(out_dir, f"warm{ps.index(p)}"), exist_ok=True)
        p.start()
    for _ in ps:
        ready.get(timeout=300)
    t0 = time.perf_counter()
    go.set()
    outs = [res.get(timeout=1200) for _ in ps]
    wall = time.perf_counter() - t0
    for p in ps:
        p.join(timeout=30)
    n = sum(o[0] for o in outs)
    return {"cuts_per_s": round(n / wall, 1), "cuts": n, "seconds": round(wall, 3), "num_jobs": num_jobs, "extractor": extractor,
            "per_process_cuts_per_s": round(n / wall / num_jobs, 1), "errors": [o[1] for o in outs if isinstance(o[1], str)] or None}


def _cgroup_cpu() -> Optional[Dict]:
    """usage / user / system microseconds and throttled periods of this container's cgroup (v2), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            d = dict(ln.split() for ln in f)
        return {k: int(d[k]) for k in ("usage_usec", "user_usec", "system_usec", "nr_periods", "nr_throttled", "throttled_usec") if k in d}
    except OSError:
        return None


def cpu_quota() -> Optional[float]:
    """CPUs this container may use per period (cgroup v2 cpu.max), None = unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        return None


def _thread_cpu(pids=()) -> Dict[str, float]:
    """CPU seconds (user + system) by thread name: this process's threads (libhipfeat names its own: hipfeat-pipe / -pack / -stripe; the
    interpreter's threads and the HIP runtime's all read "python...") and, as "loader workers", the processes `pids`."""
    tick = os.sysconf("SC_CLK_TCK")
    out: Dict[str, float] = {}

    def add(name, path):
        try:
            with open(path) as f:
                st = f.read()
            rest = st[st.rindex(")") + 2 :].split()
            out[name] = out.get(name, 0.0) + (int(rest[11]) + int(rest[12])) / tick
        except (OSError, ValueError, IndexError):
            pass

    try:
        for t in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{t}/comm") as f:
                    name = f.read().strip()
            except OSError:
                continue
            add("main thread" if int(t) == os.getpid() else name, f"/proc/self/task/{t}/stat")
    except OSError:
        pass
    for pid in pids:
        add("loader workers", f"/proc/{pid}/stat")
    return out


class _Accounting:
    """CPU of the whole container, of this process's threads by name and of the loader's workers, and the library's pipeline thread's own
    clock, over the steady region of a leg (from the first batch to the end)."""

    def __init__(self, ex, pids=()):
        self.ex, self.c0, self.p0, self.t0, self.pids, self.th0 = ex, None, None, None, list(pids), None

    def _pipe(self):
        try:
            return self.ex._native_pipe().stats() if getattr(getattr(self.ex, "plan", None), "handle", None) else None
        except Exception:  # noqa: BLE001
            return None

    def first_batch(self):
        if self.t0 is None:
            self.t0, self.c0, self.p0, self.th0 = time.perf_counter(), _cgroup_cpu(), self._pipe(), _thread_cpu(self.pids)

    def stop(self):
        """End of the steady region -- BEFORE the loader's workers and the archive's writer threads are torn down."""
        if self.t0 is not None:
            self.end = (time.perf_counter(), _cgroup_cpu(), self._pipe(), _thread_cpu(self.pids))

    def result(self) -> Dict:
        out: Dict = {}
        if self.t0 is None:
            return out
        if getattr(self, "end", None) is None:
            self.stop()
        t1, c1, p1, th1 = self.end
        dt = max(t1 - self.t0, 1e-9)
        out["cpus_busy_by_thread_name"] = {k: round((v - (self.th0 or {}).get(k, 0.0)) / dt, 2) for k, v in sorted(th1.items()) if v - (self.th0 or {}).get(k, 0.0) > 0.005 * dt}
        if self.c0 and c1:
            out["container_cpus_busy"] = round((c1["usage_usec"] - self.c0["usage_usec"]) * 1e-6 / dt, 2)
            out["container_cpus_busy_user_system"] = [round((c1[k] - self.c0[k]) * 1e-6 / dt, 2) for k in ("user_usec", "system_usec")]
            out["container_cpu_quota"] = cpu_quota()
            out["quota_periods_throttled"] = [c1["nr_throttled"] - self.c0["nr_throttled"], c1["nr_periods"] - self.c0["nr_periods"]]
        if p1:
            q0 = self.p0 or {"busy_s": 0.0, "pack_s": 0.0, "device_backpressure_s": 0.0}
            out["pipeline_thread_busy_share"] = round((p1["busy_s"] - q0["busy_s"]) / dt, 3)
            out["pipeline_thread_packing_share"] = round((p1["pack_s"] - q0["pack_s"]) / dt, 3)
            out["pipeline_thread_waiting_for_pcie_or_device_share"] = round((p1["device_backpressure_s"] - q0["device_backpressure_s"]) / dt, 3)
        return out



# ----------------------------------------------------------------------------------------------------------------------------------
# loader side of legs B / C (DataLoader worker processes)
# ----------------------------------------------------------------------------------------------------------------------------------
class DecodeDataset:
    """UnsupervisedWaveformDataset(collate=False) (lhotse/dataset/unsupervised.py:46-82) over WAV paths; with `template` it also
    serialises the two halves of every cut's manifest line, as lhotse_amd.storage's FragmentingWaveformDataset does in the workers."""

    def __init__(self, cuts: List[Cut], pcm16: bool = False, template: Optional[Dict] = None, frame_shift: float = 0.01, packed: bool = False):
        self.cuts, self.pcm16, self.template, self.frame_shift, self.packed = cuts, pcm16, template, frame_shift, packed
        self._rc = {}

    def __getstate__(self):
        """Workers started by spawn / a fork server receive the dataset by pickle, once per worker: the cut objects (thousands, with their
        supervisions) are rebuilt there from (paths, repeat) instead of travelling -- lhotse's dataset objects are just as light (the cuts
        stay with the sampler in the main process)."""
        st = dict(self.__dict__)
        paths = []
        for c in self.cuts:
            if c.path in paths:
                break
            paths.append(c.path)
        if len(self.cuts) % len(paths) == 0 and all(c.path == paths[i % len(paths)] for i, c in enumerate(self.cuts)) and self.cuts[0].id == "cut-0000000":
            st["cuts"] = ("make_cuts", paths, len(self.cuts) // len(paths))
        st["_rc"] = {}
        return st

    def __setstate__(self, st):
        if isinstance(st["cuts"], tuple) and st["cuts"][0] == "make_cuts":
            st["cuts"] = make_cuts(st["cuts"][1], st["cuts"][2])
        self.__dict__.update(st)

    def __getitem__(self, idx: List[int]):
        t0 = time.perf_counter()
        audio = [read_wav(self.cuts[i].path, self.pcm16) for i in idx]
        out = {"idx": list(idx), "audio": audio, "worker_decode_s": time.perf_counter() - t0}
        if self.packed:  # what lhotse_amd.storage.pack_batch_audio does in the product's dataset: ONE tensor per batch through the worker queue
            import torch

            al = 8 if self.pcm16 else 4
            lens = np.array([a.shape[1] for a in audio], dtype=np.int64)
            offs = np.zeros(len(audio) + 1, dtype=np.int64)
            np.cumsum((lens + al - 1) & ~(al - 1), out=offs[1:])
            buf = torch.empty(int(offs[-1]), dtype=torch.int16 if self.pcm16 else torch.float32)
            flat = buf.numpy()
            for a, o, n in zip(audio, offs, lens):
                flat[o : o + n] = a[0]
            out["audio"], out["lens"], out["offs"] = buf, torch.from_numpy(lens), torch.from_numpy(offs)
        if self.template is not None:
            from lhotse_amd.storage import manifest_fragments

            out["frags"] = [manifest_fragments(self.cuts[i], self.template, self.frame_shift, self._rc) for i in idx]
        return out


def _loader(ds, batches, num_workers: int, context: Optional[str] = None):
    from torch.utils.data import DataLoader

    if context == "forkserver" and num_workers:
        import multiprocessing as mp

        mp.set_forkserver_preload(["torch", "torch.utils.data", "numpy", "plumbing", "lhotse_amd.storage"])  # (as lhotse_amd.storage's driver does)
    return DataLoader(ds, batch_size=None, sampler=batches, num_workers=num_workers, prefetch_factor=4 if num_workers else None,
                      persistent_workers=False, multiprocessing_context=(context if num_workers else None))


# ----------------------------------------------------------------------------------------------------------------------------------
# leg B: lhotse's batch driver around HipFbank, lhotse's own save path
# ----------------------------------------------------------------------------------------------------------------------------------
def hip_batch_numpy_files(ex, cuts: List[Cut], out_dir: str, num_workers: int, keep: Optional[Dict] = None, context: Optional[str] = None) -> Dict:
    import torch

    os.makedirs(out_dir, exist_ok=True)
    store = os.path.join(out_dir, "feats")
    batches = batches_of(cuts)
    busy = {"save": 0.0}

    def _save_worker(man, idx, features):
        t0 = time.perf_counter()
        for i, feat_mat in zip(idx, features):
            c = cuts[i]
            if isinstance(feat_mat, torch.Tensor):
                feat_mat = feat_mat.cpu().numpy()
            key = numpy_files_write(store, c.id, feat_mat)
            assert feat_mat.shape == (FRAMES, NUM_MELS), feat_mat.shape  # validate_features' frame-count contract
            man.write(json.dumps(cut_manifest_dict(c, features_dict(c, ex.name, feat_mat.shape[0], "numpy_files", store, key))) + "\n")
            man.flush()  # cuts_writer.write(cut, flush=True), cut/set.py:2363
            if keep is not None and i in keep:
                keep[i] = feat_mat.copy()
        busy["save"] += time.perf_counter() - t0

    t_ext = t_load = 0.0
    futures = []
    t0 = time.perf_counter()
    t_first = None
    with gzip.open(os.path.join(out_dir, "cuts.jsonl.gz"), "wt") as man, ThreadPoolExecutor(max_workers=1) as executor:
        it = iter(_loader(DecodeDataset(cuts), batches, num_workers, context))  # (the workers start HERE: before `ex` has a plan if it is fresh)
        while True:
            ta = time.perf_counter()
            try:
                batch = next(it)
            except StopIteration:
                break
            tb = time.perf_counter()
            if t_first is None:
                t_first = tb - t0  # worker processes started + the first batch decoded and handed over
            with torch.no_grad():
                features = ex.extract_batch(batch["audio"], sampling_rate=SR)
            t_ext += time.perf_counter() - tb
            t_load += tb - ta
            futures.append(executor.submit(_save_worker, man, batch["idx"], features))
        for f in futures:
            f.result()
    wall = time.perf_counter() - t0
    steady = (len(cuts) - len(batches[0])) / max(wall - (t_first or 0.0), 1e-9)
    return {"cuts_per_s": round(steady, 1), "cuts_per_s_incl_worker_start": round(len(cuts) / wall, 1), "seconds_to_first_batch": round(t_first or 0.0, 3),
            "cuts": len(cuts), "seconds": round(wall, 3), "num_workers": num_workers, "worker_start": context or "fork",
            "main_thread_waiting_for_the_loader_share": round(t_load / wall, 3), "main_thread_extract_share": round(t_ext / wall, 3),
            "save_thread_busy_share": round(busy["save"] / wall, 3)}


# ----------------------------------------------------------------------------------------------------------------------------------
# leg C: the product's bulk driver (native pipeline + striped archive + spliced lines), fed by decoding workers
# ----------------------------------------------------------------------------------------------------------------------------------
def hip_bulk(ex, cuts: List[Cut], out_dir: str, num_workers: int, pcm16: bool = False, half: bool = False, stripes: int = 8, packed: bool = True,
             context: Optional[str] = None) -> Dict:
    from lhotse_amd import storage as S

    os.makedirs(out_dir, exist_ok=True)
    storage = "hip_archive_f16" if half else "hip_archive"
    template = {"type": ex.name, "num_features": NUM_MELS, "frame_shift": ex.frame_shift, "sampling_rate": SR, "storage_type": storage, "storage_path": ""}
    batches = batches_of(cuts)
    busy = {"save": 0.0, "wait": 0.0, "lines": 0.0}
    stats: Dict = {}
    t_load = [0.0]
    t_first = [None]
    acct = _Accounting(ex)

    def timed_batches(loader):
        it = iter(loader)
        while True:
            ta = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            tb = time.perf_counter()
            t_load[0] += tb - ta
            if t_first[0] is None:
                t_first[0] = tb - t0
            yield b
            acct.first_batch()  # (behind the first batch's extraction: the plan and its pipeline exist)

    t0 = time.perf_counter()
    with gzip.open(os.path.join(out_dir, "cuts.jsonl.gz"), "wb") as manifest, \
            S.NativeArchive(os.path.join(out_dir, "feats"), mode="w", np_dtype="<f2" if half else "<f4", stripes=stripes, name=storage) as ar:

        def extract(batch):
            if "lens" in batch:  # packed transport: 1-D views of the batch's one tensor
                buf, offs, lens = batch["audio"], batch["offs"].tolist(), batch["lens"].tolist()
                waves = [buf[o : o + n] for o, n in zip(offs, lens)]
            else:
                waves = [a.reshape(-1) for a in batch["audio"]]
            pending, frames = S._batch_features_pending(ex, waves, SR, None, half=half)
            return batch["frags"], pending, frames

        def save(frags, pending, frames):
            ta = time.perf_counter()
            host = pending.wait()
            tb = time.perf_counter()
            fr = np.ascontiguousarray(frames, dtype=np.int64)
            file_of, byte_off = ar.append(host, fr)
            del host
            pending.release()
            busy["wait"] += tb - ta
            busy["save"] += time.perf_counter() - tb
            return frags, fr, file_of, byte_off

        def lines(frags, fr, file_of, byte_off):
            ta = time.perf_counter()
            blob = ar.lines([f[0] for f in frags], [f[1] for f in frags], fr, np.fromiter((f[2] for f in frags), dtype=np.int64, count=len(frags)),
                            file_of, byte_off, NUM_MELS)
            manifest.write(blob)
            manifest.flush()
            busy["lines"] += time.perf_counter() - ta

        S.pump_batches(timed_batches(_loader(DecodeDataset(cuts, pcm16=pcm16, template=template, frame_shift=ex.frame_shift, packed=packed), batches, num_workers, context)),
                       extract, save, stats=stats, finish=lines)
        acct.stop()
        paths = [str(p) for p in ar.paths]
    wall = time.perf_counter() - t0
    steady = (len(cuts) - len(batches[0])) / max(wall - (t_first[0] or 0.0), 1e-9)
    now = time.time()
    return {"steady_region_epoch": [round(now - wall + (t_first[0] or 0.0), 3), round(now, 3)], "steady_cuts": len(cuts) - len(batches[0]),
            "cuts_per_s": round(steady, 1), "cuts_per_s_incl_worker_start": round(len(cuts) / wall, 1), "seconds_to_first_batch": round(t_first[0] or 0.0, 3),
            "cuts": len(cuts), "seconds": round(wall, 3), "num_workers": num_workers, "input": "int16" if pcm16 else "float32",
            "transport": "one packed tensor per batch" if packed else "one array per cut", "worker_start": context or "fork",
            "storage": storage, "stripes": stripes, "main_thread_waiting_for_the_loader_share": round(t_load[0] / wall, 3),
            "main_thread_submit_share": round(stats.get("extract_s", 0.0) / wall, 3), "main_thread_blocked_on_the_save_threads_share": round(stats.get("wait_s", 0.0) / wall, 3),
            "archive_thread_busy_share": round(busy["save"] / wall, 3), "archive_thread_waiting_for_the_device_share": round(busy["wait"] / wall, 3),
            "manifest_thread_busy_share": round(busy["lines"] / wall, 3), **acct.result(), "archive_paths": paths, "manifest": os.path.join(out_dir, "cuts.jsonl.gz")}


class DecodeIntoSlot:
    """`load_batch` of lhotse_amd.ring_loader.RingLoader for this corpus: the batch's cut objects arrive with the task (as lhotse's sampler
    hands a CutSet per batch to the loader: lhotse_amd.storage.LoadCutsIntoSlot), their WAV files are decoded straight into the ring slot
    (packed), the manifest-line halves serialised; what travels back is lengths + offsets + the halves."""

    def __init__(self, pcm16: bool, template: Dict, frame_shift: float):
        self.pcm16, self.template, self.frame_shift, self._rc = pcm16, template, frame_shift, {}

    def __getstate__(self):
        return {"pcm16": self.pcm16, "template": self.template, "frame_shift": self.frame_shift, "_rc": {}}

    def __call__(self, batch_cuts: List[Cut], out: np.ndarray):
        from lhotse_amd.ring_loader import SlotWriter
        from lhotse_amd.storage import manifest_fragments

        slot = SlotWriter(out)
        for c in batch_cuts:  # (every cut into the slot the moment it is decoded, as lhotse_amd.storage.LoadCutsIntoSlot does)
            if not slot.add(read_wav(c.path, self.pcm16)[0]):
                raise ValueError("batch does not fit its ring slot")
        used, offs, lens = slot.finish()
        frags = None if self.template is None else [manifest_fragments(c, self.template, self.frame_shift, self._rc) for c in batch_cuts]
        return used, {"offs": offs, "lens": lens, "frags": frags}


def hip_ring(ex, cuts: List[Cut], out_dir: str, num_workers: int, pcm16: bool = False, half: bool = False, stripes: int = 8, context: Optional[str] = None,
             pin: bool = True) -> Dict:
    """Leg D: leg C with the ring loader (lhotse_amd/ring_loader.py) in place of the DataLoader -- workers decode into slots of ONE shared
    ring, the main process hands views of a slot to the host pipeline and frees the slot once the library has packed the batch."""
    from lhotse_amd import storage as S
    from lhotse_amd.ring_loader import RingLoader

    os.makedirs(out_dir, exist_ok=True)
    storage = "hip_archive_f16" if half else "hip_archive"
    template = {"type": ex.name, "num_features": NUM_MELS, "frame_shift": ex.frame_shift, "sampling_rate": SR, "storage_type": storage, "storage_path": ""}
    batches = batches_of(cuts)
    item = 2 if pcm16 else 4
    busy = {"save": 0.0, "wait": 0.0, "lines": 0.0}
    stats: Dict = {}
    t_load = [0.0]
    t_first = [None]
    acct = _Accounting(ex)

    def direct_batches():
        try:
            return int(ex.plan.lib.raw("hipfeat_host_pipeline_direct_batches", ex._native_pipe().handle)) if getattr(getattr(ex, "_plan", None), "handle", None) else 0
        except Exception:  # noqa: BLE001
            return 0

    direct0 = direct_batches()
    t0 = time.perf_counter()
    loader = RingLoader(DecodeIntoSlot(pcm16, template, ex.frame_shift), num_workers, slot_bytes=60 * (SAMPLES + 8) * item, start_method=context)
    acct.pids = [p.pid for p in loader._procs]

    def timed(it):
        while True:
            ta = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            tb = time.perf_counter()
            t_load[0] += tb - ta
            if t_first[0] is None:
                t_first[0] = tb - t0
            yield b
            acct.first_batch()

    try:
        with gzip.open(os.path.join(out_dir, "cuts.jsonl.gz"), "wb") as manifest, \
                S.NativeArchive(os.path.join(out_dir, "feats"), mode="w", np_dtype="<f2" if half else "<f4", stripes=stripes, name=storage) as ar:

            def extract(rb):
                pending, frames = S._packed_features_pending(ex, rb.data.view(np.int16 if pcm16 else np.float32), rb.meta["offs"], rb.meta["lens"], SR, half=half)
                if pin:
                    S._pin_ring(loader, ex)  # (slots are page-locked for the GPU as they come into use: uploads straight out of the ring)
                return rb, pending, frames

            def save(rb, pending, frames):
                ta = time.perf_counter()
                host = pending.wait()  # (the library is done with the caller's buffers: the slot goes back to the ring)
                frags = rb.meta["frags"]
                rb.release()
                tb = time.perf_counter()
                fr = np.ascontiguousarray(frames, dtype=np.int64)
                file_of, byte_off = ar.append(host, fr)
                del host
                pending.release()
                busy["wait"] += tb - ta
                busy["save"] += time.perf_counter() - tb
                return frags, fr, file_of, byte_off

            def lines(frags, fr, file_of, byte_off):
                ta = time.perf_counter()
                blob = ar.lines([f[0] for f in frags], [f[1] for f in frags], fr, np.fromiter((f[2] for f in frags), dtype=np.int64, count=len(frags)),
                                file_of, byte_off, NUM_MELS)
                manifest.write(blob)
                manifest.flush()
                busy["lines"] += time.perf_counter() - ta

            S.pump_batches(timed(loader.batches([cuts[i] for i in idx] for idx in batches)), extract, save, stats=stats, finish=lines)
            acct.stop()
            pinned = loader.pinned_slots()
            direct = direct_batches() - direct0
            paths = [str(p) for p in ar.paths]
    finally:
        loader.close()
    wall = time.perf_counter() - t0
    steady = (len(cuts) - len(batches[0])) / max(wall - (t_first[0] or 0.0), 1e-9)
    now = time.time()
    return {"steady_region_epoch": [round(now - wall + (t_first[0] or 0.0), 3), round(now, 3)], "steady_cuts": len(cuts) - len(batches[0]),
            "cuts_per_s": round(steady, 1), "cuts_per_s_incl_worker_start": round(len(cuts) / wall, 1), "seconds_to_first_batch": round(t_first[0] or 0.0, 3),
            "cuts": len(cuts), "seconds": round(wall, 3), "num_workers": num_workers, "input": "int16" if pcm16 else "float32",
            "transport": f"shared ring of {loader.num_slots} slots", "ring_slots_page_locked": pinned, "batches_uploaded_straight_from_the_ring": direct,
            "batches": len(batches), "worker_start": loader.start_method, "storage": storage, "stripes": stripes,
            "main_thread_waiting_for_the_loader_share": round(t_load[0] / wall, 3), "main_thread_submit_share": round(stats.get("extract_s", 0.0) / wall, 3),
            "main_thread_blocked_on_the_save_threads_share": round(stats.get("wait_s", 0.0) / wall, 3), "archive_thread_busy_share": round(busy["save"] / wall, 3),
            "archive_thread_waiting_for_the_device_share": round(busy["wait"] / wall, 3), "manifest_thread_busy_share": round(busy["lines"] / wall, 3),
            **acct.result(), "archive_paths": paths, "manifest": os.path.join(out_dir, "cuts.jsonl.gz")}


def hip_ring_numpy_files(ex, cuts: List[Cut], out_dir: str, num_workers: int, context: Optional[str] = None) -> Dict:
    """Leg F: the product's driver with lhotse's OWN storage (NumpyFilesWriter: one .npy per cut, one manifest dict per cut, flushed per
    batch) behind the ring loader -- lhotse_amd.compute_and_store_features_batch(storage_type=NumpyFilesWriter): leg B's save path, the
    ring's transport, the library's host pipeline."""
    from lhotse_amd import storage as S
    from lhotse_amd.ring_loader import RingLoader

    os.makedirs(out_dir, exist_ok=True)
    store = os.path.join(out_dir, "feats")
    batches = batches_of(cuts)
    busy = {"save": 0.0, "wait": 0.0}
    stats: Dict = {}
    t_load, t_first = [0.0], [None]
    acct = _Accounting(ex)
    t0 = time.perf_counter()
    loader = RingLoader(DecodeIntoSlot(False, None, ex.frame_shift), num_workers, slot_bytes=60 * (SAMPLES + 8) * 4, start_method=context)
    acct.pids = [p.pid for p in loader._procs]

    def timed(it):
        while True:
            ta = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            tb = time.perf_counter()
            t_load[0] += tb - ta
            if t_first[0] is None:
                t_first[0] = tb - t0
            yield b
            acct.first_batch()

    try:
        with gzip.open(os.path.join(out_dir, "cuts.jsonl.gz"), "wt") as man:

            def extract(rb):
                pending, frames = S._packed_features_pending(ex, rb.data.view(np.float32), rb.meta["offs"], rb.meta["lens"], SR, half=False)
                S._pin_ring(loader, ex)
                return rb, pending, frames

            def save(rb, pending, frames):
                ta = time.perf_counter()
                host = pending.wait()
                batch_cuts = rb.spec
                rb.release()
                tb = time.perf_counter()
                row = 0
                for c, t in zip(batch_cuts, frames):
                    feat_mat = host[row : row + t]
                    row += t
                    key = numpy_files_write(store, c.id, feat_mat)
                    assert feat_mat.shape == (FRAMES, NUM_MELS)
                    man.write(json.dumps(cut_manifest_dict(c, features_dict(c, ex.name, t, "numpy_files", store, key))) + "\n")
                man.flush()  # (one flush per batch)
                del host
                pending.release()
                busy["wait"] += tb - ta
                busy["save"] += time.perf_counter() - tb

            S.pump_batches(timed(loader.batches([cuts[i] for i in idx] for idx in batches)), extract, save, stats=stats)
            acct.stop()
    finally:
        loader.close()
    wall = time.perf_counter() - t0
    steady = (len(cuts) - len(batches[0])) / max(wall - (t_first[0] or 0.0), 1e-9)
    return {"cuts_per_s": round(steady, 1), "cuts_per_s_incl_worker_start": round(len(cuts) / wall, 1), "seconds_to_first_batch": round(t_first[0] or 0.0, 3),
            "cuts": len(cuts), "seconds": round(wall, 3), "num_workers": num_workers, "transport": f"shared ring of {loader.num_slots} slots", "storage": "numpy_files",
            "worker_start": loader.start_method, "main_thread_waiting_for_the_loader_share": round(t_load[0] / wall, 3),
            "main_thread_submit_share": round(stats.get("extract_s", 0.0) / wall, 3), "main_thread_blocked_on_the_save_threads_share": round(stats.get("wait_s", 0.0) / wall, 3),
            "save_thread_busy_share": round(busy["save"] / wall, 3), "save_thread_waiting_for_the_device_share": round(busy["wait"] / wall, 3), **acct.result(),
            "manifest": os.path.join(out_dir, "cuts.jsonl.gz")}


def read_back(result: Dict, index: int) -> np.ndarray:
    """Cut `index` of a leg-C run, through the archive reader named by its manifest line."""
    from lhotse_amd import storage as S

    with gzip.open(result["manifest"], "rt") as f:
        for k, ln in enumerate(f):
            if k == index:
                d = json.loads(ln)["features"]
                return S.HipArchiveReader(d["storage_path"]).read(d["storage_key"])
    raise IndexError(index)


def default_workers() -> int:
    """Loader workers for the GPU legs: half of the CPUs this container may use (affinity mask AND cgroup quota), 2 ... 16."""
    n = len(os.sched_getaffinity(0))
    q = cpu_quota()
    if q is not None:
        n = min(n, int(q + 0.5))
    return max(2, min(16, n // 2))


# ----------------------------------------------------------------------------------------------------------------------------------
# one leg in a FRESH process (what a user's script is): the GPU is first touched when the first batch is extracted, i.e. after the
# loader's workers were forked -- the order lhotse's own driver produces (cut/set.py:2302-2304, :2374-2398).  bench.py calls this.
# ----------------------------------------------------------------------------------------------------------------------------------
def main() -> None:
    import argparse
    import shutil
    import tempfile

    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", required=True, choices=["B", "C", "D", "E", "F"])
    ap.add_argument("--jobs", type=int, default=8, help="leg E: job processes of the per-cut driver")
    ap.add_argument("--wav-dir", required=True)
    ap.add_argument("--repeat", type=int, default=50)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--pcm16", action="store_true")
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--stripes", type=int, default=8)
    ap.add_argument("--per-cut-transport", action="store_true")
    ap.add_argument("--no-pin", action="store_true", help="leg D: keep the library's staging copy (do not page-lock the ring's slots)")
    ap.add_argument("--context", default=None)
    ap.add_argument("--gpu-first", action="store_true", help="touch the GPU BEFORE the workers are forked (the hazardous order)")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--start-at", type=float, default=0.0, help="epoch second at which to begin (several processes sharing one GPU start together)")
    a = ap.parse_args()
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import lhotse_amd

    paths = sorted(os.path.join(a.wav_dir, f) for f in os.listdir(a.wav_dir) if f.endswith(".wav"))
    cuts = make_cuts(paths, a.repeat)
    if a.leg == "E":  # the per-cut driver with HipFbank: job processes forked off THIS process, which never touches the GPU
        base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
        with tempfile.TemporaryDirectory(prefix="hipfeat_leg_", dir=base) as td:
            r = cpu_per_cut(cuts, td, a.jobs, extractor="hip")
        print(json.dumps(r), flush=True)
        return
    ex = lhotse_amd.HipFbank()  # (no plan yet: created lazily by the first extraction)
    if a.gpu_first:
        import torch

        ex.extract(torch.zeros(16000), SR)
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    best = None
    if a.start_at:
        time.sleep(max(0.0, a.start_at - time.time()))
    with tempfile.TemporaryDirectory(prefix="hipfeat_leg_", dir=base) as td:
        for k in range(a.passes):  # (pass 2 forks its workers with the plan of pass 1 alive unless a start method is given)
            d = os.path.join(td, f"p{k}")
            if a.leg == "B":
                r = hip_batch_numpy_files(ex, cuts, d, a.workers, context=a.context)
            elif a.leg == "F":
                r = hip_ring_numpy_files(ex, cuts, d, a.workers, context=a.context)
            elif a.leg == "D":
                r = hip_ring(ex, cuts, d, a.workers, pcm16=a.pcm16, half=a.half, stripes=a.stripes, context=a.context, pin=not a.no_pin)
            else:
                r = hip_bulk(ex, cuts, d, a.workers, pcm16=a.pcm16, half=a.half, stripes=a.stripes, packed=not a.per_cut_transport, context=a.context)
            r.pop("archive_paths", None), r.pop("manifest", None)
            r["pass"] = k
            r["gpu_touched_before_the_workers_started"] = bool(a.gpu_first or k > 0)
            shutil.rmtree(d, ignore_errors=True)
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
