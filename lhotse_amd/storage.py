"""
Bulk save path for batch feature extraction.

lhotse's ``CutSet.compute_and_store_features_batch`` (lhotse/cut/set.py:2197-2408) was written around a CPU extractor: its
``_save_worker`` (:2307-2363) stores one object per cut through a ``FeaturesWriter`` (one file open / one HDF5 dataset
per cut), builds a ``Features`` dataclass, validates it, ``fastcopy``-s the cut and serialises it with a recursive
``dataclasses.asdict`` (≈0.5 ms per cut in all).  Behind a GPU extractor that produces a 600 s batch in 0.2 ms this is
what a run spends its time in.  This module is a different design for the same job, in three pieces:

``HipArchiveWriter`` / ``HipArchiveReader`` -- a storage backend registered with lhotse as ``"hip_archive"``
    (``lhotse/features/io.py:288-337``): ONE flat file of little-endian float32 rows per run.  A whole batch is appended
    with one ``write`` (the packed ``(sum T_b, F)`` matrix exactly as it leaves the device); the storage key of a cut is
    self-describing -- ``"<byte offset>:<rows>:<cols>"`` -- so there is no index to maintain and the reader is a
    positioned read (sub-ranges of frames read only their own bytes).  ``Features.load()`` / ``cut.load_features()`` work
    through lhotse's registry as for any other backend.

manifest templates -- the output manifest of a ``MonoCut`` is assembled as a plain dict from parts that are serialised once
    (the recording, the constant ``Features`` fields) instead of dataclass -> copy -> recursive ``asdict`` per cut;
    ``SequentialJsonlWriter.write`` accepts dicts (``lhotse/serialization.py:236-252``).  Other cut types go through
    lhotse's own objects.  The frame-count contract that ``validate_features`` asserts (``lhotse/qa.py:286-301``) is
    checked for the whole batch at once.

``compute_and_store_features_batch`` -- the driver: lhotse's sampler + ``UnsupervisedWaveformDataset`` + ``DataLoader`` load
    the audio (in ``num_workers`` processes), the extractor runs on the main thread, and one background thread appends the
    batch to the archive and emits the manifests, in input order.  Same arguments and resume semantics as the method it
    replaces; any registered ``FeaturesWriter`` can be passed as ``storage_type`` (per-cut ``write`` calls then).

Measured with an instant extractor on 3000 one-second cuts (``tools/bench_storage.py``): see DESIGN.md section 4.4b.
On-GPU lossy compression in the style of lilcom is NOT provided: lilcom is a third-party codec that is not available
offline, so its bit stream cannot be pinned.
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .compat import HAVE_LHOTSE

ARCHIVE_SUFFIX = ".hfa"


def _archive_path(storage_path) -> Path:
    p = Path(storage_path)
    return p if p.suffix == ARCHIVE_SUFFIX else p.with_suffix(p.suffix + ARCHIVE_SUFFIX) if p.suffix else p.with_suffix(ARCHIVE_SUFFIX)


def _parse_key(key: str) -> Tuple[int, int, int, str]:
    """``"<byte offset>:<rows>:<cols>"`` (float32 rows) or ``"<byte offset>:<rows>:<cols>:f16"`` (binary16 rows)."""
    parts = key.split(":")
    return int(parts[0]), int(parts[1]), int(parts[2]), ("<f2" if len(parts) > 3 and parts[3] == "f16" else "<f4")


class _ArchiveWriterImpl:
    """Append-only flat file of float32 (or, ``np_dtype = "<f2"``, binary16) rows; keys are ``"<byte offset>:<rows>:<cols>[:f16]"``."""

    name = "hip_archive"
    np_dtype = "<f4"

    def __init__(self, storage_path, mode: str = "w", *args, **kwargs):
        assert mode in ("w", "a"), mode
        self._path = _archive_path(storage_path)
        self._path.parent.mkdir(parents=True, exist_ok=True)
        self._file = open(self._path, "wb" if mode == "w" else "ab")
        self._offset = self._file.seek(0, os.SEEK_END)
        self._lock = threading.Lock()

    @property
    def storage_path(self) -> str:
        return str(self._path)

    def write(self, key: str, value: np.ndarray) -> str:
        value = np.ascontiguousarray(value)
        assert value.ndim == 2, value.shape
        return self.write_packed(value, [value.shape[0]])[0]

    def write_packed(self, matrix: np.ndarray, frames: Sequence[int]) -> List[str]:
        """Append the packed ``(sum(frames), F)`` matrix of a batch with ONE write; returns one key per item.  A matrix that already
        has the archive's dtype (the driver converts to binary16 on the device) is written as it is."""
        with np.errstate(over="ignore"):
            matrix = np.ascontiguousarray(matrix, dtype=self.np_dtype)
        cols = int(matrix.shape[1])
        item, tag = matrix.dtype.itemsize, (":f16" if matrix.dtype.itemsize == 2 else "")
        if item == 2 and matrix.size and int((matrix.view(np.uint16) & 0x7FFF).max()) >= 0x7C00:
            # binary16 tops out at 65504: log-domain features (|x| < 32) are far inside, linear-domain ones (power spectra, energies) are not
            raise ValueError(f"{self.name}: the batch holds values that are not finite in binary16 (|x| > 65504, inf or nan); this storage is "
                             "meant for log-domain features -- use 'hip_archive' (float32) for linear-domain ones")
        assert int(sum(frames)) == matrix.shape[0], (sum(frames), matrix.shape)
        with self._lock:
            base = self._offset
            self._file.write(memoryview(matrix).cast("B"))
            self._offset += matrix.nbytes
        keys, off = [], base
        for t in frames:
            keys.append(f"{off}:{int(t)}:{cols}{tag}")
            off += int(t) * cols * item
        return keys

    def flush(self):
        self._file.flush()

    def close(self):
        if self._file is not None:
            self._file.close()
            self._file = None

    def __enter__(self):
        return self

    def __exit__(self, *args, **kwargs):
        self.close()


class _ArchiveReaderImpl:
    name = "hip_archive"

    def __init__(self, storage_path, *args, **kwargs):
        self._path = _archive_path(storage_path)
        self._fd = None
        self._lock = threading.Lock()

    def read(self, key: str, left_offset_frames: int = 0, right_offset_frames: Optional[int] = None) -> np.ndarray:
        off, rows, cols, dt = _parse_key(key)
        lo = max(0, int(left_offset_frames))
        hi = rows if right_offset_frames is None else min(rows, int(right_offset_frames))
        n = max(0, hi - lo)
        out = np.empty((n, cols), dtype=dt)
        if n:
            with self._lock:
                if self._fd is None:
                    self._fd = os.open(self._path, os.O_RDONLY)
            got = os.preadv(self._fd, [memoryview(out).cast("B")], off + lo * cols * out.dtype.itemsize)
            if got != out.nbytes:
                raise IOError(f"{self._path}: short read for key {key!r} ({got} of {out.nbytes} bytes)")
        return out if dt == "<f4" else out.astype("<f4")  # lhotse's readers hand out float32

    def __del__(self):
        try:
            if self._fd is not None:
                os.close(self._fd)
        except Exception:
            pass


if HAVE_LHOTSE:
    from lhotse.features.io import FeaturesReader, FeaturesWriter, register_reader, register_writer

    @register_writer
    class HipArchiveWriter(_ArchiveWriterImpl, FeaturesWriter):
        """Registered with lhotse as storage backend ``"hip_archive"`` (writer side)."""

        name = "hip_archive"

    @register_reader
    class HipArchiveReader(_ArchiveReaderImpl, FeaturesReader):
        """Registered with lhotse as storage backend ``"hip_archive"`` (reader side)."""

        name = "hip_archive"

    @register_writer
    class HipArchiveF16Writer(_ArchiveWriterImpl, FeaturesWriter):
        """``"hip_archive_f16"``: the same archive with binary16 rows -- half the file, and with the Hip* extractors half the
        device -> host traffic (the batch driver converts on the device).  Lossy like the reference's default lilcom storage: log-domain
        features (|x| < 32) keep 2^-6 ... 2^-7 absolute, the error of the lilcom fixture the reference ships.  NOT for linear-domain
        features: binary16 overflows above 65504 and flushes below 6e-8, so power spectra / unlogged energies would be stored as inf or 0.
        ``write`` / ``write_packed`` raise on values that are not finite in binary16, and the batch driver refuses the combination of
        this storage with a non-log extractor (``HipSpectrogram``) up front."""

        name = "hip_archive_f16"
        np_dtype = "<f2"

    @register_reader
    class HipArchiveF16Reader(_ArchiveReaderImpl, FeaturesReader):
        name = "hip_archive_f16"

else:  # usable on their own (tests, tools) without lhotse

    class HipArchiveWriter(_ArchiveWriterImpl):
        pass

    class HipArchiveReader(_ArchiveReaderImpl):
        pass

    class HipArchiveF16Writer(_ArchiveWriterImpl):
        name = "hip_archive_f16"
        np_dtype = "<f2"

    HipArchiveF16Reader = HipArchiveReader


# ---- manifest templates ---------------------------------------------------------------------------------------------------
def _features_dict(template: Dict, cut, num_frames: int, storage_key: str) -> Dict:
    """``Features(...).to_dict()`` (lhotse/features/base.py:444-474, asdict_nonull field order) without the dataclass."""
    d = dict(template)  # type, num_features, frame_shift, sampling_rate, storage_type, storage_path in dataclass order
    d["num_frames"] = int(num_frames)
    d["sampling_rate"] = cut.sampling_rate
    d["start"] = cut.start
    d["duration"] = cut.duration
    d["storage_key"] = storage_key
    return d


def _nonull(obj):
    """``lhotse.utils.asdict_nonull`` (lhotse/utils.py:166-182) for manifests whose leaves are JSON scalars: the same
    recursion over dataclass fields, lists and dicts, dropping None fields -- without ``dataclasses.asdict``'s deep copy of
    every leaf, which is where the reference spends most of a cut's serialisation."""
    fields = getattr(obj, "__dataclass_fields__", None)
    if fields is not None:
        out = {}
        for name in fields:
            v = getattr(obj, name)
            if v is not None:
                out[name] = _nonull(v)
        return out
    if isinstance(obj, (list, tuple)):
        return [_nonull(v) for v in obj]
    if isinstance(obj, dict):
        return {k: _nonull(v) for k, v in obj.items()}
    return obj


def _recording_dict(rec) -> Dict:
    if getattr(rec, "transforms", None) is not None:
        return rec.to_dict()  # transforms may be objects with their own to_dict (lhotse/audio/recording.py:365-371)
    return _nonull(rec)


_FEATURE_FIELD_ORDER = ("type", "num_frames", "num_features", "frame_shift", "sampling_rate", "start", "duration", "storage_type",
                        "storage_path", "storage_key", "recording_id", "channels")


_PLAIN_SCALARS = (str, int, float, bool, type(None))
_NOT_PLAIN = object()


def _plain_copy(v):
    """A copy of ``v`` if it is made of JSON scalars, lists / tuples and str-keyed dicts only (what ``dataclasses.asdict`` would
    hand back unchanged, None leaves included: lhotse/utils.py:166-182 filters dataclass fields, not plain dicts); else _NOT_PLAIN."""
    if isinstance(v, _PLAIN_SCALARS):
        return v
    if type(v) in (list, tuple):
        out = [_plain_copy(x) for x in v]
        return _NOT_PLAIN if any(x is _NOT_PLAIN for x in out) else type(v)(out)
    if type(v) is dict:
        out = {}
        for k, x in v.items():
            x = _plain_copy(x)
            if x is _NOT_PLAIN or not isinstance(k, str):
                return _NOT_PLAIN
            out[k] = x
        return out
    return _NOT_PLAIN


# how many cuts took the template path / lhotse's own serialiser (tests assert that the fast path really runs behind lhotse's sampler,
# which attaches a `dataloading_info` custom field to every cut: lhotse/dataset/sampling/base.py:473-487)
TEMPLATE_STATS = {"template": 0, "fallback": 0}
_REC_CACHE_MAX = 4096
_SAVE_BACKLOG = 8  # batches in flight between the extractor and the save thread


def _mono_cut_dict(cut, feats: Dict, rec_cache: Dict[str, Tuple[object, Dict]]) -> Optional[Dict]:
    """``fastcopy(cut, features=Features(...)).to_dict()`` for a MonoCut (lhotse/cut/data.py:90-98) assembled from parts; the
    recording's dict is built once per Recording OBJECT (the cache is keyed by the recording id and holds the object it was built
    from: an ``id()`` key alone would be reused by CPython for another object once a lazily loaded batch is freed).
    None = leave this cut to lhotse's own serialiser."""
    custom = None
    if cut.custom is not None:
        custom = _plain_copy(cut.custom)  # e.g. {"dataloading_info": {...}}; manifests / arrays in custom fields go the slow way
        if custom is _NOT_PLAIN:
            return None
    feats = dict(feats)
    feats["recording_id"] = cut.recording_id
    feats["channels"] = cut.channel
    feats = {k: feats[k] for k in _FEATURE_FIELD_ORDER if feats.get(k) is not None}
    d = {"id": cut.id, "start": cut.start, "duration": cut.duration, "channel": cut.channel}
    d["supervisions"] = [_nonull(s) if s.custom is None and s.alignment is None else s.to_dict() for s in cut.supervisions]
    d["features"] = feats
    rec = cut.recording
    if rec is not None:
        entry = rec_cache.get(rec.id)
        if entry is None or entry[0] is not rec:
            if len(rec_cache) >= _REC_CACHE_MAX:
                rec_cache.clear()
            entry = rec_cache[rec.id] = (rec, _recording_dict(rec))
        d["recording"] = entry[1]
    if custom is not None:
        d["custom"] = custom
    d["type"] = "MonoCut"
    return d


def _batch_features_on_host(extractor, waves, sampling_rate: int, lengths, half: bool = False) -> Tuple[np.ndarray, List[int]]:
    """Packed ``(sum T_b, F)`` host matrix + per-cut frame counts (one D2H transfer for the Hip* extractors).  `half`: binary16,
    converted on the device in front of the transfer when the extractor's pipeline is available (on the host otherwise)."""
    if lengths is None and hasattr(extractor, "_extract_items") and hasattr(extractor, "_to_host"):
        from .extractors import _as_1d_float

        extractor._check_sr(sampling_rate)
        items = [_as_1d_float(w.squeeze() if w.ndim > 1 else w, "compute_and_store_features_batch()") for w in waves]
        zero_pad = getattr(extractor.config, "edge_rule", "reflect") == "batch_zero_pad"
        pmax = max(int(x.shape[0]) for x in items) if zero_pad else None
        with torch.no_grad():
            on_host = all(not isinstance(x, torch.Tensor) or x.device.type == "cpu" for x in items)
            if on_host and hasattr(extractor, "_host_items_to_host") and extractor.plan.device.type == "cuda":
                host, frames = extractor._host_items_to_host(items, pmax, half=half)  # chunked H2D / kernel / D2H pipeline, one pinned result
            else:
                packed, frames = extractor._extract_items(items, pmax)
                host = extractor._to_host(packed)
        return host.numpy(), [int(t) for t in frames]
    with torch.no_grad():
        feats = extractor.extract_batch(waves, sampling_rate=sampling_rate, lengths=lengths)
    if isinstance(feats, (np.ndarray, torch.Tensor)) and feats.ndim == 2:
        feats = [feats]
    mats = [f.cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f) for f in feats]
    return (np.concatenate(mats, axis=0) if len(mats) != 1 else mats[0]), [int(m.shape[0]) for m in mats]


def pump_batches(batches, extract, save, backlog: int = None, stats: Optional[Dict] = None, finish=None) -> None:
    """The loop of the batch driver (lhotse/cut/set.py:2365-2404): the calling thread runs ``extract(batch)`` -> arguments of ``save`` (or
    None to skip the batch), ONE background thread runs ``save(*args)`` behind it -- and, with ``finish``, a second one runs
    ``finish(*save's result)`` behind that (the archive write and the manifest lines of a batch cost about the same, and neither needs
    the other's thread: measured on the MI355X box, profiles/r04_bulk_save.json).  Both keep submission order.  At most ``backlog``
    (_SAVE_BACKLOG) finished batches wait for the save thread: each holds its page-locked result, and a failed save (disk full) stops the
    run at the next batch instead of after the whole corpus (the reference collects its futures at the very end, cut/set.py:2400-2404).
    ``stats`` (optional) receives the seconds the calling thread spent extracting (``extract_s``) and blocked on the background threads
    (``wait_s``) -- bench.py's ``--config bulk_save`` reads them to say which stage binds."""
    import time
    from collections import deque

    backlog = _SAVE_BACKLOG if backlog is None else backlog
    futures = deque()
    t_ext = t_wait = 0.0
    # the finisher is the OUTER context: on the way out (also on an exception) the saver drains first, while the finisher still accepts
    # what its stages submit -- the other order shut the finisher down under queued stages, whose rows then reached the archive without
    # manifest lines
    with ThreadPoolExecutor(max_workers=1) as finisher, ThreadPoolExecutor(max_workers=1) as saver:

        def stage(*item):
            res = save(*item)
            return None if finish is None else finisher.submit(finish, *res)

        def collect(fut):
            inner = fut.result()
            if inner is not None:
                inner.result()

        try:
            for batch in batches:
                t0 = time.perf_counter()
                item = extract(batch)
                t1 = time.perf_counter()
                t_ext += t1 - t0
                if item is None:
                    continue
                futures.append(saver.submit(stage, *item))
                while len(futures) > backlog or (futures and futures[0].done() and (futures[0].exception() is not None or futures[0].result() is None or futures[0].result().done())):
                    collect(futures.popleft())
                t_wait += time.perf_counter() - t1
            t1 = time.perf_counter()
            while futures:
                collect(futures.popleft())
            t_wait += time.perf_counter() - t1
        except BaseException:
            # a failed run: batches that have not started saving are dropped (nothing of them reaches the archive); the one being saved
            # finishes with its manifest lines.  Rows of the batch that FAILED may be in the archive without lines -- harmless for the
            # readers (they go by the manifest) and overwritten space-wise by nothing: a resumed run appends behind them.
            for f in futures:
                f.cancel()
            raise
    if stats is not None:
        stats["extract_s"] = stats.get("extract_s", 0.0) + t_ext
        stats["wait_s"] = stats.get("wait_s", 0.0) + t_wait


def compute_and_store_features_batch(
    cuts,
    extractor,
    storage_path,
    manifest_path=None,
    batch_duration: float = 600.0,
    num_workers: int = 4,
    collate: bool = False,
    augment_fn: Optional[Callable] = None,
    storage_type=None,
    overwrite: bool = False,
):
    """``CutSet.compute_and_store_features_batch`` with the bulk save path of this module (same arguments; ``storage_type``
    defaults to ``HipArchiveWriter``).  Returns the CutSet with the ``Features`` manifests attached."""
    if not HAVE_LHOTSE:
        raise ImportError("compute_and_store_features_batch produces lhotse manifests: install lhotse")
    from lhotse import CutSet, Features, MonoCut
    from lhotse.cut import MixedCut, PaddingCut
    from lhotse.dataset import SimpleCutSampler, UnsupervisedWaveformDataset
    from lhotse.qa import validate_features
    from lhotse.utils import compute_num_frames, fastcopy
    from torch.utils.data import DataLoader

    storage_type = storage_type or HipArchiveWriter  # NB manifests that name "hip_archive" need `import lhotse_amd` to be read back
    if getattr(storage_type, "name", None) == "numpy_files":  # lhotse/cut/set.py:2285-2288
        storage_path = Path(storage_path)
        if storage_path.exists() and storage_path.is_file():
            storage_path = storage_path.with_name(f"{storage_path.name}_storage")
    frame_shift = extractor.frame_shift
    manifest = CutSet.open_writer(manifest_path, overwrite=overwrite)
    # rank / world pinned: under torchrun lhotse's samplers would otherwise split (and pad with duplicated cuts) what they are given once
    # more (lhotse/dataset/sampling/base.py:152-163); sharding over GPUs is explicit here (lhotse_amd.compute_and_store_features_sharded)
    sampler = SimpleCutSampler(cuts, max_duration=batch_duration, world_size=1, rank=0)
    sampler.filter(lambda cut: cut.id not in manifest.ignore_ids)  # resume: skip what the manifest already holds
    loader = DataLoader(UnsupervisedWaveformDataset(collate=collate), batch_size=None, sampler=sampler, num_workers=num_workers)
    rec_cache: Dict[str, Tuple[object, Dict]] = {}

    def save(writer, batch_cuts, host: np.ndarray, frames: List[int], template: Dict):
        # the frame-count contract of validate_features (qa.py:286-301), for the whole batch
        for c, t in zip(batch_cuts, frames):
            if isinstance(c, PaddingCut):
                continue
            hop = round(frame_shift * c.sampling_rate, ndigits=12)  # qa.py:286-291: the hop must be a whole number of samples
            if not float(hop).is_integer():
                raise AssertionError(f"cut {c.id}: frame_shift {frame_shift} s is {hop} samples at {c.sampling_rate} Hz (not an integer)")
            if compute_num_frames(c.duration, frame_shift, c.sampling_rate) != t:
                raise AssertionError(f"cut {c.id}: {t} frames for {c.duration} s at frame_shift {frame_shift} (lhotse expects "
                                     f"{compute_num_frames(c.duration, frame_shift, c.sampling_rate)})")
        stored = [i for i, c in enumerate(batch_cuts) if not isinstance(c, PaddingCut)]
        bounds = np.concatenate([[0], np.cumsum(frames)])
        if hasattr(writer, "write_packed") and len(stored) == len(batch_cuts):
            keys = writer.write_packed(host, frames)
        else:
            keys = [None] * len(batch_cuts)
            for i in stored:
                keys[i] = writer.write(batch_cuts[i].id, host[int(bounds[i]) : int(bounds[i + 1])])
        if hasattr(writer, "flush"):
            writer.flush()
        return batch_cuts, frames, keys, int(host.shape[1]), template  # -> write_manifests, on the second background thread

    def write_manifests(batch_cuts, frames: List[int], keys: List[str], num_features: int, template: Dict) -> None:
        for i, c in enumerate(batch_cuts):
            if isinstance(c, PaddingCut):
                manifest.write(fastcopy(c, num_frames=frames[i], num_features=num_features, frame_shift=frame_shift))
                continue
            fd = _features_dict(template, c, frames[i], keys[i])
            # (an in-memory manifest -- no manifest_path -- keeps the objects it is given: no templates there)
            out = _mono_cut_dict(c, fd, rec_cache) if type(c) is MonoCut and manifest_path is not None else None
            TEMPLATE_STATS["fallback" if out is None else "template"] += 1
            if out is None:  # mixed cuts, custom fields: lhotse's own objects (lhotse/cut/set.py:2335-2363)
                fm = Features(recording_id=c.id if isinstance(c, MixedCut) else c.recording_id, channels=0 if isinstance(c, MixedCut) else c.channel,
                              **{k: v for k, v in fd.items()})
                validate_features(fm)
                if isinstance(c, MixedCut):
                    out = MonoCut(id=c.id, start=0, duration=c.duration, channel=0, features=fm, recording=None,
                                  supervisions=[fastcopy(s, recording_id=c.id, channel=0) for s in c.supervisions])
                else:
                    out = fastcopy(c, features=fm)
            manifest.write(out)
        if getattr(manifest, "file", None) is not None:
            manifest.file.flush()  # one flush per batch

    if getattr(storage_type, "np_dtype", "<f4") == "<f2" and not getattr(extractor, "log_domain", True):
        raise ValueError(f"storage '{storage_type.name}' keeps binary16 rows, which cannot hold the linear-domain output of '{extractor.name}' "
                         "(overflow above 65504, flush to zero below 6e-8): use 'hip_archive'")
    with manifest, storage_type(storage_path, mode="w" if overwrite else "a") as writer:
        state = {"template": None}

        def extract(batch):
            batch_cuts, waves = batch["cuts"], batch["audio"]
            lens = batch["audio_lens"] if collate else None
            if len(batch_cuts) == 0:
                return None
            sr = batch_cuts[0].sampling_rate
            assert all(c.sampling_rate == sr for c in batch_cuts)
            if augment_fn is not None:
                waves = [augment_fn(w, c.sampling_rate) for c, w in zip(batch_cuts, waves)]
            host, frames = _batch_features_on_host(extractor, waves, sr, lens, half=getattr(writer, "np_dtype", "<f4") == "<f2")
            if state["template"] is None:
                state["template"] = {"type": extractor.name, "num_features": int(host.shape[1]), "frame_shift": frame_shift, "sampling_rate": sr,
                                     "storage_type": writer.name, "storage_path": str(writer.storage_path)}
            return writer, list(batch_cuts), host, frames, state["template"]

        pump_batches(loader, extract, save, finish=write_manifests)
    return manifest.open_manifest()
