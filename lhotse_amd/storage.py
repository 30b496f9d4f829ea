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
import warnings
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
# ---- manifest lines without the interpreter (round 5) --------------------------------------------------------------------------
# A cut's manifest line depends on the extraction in ONE place: where its rows are (storage_path / storage_key).  Everything else --
# ids, supervisions, the recording, custom fields, and also the Features fields num_frames / start / duration, which are functions of
# the cut -- is known where the cut is LOADED.  So the line is serialised there (lhotse's loader workers: other processes, other GILs),
# cut in two around the storage fields, and the save path only splices the key in: libhipfeat's hipfeat_manifest_lines, one call per
# batch with the GIL released, which also enforces the frame-count contract of validate_features (lhotse/qa.py:286-301) on the
# extractor's actual frame counts.  Per batch the main process then runs a handful of C calls (archive append, line splice, zlib,
# write) and no per-cut Python at all -- the per-cut Python of three threads sharing one GIL was what held the offline path at
# 10-15 k cuts/s behind an extractor that sustains 50-120 k (profiles/r04_bulk_save.json).
PATH_TOKEN = "@@HIPFEAT_STORAGE_PATH@@"
KEY_TOKEN = "@@HIPFEAT_STORAGE_KEY@@"
_SPLICE = PATH_TOKEN + '", "storage_key": "' + KEY_TOKEN  # storage_path and storage_key are neighbours in Features.to_dict()


def expected_num_frames(duration: float, frame_shift: float, sampling_rate: int) -> int:
    """lhotse.utils.compute_num_frames (lhotse/utils.py:410-421), the count validate_features holds a Features manifest to."""
    hop = round(frame_shift * sampling_rate)
    return int((round(duration * sampling_rate) + hop // 2) // hop)


def manifest_fragments(cut, template: Dict, frame_shift: float, rec_cache: Dict, mono_type=None) -> Optional[Tuple[bytes, bytes, int]]:
    """(head, tail, frames the line states) of a MonoCut's manifest line -- `json.dumps(cut-with-features.to_dict(), ensure_ascii=False)`
    as SequentialJsonlWriter.write produces it (lhotse/serialization.py:236-252), cut around the storage fields -- or None when the cut
    has to go through lhotse's own objects (mixed / padding cuts, custom fields that are not plain JSON, a hop that is not a whole
    number of samples: the checks of the per-cut path then word the error)."""
    import json

    if mono_type is not None and type(cut) is not mono_type:
        return None
    hop = round(frame_shift * cut.sampling_rate, ndigits=12)  # qa.py:286-291
    if not float(hop).is_integer():
        return None
    frames = expected_num_frames(cut.duration, frame_shift, cut.sampling_rate)
    t = dict(template)
    t["storage_path"] = PATH_TOKEN
    d = _mono_cut_dict(cut, _features_dict(t, cut, frames, KEY_TOKEN), rec_cache)
    if d is None:
        return None
    line = json.dumps(d, ensure_ascii=False)
    head, sep, tail = line.partition(_SPLICE)
    if not sep or _SPLICE in tail or PATH_TOKEN in head or KEY_TOKEN in head:
        return None
    return head.encode("utf-8"), tail.encode("utf-8"), frames


def _concat(parts: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in parts], out=off[1:])
    blob = np.frombuffer(b"".join(parts) or b"\0", dtype=np.uint8)
    return blob, off


def stripe_paths(storage_path, stripes: int) -> List[Path]:
    """File 0 is the archive path itself (``feats.hfa``), file k > 0 ``feats.<k>.hfa``."""
    base = _archive_path(storage_path)
    stem = str(base)[: -len(ARCHIVE_SUFFIX)]
    return [base] + [Path(f"{stem}.{k}{ARCHIVE_SUFFIX}") for k in range(1, int(stripes))]


class NativeArchive:
    """The batch driver's writer for the ``hip_archive`` / ``hip_archive_f16`` storages: libhipfeat's hipfeat_archive_* (append of a
    whole batch, striped over ``stripes`` files by as many writer threads, off the GIL) and hipfeat_manifest_lines.  The files are what
    ``HipArchiveWriter.write_packed`` would have written (same keys, same bytes); with one stripe there is ONE file, as before."""

    def __init__(self, storage_path, mode: str = "w", np_dtype: str = "<f4", stripes: int = 1, name: str = "hip_archive"):
        import ctypes
        import json

        from . import _lib

        assert mode in ("w", "a"), mode
        self.lib = _lib.load()
        self.name, self.np_dtype = name, np_dtype
        self.item = 2 if np_dtype == "<f2" else 4
        self.paths = stripe_paths(storage_path, max(1, int(stripes)))
        self.paths[0].parent.mkdir(parents=True, exist_ok=True)
        if mode == "w":  # an earlier run with MORE stripes left feats.<k>.hfa files this run would neither truncate nor name
            stem = str(self.paths[0])[: -len(ARCHIVE_SUFFIX)]
            k = len(self.paths)
            while Path(f"{stem}.{k}{ARCHIVE_SUFFIX}").exists():
                Path(f"{stem}.{k}{ARCHIVE_SUFFIX}").unlink()
                k += 1
        raw = [str(p).encode("utf-8") for p in self.paths]
        arr = (ctypes.c_char_p * len(raw))(*raw)
        h = np.zeros(1, dtype=np.uint64)
        self.handle = 0
        self.lib.check("hipfeat_archive_open", ctypes.addressof(arr), len(raw), 1 if mode == "a" else 0, _lib.addr(h))
        self.handle = int(h[0])
        # mid[k] = `<JSON-escaped path k>", "storage_key": "`
        self._mids, self._mid_off = _concat([(json.dumps(str(p), ensure_ascii=False)[1:-1] + '", "storage_key": "').encode("utf-8") for p in self.paths])
        self._append, self._lines = self.lib.fn("hipfeat_archive_append"), self.lib.fn("hipfeat_manifest_lines")

    @property
    def storage_path(self) -> str:
        return str(self.paths[0])

    def append(self, matrix: np.ndarray, frames: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Append the packed (sum(frames), F) matrix of a batch -> (file index, byte offset) per cut."""
        from . import _lib

        with np.errstate(over="ignore"):
            matrix = np.ascontiguousarray(matrix, dtype=self.np_dtype)  # (no copy for what the driver hands over: the device converted already)
        assert matrix.ndim == 2, matrix.shape
        frames = np.ascontiguousarray(frames, dtype=np.int64)
        assert int(frames.sum()) == matrix.shape[0], (int(frames.sum()), matrix.shape)
        file_of, byte_off = np.zeros(len(frames), dtype=np.int32), np.zeros(len(frames), dtype=np.int64)
        st = self._append(self.handle, matrix.ctypes.data, len(frames), _lib.addr(frames), int(matrix.shape[1]), self.item, _lib.addr(file_of), _lib.addr(byte_off))
        if st != 0:
            msg = self.lib.last_error()
            if "not finite in binary16" in msg:  # (checked by the writer threads on their own runs, before anything is written: as write_packed)
                raise ValueError(f"{self.name}: {msg}; this storage is meant for log-domain features -- use 'hip_archive' (float32) for linear-domain ones")
            raise _lib.HipFeatError(int(st), msg)
        return file_of, byte_off

    def lines(self, heads: Sequence[bytes], tails: Sequence[bytes], frames: np.ndarray, expected: Optional[np.ndarray], file_of: np.ndarray,
              byte_off: np.ndarray, cols: int) -> bytes:
        """The JSONL lines of the batch (bytes, one b"\\n"-terminated line per cut); raises when a frame count differs from `expected`."""
        from . import _lib

        hb, ho = _concat(heads)
        tb, to = _concat(tails)
        frames = np.ascontiguousarray(frames, dtype=np.int64)
        exp = None if expected is None else np.ascontiguousarray(expected, dtype=np.int64)
        cap = int(ho[-1] + to[-1]) + len(heads) * (int(np.diff(self._mid_off).max()) + 80)
        out, n = np.empty(cap, dtype=np.uint8), np.zeros(1, dtype=np.int64)
        st = self._lines(hb.ctypes.data, _lib.addr(ho), tb.ctypes.data, _lib.addr(to), len(heads), _lib.addr(frames), _lib.addr(exp), self._mids.ctypes.data,
                         _lib.addr(self._mid_off), len(self.paths), _lib.addr(file_of), _lib.addr(byte_off), int(cols), self.item, out.ctypes.data, cap, _lib.addr(n))
        if st != 0:
            msg = self.lib.last_error()
            if "frame-count contract" in msg:
                raise AssertionError(msg)  # what validate_features raises in the per-cut path
            raise _lib.HipFeatError(int(st), msg)
        return out[: int(n[0])].tobytes()

    def keys(self, frames: np.ndarray, file_of: np.ndarray, byte_off: np.ndarray, cols: int) -> List[Tuple[str, str]]:
        """(storage_path, storage_key) per cut -- for the cuts of a batch that go through lhotse's own objects."""
        tag = ":f16" if self.item == 2 else ""
        return [(str(self.paths[int(k)]), f"{int(o)}:{int(t)}:{int(cols)}{tag}") for t, k, o in zip(frames, file_of, byte_off)]

    def size(self, k: int = 0) -> int:
        return int(self.lib.raw("hipfeat_archive_size", self.handle, int(k)))

    def flush(self):
        pass  # appends are pwrite()s: nothing is buffered in the process

    def close(self):
        if self.handle:
            h, self.handle = self.handle, 0
            self.lib.check("hipfeat_archive_close", h)

    def __enter__(self):
        return self

    def __exit__(self, *args, **kwargs):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_FRAGMENTING_CLS = None


def _fragmenting_dataset_class():
    """`FragmentingWaveformDataset`, created on first use (importing lhotse.dataset is not free) and reachable as a module attribute
    (`__getattr__` below), which is what pickle needs to ship an instance to spawned DataLoader workers."""
    global _FRAGMENTING_CLS
    if _FRAGMENTING_CLS is None:
        from lhotse import MonoCut
        from lhotse.dataset import UnsupervisedWaveformDataset

        class FragmentingWaveformDataset(UnsupervisedWaveformDataset):
            """lhotse.dataset.UnsupervisedWaveformDataset (lhotse/dataset/unsupervised.py:44-98) whose batches also carry
            `hipfeat_fragments`: per cut the two halves of its manifest line + the frame count they state (`manifest_fragments`), or None
            for cuts that have to go through lhotse's own objects."""

            def __init__(self, collate: bool, template: Optional[Dict], frame_shift: float, pack: bool = True):
                super().__init__(collate=collate)
                self.hipfeat_template, self.hipfeat_frame_shift, self.hipfeat_pack = template, frame_shift, pack

            def __getitem__(self, batch_cuts):
                batch = super().__getitem__(batch_cuts)
                cache = self.__dict__.setdefault("_hipfeat_rec_cache", {})
                t = self.hipfeat_template
                batch["hipfeat_fragments"] = None if t is None else [manifest_fragments(c, t, self.hipfeat_frame_shift, cache, MonoCut) for c in batch["cuts"]]
                if self.hipfeat_pack and not self.collate:
                    pack_batch_audio(batch)
                return batch

        FragmentingWaveformDataset.__module__, FragmentingWaveformDataset.__qualname__ = __name__, "FragmentingWaveformDataset"
        _FRAGMENTING_CLS = FragmentingWaveformDataset
    return _FRAGMENTING_CLS


def __getattr__(name):  # PEP 562: `lhotse_amd.storage.FragmentingWaveformDataset` exists once asked for
    if name == "FragmentingWaveformDataset" and HAVE_LHOTSE:
        return _fragmenting_dataset_class()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def pack_batch_audio(batch: Dict) -> None:
    """Round 6: the un-collated batch of lhotse's waveform dataset is a LIST of (1, T) float32 arrays; through a DataLoader's worker
    queue every one of them becomes its own shared-memory segment (created, filled, its descriptor passed, mapped and unmapped again in
    the main process): measured with 64 x 10 s WAV files, that transport -- not decoding, not the extractor -- bounds the batch driver at
    1-3 k cuts/s (profiles/r06_bench_plumbing.json, leg B).  Here, still inside the worker, the cuts are packed into ONE float32 tensor
    (every cut on a 16-byte boundary, as the extractor's staging wants them) + their lengths: one segment per batch.  The main process
    takes 1-D views of it (`unpack_batch_audio`).  Left alone: anything that is not a list of mono float32 arrays."""
    audio = batch.get("audio")
    if not isinstance(audio, (list, tuple)) or len(audio) == 0:
        return
    if not all(isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[0] == 1 for a in audio):
        return
    lens = np.array([a.shape[1] for a in audio], dtype=np.int64)
    offs = np.zeros(len(audio) + 1, dtype=np.int64)
    np.cumsum((lens + 3) & ~3, out=offs[1:])
    buf = torch.empty(int(offs[-1]), dtype=torch.float32)
    flat = buf.numpy()
    for a, o, n in zip(audio, offs, lens):
        flat[o : o + n] = a[0]
    batch["audio"], batch["hipfeat_lens"] = buf, torch.from_numpy(lens)


def unpack_batch_audio(batch: Dict):
    """The waveforms of a batch as the extractor takes them: 1-D views of the packed tensor (no copy), or whatever the dataset delivered."""
    lens = batch.get("hipfeat_lens")
    if lens is None:
        return batch["audio"]
    buf, o, out = batch["audio"], 0, []
    for n in lens.tolist():
        out.append(buf[o : o + n])
        o += (n + 3) & ~3
    return out


def pcm16_wav_segment(cut, mono_type=None) -> Optional[np.ndarray]:
    """The int16 samples of which ``cut.load_audio()`` returns ``x / 32768`` as float32 -- read with the stdlib ``wave`` module -- or None
    when that cannot be guaranteed: only a MonoCut over a recording that is ONE local mono 16-bit PCM ``.wav`` file without transforms
    qualifies, and only when the file holds exactly the samples lhotse would end up with (``Recording.load_audio`` +
    ``assert_and_maybe_fix_num_samples``, lhotse/audio/recording.py:389-492, 1032-1068: anything that would make lhotse pad, trim, warn
    or raise is left to lhotse).  The device converts int16 -> float32 as x / 32768 (``hipfeat_pcm16_to_float``, exact), which is what the
    audio backends do on the host: same features, a third of the loader's memory traffic and half the bytes over PCIe."""
    import wave
    from math import isclose

    from lhotse.utils import compute_num_samples

    if mono_type is not None and type(cut) is not mono_type:
        return None
    rec = getattr(cut, "recording", None)
    if rec is None or getattr(rec, "transforms", None) or len(rec.sources) != 1 or getattr(rec, "has_video", False):
        return None
    src = rec.sources[0]
    if src.type != "file" or not str(src.source).lower().endswith(".wav") or list(src.channels) != [0] or list(rec.channel_ids) != [0] or cut.channel != 0:
        return None
    sr = rec.sampling_rate
    start = compute_num_samples(cut.start, sr)
    whole = isclose(cut.duration, rec.duration, abs_tol=1e-3)  # (lhotse then reads to the end of the file and fixes the count up)
    want = compute_num_samples(cut.duration if cut.duration is not None else rec.duration - cut.start, sr)
    try:
        with wave.open(str(src.source), "rb") as f:
            if f.getnchannels() != 1 or f.getsampwidth() != 2 or f.getframerate() != sr or f.getcomptype() != "NONE":
                return None
            n = f.getnframes()
            if start + want > n or (whole and n - start != want):
                return None
            f.setpos(start)
            raw = f.readframes(want)
    except (OSError, EOFError, wave.Error):
        return None
    x = np.frombuffer(raw, dtype="<i2")
    return x if x.shape[0] == want else None


class LoadCutsIntoSlot:
    """`load_batch` of ``lhotse_amd.ring_loader.RingLoader`` for lhotse cuts: what ``FragmentingWaveformDataset.__getitem__`` does with
    ``collate=False`` (lhotse/dataset/unsupervised.py:66-80: validate, load every cut's audio, drop the cuts whose audio fails to load the
    way ``suppress_audio_loading_errors`` words it; + the halves of every manifest line), with the samples written straight into a slot of
    the shared ring instead of into fresh arrays that travel by pickle.  Runs in the loader's worker processes.  What comes back is small:
    ``{"kept": positions of the cuts that loaded, "offs" / "lens": element offsets / lengths in the slot, "frags": line halves}``; a batch
    that is not all mono float32, or does not fit a slot (one cut longer than ``batch_duration``), travels as ``"audio"`` by pickle -- the
    DataLoader's transport, for that batch only."""

    def __init__(self, template: Optional[Dict], frame_shift: float, pcm16: bool = False):
        self.template, self.frame_shift, self.pcm16 = template, frame_shift, pcm16

    def __getstate__(self):
        return {"template": self.template, "frame_shift": self.frame_shift, "pcm16": self.pcm16}

    def _pcm16_batch(self, cuts, out: np.ndarray, mono_type):
        """The whole batch as int16 PCM in the slot (``pcm16_wav_segment`` for EVERY cut), or None: the batch then takes lhotse's route."""
        from .ring_loader import SlotWriter

        slot = SlotWriter(out)
        for c in cuts:
            x = pcm16_wav_segment(c, mono_type)
            if x is None or not slot.add(x):
                return None
        return slot.finish()

    def __call__(self, cuts, out: np.ndarray):
        from lhotse import CutSet, MonoCut, validate
        from lhotse.audio.utils import suppress_audio_loading_errors

        from .ring_loader import SlotWriter

        validate(CutSet.from_cuts(cuts))
        assert all(c.has_recording for c in cuts)
        cache = self.__dict__.setdefault("_rec_cache", {})
        t = self.template
        if self.pcm16:
            packed = self._pcm16_batch(cuts, out, MonoCut)
            if packed is not None:
                kept = list(range(len(cuts)))
                meta = {"kept": kept, "pcm16": True, "frags": None if t is None else [manifest_fragments(c, t, self.frame_shift, cache, MonoCut) for c in cuts]}
                used, meta["offs"], meta["lens"] = packed
                return used, meta
        kept, slot, loose = [], SlotWriter(out), None
        for i, c in enumerate(cuts):
            with suppress_audio_loading_errors():
                a = c.load_audio()
                kept.append(i)
                # every cut goes into the slot the moment it is loaded (one decoded array alive at a time); the first one that is not mono
                # float32, or does not fit, turns the batch into a list of arrays
                if loose is None and not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[0] == 1 and slot.add(a[0])):
                    loose = [x.reshape(1, -1) for x in slot.arrays()]
                if loose is not None:
                    loose.append(a)
        meta = {"kept": kept, "frags": None if t is None else [manifest_fragments(cuts[i], t, self.frame_shift, cache, MonoCut) for i in kept]}
        if loose is not None:
            meta["audio"] = loose
            return 0, meta
        used, meta["offs"], meta["lens"] = slot.finish()
        return used, meta


def _shm_free_bytes() -> Optional[int]:
    try:
        st = os.statvfs("/dev/shm")
        return int(st.f_bavail) * int(st.f_frsize)
    except OSError:
        return None


def write_lines(manifest, blob: bytes) -> None:
    """Put pre-serialised JSONL lines behind what a SequentialJsonlWriter has written so far (its `file` is a text-mode handle over a
    GzipFile or a plain file: the bytes go to the layer underneath; zlib and the file write release the GIL)."""
    manifest._maybe_open() if hasattr(manifest, "_maybe_open") else None
    f = manifest.file if hasattr(manifest, "file") else manifest
    f.flush()
    raw = getattr(f, "buffer", None)
    if raw is not None:
        raw.write(blob)
    else:
        f.write(blob if "b" in getattr(f, "mode", "") else blob.decode("utf-8"))
    f.flush()  # one flush per batch



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


class _HostFeatures:
    """A feature matrix that is on the host already, with the interface of extractors.PendingFeatures."""

    __slots__ = ("_array", "frames")

    def __init__(self, array: np.ndarray, frames):
        self._array, self.frames = array, frames

    @property
    def shape(self):
        return self._array.shape

    def wait(self) -> np.ndarray:
        return self._array

    def release(self) -> None:
        self._array = None


def _pin_ring(ring, extractor) -> None:
    """Once the extractor has its plan on a GPU: have the ring's slots page-locked for that GPU as they come into use, so that the
    library's host pipeline uploads straight out of them (``RingLoader.pin_for``; ``HIPFEAT_RING_PIN=0`` keeps the staging copy)."""
    if ring._pin is not None or os.environ.get("HIPFEAT_RING_PIN", "1") == "0":
        return
    plan = getattr(extractor, "_plan", None) or getattr(extractor, "plan", None)
    if getattr(plan, "handle", None) and getattr(plan, "device", None) is not None and plan.device.type == "cuda":
        ring.pin_for(plan.lib, plan.device.index or 0)


class _SlotPending:
    """A pending feature matrix whose input lives in a slot of the ring loader: the slot goes back to the ring when the result is there
    (the library packs a batch on its pipeline thread -- a finished result is the proof that the caller's buffers are no longer read)."""

    __slots__ = ("_pending", "_batch")

    def __init__(self, pending, batch):
        self._pending, self._batch = pending, batch

    @property
    def shape(self):
        return self._pending.shape

    def wait(self) -> np.ndarray:
        try:
            return self._pending.wait()
        finally:
            self._batch.release()

    def release(self) -> None:
        self._batch.release()
        self._pending.release()


def _batch_features_pending(extractor, waves, sampling_rate: int, lengths, half: bool = False):
    """-> (pending packed ``(sum T_b, F)`` host matrix, per-cut frame counts).  Host waveforms in front of a Hip* extractor on a GPU go
    through the library's asynchronous host pipeline (``submit_host_items``): the call returns once the batch is packed and enqueued,
    ``pending.wait()`` (on the save thread) gives the matrix, ``pending.release()`` its buffer back.  Everything else is computed here
    and wrapped."""
    if lengths is None and hasattr(extractor, "submit_host_items") and getattr(extractor.plan, "handle", None) and extractor.plan.device.type == "cuda" \
            and not getattr(extractor.config, "dither", 0.0) \
            and all(not isinstance(w, torch.Tensor) or w.device.type == "cpu" for w in waves) and len(waves) > 0:
        pending = extractor.submit_host_items(waves, sampling_rate, half=half)
        return pending, [int(t) for t in pending.frames]
    host, frames = _batch_features_on_host(extractor, waves, sampling_rate, lengths, half=half)
    return _HostFeatures(host, frames), frames


def _packed_features_pending(extractor, flat: np.ndarray, offs: np.ndarray, lens: np.ndarray, sampling_rate: int, half: bool = False):
    """``_batch_features_pending`` for a batch that lies in ONE host buffer (a slot of the ring loader): handed to the library's host
    pipeline by base pointer + offsets when that route is open, as 1-D views otherwise."""
    if hasattr(extractor, "submit_host_packed") and getattr(extractor.plan, "handle", None) and extractor.plan.device.type == "cuda" \
            and not getattr(extractor.config, "dither", 0.0) and len(lens) > 0:
        pending = extractor.submit_host_packed(flat, offs, lens, sampling_rate, half=half)
        return pending, [int(t) for t in pending.frames]
    waves = [flat[o : o + n] for o, n in zip(offs.tolist(), lens.tolist())]
    return _batch_features_pending(extractor, waves, sampling_rate, None, half=half)


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
    archive_stripes: int = 1,
    loader_start_method: Optional[str] = None,
    worker_init_fn: Optional[Callable] = None,
    loader: Optional[str] = None,
    wav_pcm16: bool = False,
):
    """``CutSet.compute_and_store_features_batch`` with the bulk save path of this module (same arguments; ``storage_type``
    defaults to ``HipArchiveWriter``).  Returns the CutSet with the ``Features`` manifests attached.

    ``loader`` (round 6): ``"ring"`` = the worker processes load every batch's audio straight into a slot of ONE shared-memory ring
    (``lhotse_amd.ring_loader``; what travels per batch is a few hundred bytes + the manifest-line halves, and the cut objects never
    come back: the main process kept them), ``"dataloader"`` = ``torch.utils.data.DataLoader`` over lhotse's waveform dataset as
    lhotse's own driver uses it (lhotse/cut/set.py:2302-2304).  ``None`` = the ring wherever it applies (``collate=False``, no
    ``augment_fn``, ``num_workers`` > 0, enough room in /dev/shm; any storage type -- with the ``hip_archive`` storages and a manifest
    path the workers also serialise the manifest-line halves), the DataLoader otherwise.  Same
    ``wav_pcm16`` (with the ring): batches whose cuts are ALL plain mono 16-bit PCM ``.wav`` segments (``pcm16_wav_segment``) are read as
    int16 straight into the slot and converted on the device (x / 32768, exact: what the audio backend does on the host) -- the same
    features; any other batch takes lhotse's ``load_audio``.  Off by default: it bypasses a custom audio backend for those files.
    Same batches, same manifests, same archive bytes either way; measured with real WAV decoding in the workers the ring moves 13.7 k / 20 k
    cuts/s (float32 / int16 corpus) where the DataLoader moves 5-6 k / 8-10 k (profiles/r06_ring_loader_ab.txt).

    ``loader_start_method`` (round 6): how the DataLoader's worker processes are started -- ``None`` = ``"fork"`` (what lhotse's driver
    gets from torch's default) unless this process ALREADY holds a live HIP context, in which case ``"forkserver"``: workers forked off a
    process with a live context slow every device round trip of that process by ~30 ms while they live (``_lib.hip_live``; measured 1.35 k
    against 5.5-6.7 k cuts/s).  Fork-server workers import lhotse afresh: process-global settings of the caller (a custom audio backend,
    ``set_caching_enabled``) have to be re-established in ``worker_init_fn(worker_id)``.

    With the ``hip_archive`` / ``hip_archive_f16`` storages and a ``manifest_path`` the per-batch host work runs in libhipfeat
    (``NativeArchive``): the loader's worker processes serialise every cut's manifest line up to the storage fields
    (``manifest_fragments``), the save threads append the batch to the archive and splice the keys in -- no per-cut Python in this
    process.  ``archive_stripes`` > 1 stripes the archive over that many files (``feats.hfa``, ``feats.1.hfa``, ...), one writer
    thread each (a page-cache file takes one writer's copy rate; the readers follow each cut's ``storage_path``)."""
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
    if getattr(storage_type, "np_dtype", "<f4") == "<f2" and not getattr(extractor, "log_domain", True):
        raise ValueError(f"storage '{storage_type.name}' keeps binary16 rows, which cannot hold the linear-domain output of '{extractor.name}' "
                         "(overflow above 65504, flush to zero below 6e-8): use 'hip_archive'")
    frame_shift = extractor.frame_shift
    loader_kw: Dict = {}
    if num_workers > 0:
        from . import _lib as _hiplib

        method = loader_start_method
        if method is None and _hiplib.hip_live():
            method = "forkserver"
            warnings.warn("lhotse_amd.compute_and_store_features_batch: this process already holds a live HIP context; the DataLoader's workers are started "
                          "by a fork server (workers forked off such a process slow its device round trips by ~30 ms each while they live).  They import lhotse "
                          "afresh (the calling script needs the usual `if __name__ == '__main__':` guard): pass worker_init_fn to re-establish process-global "
                          "settings, or loader_start_method='fork' to keep torch's default.",
                          RuntimeWarning, stacklevel=2)
        if method is not None and method != "fork":
            loader_kw["multiprocessing_context"] = method
            if method == "forkserver":
                import multiprocessing as mp

                # the workers are forked off the SERVER: with the heavy imports done there once, a worker starts in milliseconds -- without,
                # every worker imports torch + lhotse on its own (measured: 28 s to the first batch with 32 workers)
                mp.set_forkserver_preload(["torch", "torch.utils.data", "numpy", "lhotse", "lhotse.dataset", "lhotse_amd.storage"])
        if worker_init_fn is not None:
            loader_kw["worker_init_fn"] = worker_init_fn
    manifest = CutSet.open_writer(manifest_path, overwrite=overwrite)
    # rank / world pinned: under torchrun lhotse's samplers would otherwise split (and pad with duplicated cuts) what they are given once
    # more (lhotse/dataset/sampling/base.py:152-163); sharding over GPUs is explicit here (lhotse_amd.compute_and_store_features_sharded)
    sampler = SimpleCutSampler(cuts, max_duration=batch_duration, world_size=1, rank=0)
    sampler.filter(lambda cut: cut.id not in manifest.ignore_ids)  # resume: skip what the manifest already holds
    # (exactly the registered archive classes: a subclass that overrides write / write_packed is served through its own methods)
    native = manifest_path is not None and storage_type in (HipArchiveWriter, HipArchiveF16Writer)
    rec_cache: Dict[str, Tuple[object, Dict]] = {}

    def check_frames(batch_cuts, frames):
        """The frame-count contract of validate_features (qa.py:286-301), for the whole batch (per-cut path)."""
        for c, t in zip(batch_cuts, frames):
            if isinstance(c, PaddingCut):
                continue
            hop = round(frame_shift * c.sampling_rate, ndigits=12)  # qa.py:286-291: the hop must be a whole number of samples
            if not float(hop).is_integer():
                raise AssertionError(f"cut {c.id}: frame_shift {frame_shift} s is {hop} samples at {c.sampling_rate} Hz (not an integer)")
            if compute_num_frames(c.duration, frame_shift, c.sampling_rate) != t:
                raise AssertionError(f"cut {c.id}: {t} frames for {c.duration} s at frame_shift {frame_shift} (lhotse expects "
                                     f"{compute_num_frames(c.duration, frame_shift, c.sampling_rate)})")

    def write_cut_objects(batch_cuts, frames, feature_dicts, num_features: int) -> None:
        """One manifest per cut through Python objects: template dicts for plain MonoCuts, lhotse's own objects for the rest."""
        for i, c in enumerate(batch_cuts):
            if isinstance(c, PaddingCut):
                manifest.write(fastcopy(c, num_frames=frames[i], num_features=num_features, frame_shift=frame_shift))
                continue
            fd = feature_dicts[i]
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

    def run(writer, loader, save, finish, half: bool, template_of, extract=None):
        def extract_loaded(batch):
            batch_cuts, waves = batch["cuts"], unpack_batch_audio(batch)
            lens = batch["audio_lens"] if collate else None
            if len(batch_cuts) == 0:
                return None
            sr = batch_cuts[0].sampling_rate
            assert all(c.sampling_rate == sr for c in batch_cuts)
            if augment_fn is not None:
                waves = [augment_fn(w, c.sampling_rate) for c, w in zip(batch_cuts, waves)]
            pending, frames = _batch_features_pending(extractor, waves, sr, lens, half=half)
            return writer, list(batch_cuts), pending, frames, template_of(pending, sr), batch.get("hipfeat_fragments")

        pump_batches(loader, extract or extract_loaded, save, finish=finish)

    if loader not in (None, "ring", "dataloader"):
        raise ValueError(f"loader={loader!r}: expected None, 'ring' or 'dataloader'")
    ring_applies = not collate and augment_fn is None and num_workers > 0
    if loader == "ring" and not ring_applies:
        raise ValueError("loader='ring' serves collate=False without an augment_fn and with num_workers > 0")
    use_ring = ring_applies and loader != "dataloader"

    def open_ring(template_base, first):
        """The shared-memory ring loader for this run, or None (not asked for / no room in /dev/shm: the DataLoader then)."""
        if not use_ring:
            return None
        from .ring_loader import RingLoader

        sr0 = getattr(first, "sampling_rate", None) or 16000
        slot_bytes = int(batch_duration * sr0 * 4 * 1.01) + 65536  # a batch of `batch_duration` seconds, float32, every cut on a 16-byte boundary
        holds = _SAVE_BACKLOG + 4  # slots the extractor / save threads keep at any time
        per_worker = 2 + -(-holds // num_workers)  # (every worker owns its slots: lhotse_amd/ring_loader.py)
        room = _shm_free_bytes()
        if room is not None and per_worker * num_workers * slot_bytes > 0.8 * room:
            per_worker = min(per_worker, int(0.8 * room // (slot_bytes * num_workers)))
        if per_worker < 2:
            want = 2 * num_workers * slot_bytes >> 20
            if loader == "ring":
                raise OSError(f"loader='ring' needs {want} MiB of /dev/shm ({room >> 20} MiB are free): lower batch_duration / num_workers")
            warnings.warn(f"lhotse_amd.compute_and_store_features_batch: /dev/shm has {room >> 20} MiB free, the ring loader wants {want} MiB; "
                          "using the DataLoader", RuntimeWarning, stacklevel=3)
            return None
        return RingLoader(LoadCutsIntoSlot(template_base, frame_shift, pcm16=wav_pcm16), num_workers, slot_bytes, per_worker * num_workers, start_method=loader_kw.get("multiprocessing_context"),
                          worker_init_fn=worker_init_fn, preload=["lhotse", "lhotse.dataset", "lhotse_amd.storage"])

    def ring_extract(ring, writer, half: bool, template_of):
        """`extract` of pump_batches for batches that arrive in slots of the ring."""

        def extract_ring(rb):
            meta = rb.meta
            batch_cuts = [rb.spec[i] for i in meta["kept"]]
            if len(batch_cuts) == 0:
                rb.release()
                return None
            sr = batch_cuts[0].sampling_rate
            assert all(c.sampling_rate == sr for c in batch_cuts)
            if "audio" in meta:  # (a batch that did not fit a slot / is not mono float32: it came by pickle)
                pending, frames = _batch_features_pending(extractor, meta["audio"], sr, None, half=half)
            else:
                if meta.get("pcm16"):
                    TEMPLATE_STATS["pcm16_batches"] = TEMPLATE_STATS.get("pcm16_batches", 0) + 1
                pending, frames = _packed_features_pending(extractor, rb.data.view(np.int16 if meta.get("pcm16") else np.float32), meta["offs"], meta["lens"], sr, half=half)
            _pin_ring(ring, extractor)
            return writer, batch_cuts, _SlotPending(pending, rb), frames, template_of(pending, sr), meta["frags"]

        return extract_ring

    def close_ring(ring) -> None:
        if ring is not None:
            pipe = getattr(extractor, "__dict__", {}).get("_native_pipeline")
            if pipe is not None:  # (after an exception batches may still be queued that read the ring's slots: not unmapped under them)
                pipe.drain()
            ring.close()

    if native:
        # ---- the native path: archive appends and manifest lines in libhipfeat, fragments from the loader's workers -------------------
        np_dtype = getattr(storage_type, "np_dtype", "<f4")
        with manifest, NativeArchive(storage_path, mode="w" if overwrite else "a", np_dtype=np_dtype, stripes=archive_stripes, name=storage_type.name) as archive:
            first = next(iter(cuts), None)
            base = None
            if first is not None and not isinstance(first, PaddingCut):
                base = {"type": extractor.name, "num_features": int(extractor.feature_dim(first.sampling_rate)), "frame_shift": frame_shift,
                        "sampling_rate": first.sampling_rate, "storage_type": archive.name, "storage_path": archive.storage_path}

            # lhotse's waveform dataset + the halves of every cut's manifest line, made where the cut is loaded (the DataLoader's worker
            # processes when num_workers > 0); a module-level class: picklable, so the `spawn` start method works too
            # (packed in the worker -- one shared-memory segment per batch instead of one per cut -- unless an augment_fn wants the per-cut arrays)
            ring = open_ring(base, first)
            if ring is None:
                batches = DataLoader(_fragmenting_dataset_class()(collate, base, frame_shift, pack=augment_fn is None), batch_size=None, sampler=sampler,
                                     num_workers=num_workers, **loader_kw)
                extract_ring = None
            else:
                batches = ring.batches(list(b) for b in sampler)
                extract_ring = ring_extract(ring, archive, np_dtype == "<f2", lambda pending, sr: template_of(pending, sr))

            def save(archive, batch_cuts, pending, frames: List[int], template: Dict, frags):
                frames = np.ascontiguousarray(frames, dtype=np.int64)
                host = pending.wait()  # (the batch's download may still be in flight: the extractor only enqueued it)
                with np.errstate(over="ignore"):
                    host = np.ascontiguousarray(host, dtype=np_dtype)
                stored = [i for i, c in enumerate(batch_cuts) if not isinstance(c, PaddingCut)]
                if len(stored) != len(batch_cuts):  # padding cuts store nothing: their rows are cut out of the batch matrix
                    bounds = np.concatenate([[0], np.cumsum(frames)])
                    host = np.ascontiguousarray(np.concatenate([host[int(bounds[i]) : int(bounds[i + 1])] for i in stored], axis=0)) if stored else host[:0]
                file_of = np.zeros(len(batch_cuts), dtype=np.int32)
                byte_off = np.zeros(len(batch_cuts), dtype=np.int64)
                f2, b2 = archive.append(host, frames[stored])
                num_features = int(host.shape[1])
                del host
                pending.release()  # the page-locked result goes back to the pipeline
                file_of[stored], byte_off[stored] = f2, b2
                return archive, batch_cuts, frames, file_of, byte_off, num_features, template, frags

            def finish(archive, batch_cuts, frames, file_of, byte_off, num_features: int, template: Dict, frags) -> None:
                spliced = frags is not None and all(f is not None for f in frags) and base is not None and num_features == base["num_features"]
                if spliced:
                    blob = archive.lines([f[0] for f in frags], [f[1] for f in frags], frames, np.fromiter((f[2] for f in frags), dtype=np.int64, count=len(frags)),
                                         file_of, byte_off, num_features)
                    write_lines(manifest, blob)
                    TEMPLATE_STATS["template"] += len(frags)
                    TEMPLATE_STATS["native"] = TEMPLATE_STATS.get("native", 0) + len(frags)
                    return
                check_frames(batch_cuts, frames)
                where = archive.keys(frames, file_of, byte_off, num_features)
                dicts = []
                for i, c in enumerate(batch_cuts):
                    t = dict(template)
                    t["storage_path"] = where[i][0]
                    dicts.append(None if isinstance(c, PaddingCut) else _features_dict(t, c, int(frames[i]), where[i][1]))
                write_cut_objects(batch_cuts, [int(t) for t in frames], dicts, num_features)

            def template_of(host, sr):
                return {"type": extractor.name, "num_features": int(host.shape[1]), "frame_shift": frame_shift, "sampling_rate": sr,
                        "storage_type": archive.name, "storage_path": archive.storage_path}

            try:
                run(archive, batches, save, finish, np_dtype == "<f2", template_of, extract=extract_ring)
            finally:
                close_ring(ring)
        return manifest.open_manifest()

    # ---- any other registered FeaturesWriter: per-cut write() calls, manifests through Python objects ------------------------------
    ring = open_ring(None, next(iter(cuts), None))  # (no line halves here: the manifests of this path go through Python objects)
    if ring is None:
        batches = DataLoader(_fragmenting_dataset_class()(collate, None, frame_shift, pack=augment_fn is None), batch_size=None, sampler=sampler, num_workers=num_workers,
                             **loader_kw)
    else:
        batches = ring.batches(list(b) for b in sampler)

    def save(writer, batch_cuts, pending, frames: List[int], template: Dict, frags):
        check_frames(batch_cuts, frames)
        host = pending.wait()
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
        num_features = int(host.shape[1])
        del host
        pending.release()
        return batch_cuts, frames, keys, num_features, template  # -> write_manifests, on the second background thread

    def write_manifests(batch_cuts, frames: List[int], keys: List[str], num_features: int, template: Dict) -> None:
        dicts = [None if isinstance(c, PaddingCut) else _features_dict(template, c, frames[i], keys[i]) for i, c in enumerate(batch_cuts)]
        write_cut_objects(batch_cuts, frames, dicts, num_features)

    with manifest, storage_type(storage_path, mode="w" if overwrite else "a") as writer:
        state = {"template": None}

        def template_of(host, sr):
            if state["template"] is None:
                state["template"] = {"type": extractor.name, "num_features": int(host.shape[1]), "frame_shift": frame_shift, "sampling_rate": sr,
                                     "storage_type": writer.name, "storage_path": str(writer.storage_path)}
            return state["template"]

        half = getattr(writer, "np_dtype", "<f4") == "<f2"
        try:
            run(writer, batches, save, write_manifests, half, template_of, extract=None if ring is None else ring_extract(ring, writer, half, template_of))
        finally:
            close_ring(ring)
    return manifest.open_manifest()
