"""
Drop-in lhotse feature extractors backed by the MI355X HIP library.

    HipFbank            <->  lhotse.features.kaldi.extractors.Fbank            ("kaldi-fbank")
    HipMfcc             <->  ...Mfcc                                           ("kaldi-mfcc")
    HipSpectrogram      <->  ...Spectrogram                                    ("kaldi-spectrogram")
    HipLogSpectrogram   <->  ...LogSpectrogram                                 ("kaldi-log-spectrogram")

Same config fields (lhotse/features/kaldi/extractors.py:23-63, 155-197, 265-293, 375-403),
same ``extract`` / ``extract_batch`` / ``frame_shift`` / ``feature_dim`` / ``device`` /
``mix`` / ``compute_energy`` / ``scale`` / ``to_dict`` / ``from_dict`` surface
(lhotse/features/base.py:37-365), same input/return conventions (numpy in -> numpy out,
torch in -> torch out, list / stacked / bare item; extractors.py:485-554).  What differs:

  * the arithmetic runs in hand-written HIP kernels through the C ABI of include/hipfeat.h
    (one fused launch per batch instead of ~12 full-tensor torch ops);
  * ``edge_rule`` selects how the right edge of SHORTER batch items is treated:
    "reflect" (default) frames every item on its own -- identical to ``extract()`` and to
    kaldifeat -- while "batch_zero_pad" reproduces ``_extract_batch``'s zero-padded batch
    (SURVEY.md section 8a, quirk Q1);
  * there is no CPU fallback: without a GPU / the built library, calls raise.

Objects are cheap, picklable (config only; the device plan is created lazily on first use)
and default-constructible without a GPU, as lhotse requires of registered extractors
(lhotse/features/base.py:381-388).
"""
from __future__ import annotations

import os
import threading
import warnings
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, constants
from .compat import EPSILON, LOG_EPSILON, FeatureExtractor, Seconds, asdict_nonull, compute_num_frames_from_samples, register_extractor

KIND_SPECTROGRAM, KIND_LOG_SPECTROGRAM, KIND_FBANK, KIND_MFCC, KIND_WHISPER, KIND_LIBROSA_FBANK = 0, 1, 2, 3, 4, 5
EDGE_RULES = ("reflect", "batch_zero_pad")

ArrayLike = Union[np.ndarray, torch.Tensor]


# --------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------
def _check_common(cfg) -> None:
    if cfg.snip_edges:
        warnings.warn(
            "`snip_edges` is set to True, which may cause issues in duration to num-frames conversion in Lhotse."
        )
    if cfg.edge_rule not in EDGE_RULES:
        raise ValueError(f"edge_rule must be one of {EDGE_RULES}, got {cfg.edge_rule!r}")


@dataclass
class HipSpectrogramConfig:
    sampling_rate: int = 16000
    frame_length: Seconds = 0.025
    frame_shift: Seconds = 0.01
    round_to_power_of_two: bool = True
    remove_dc_offset: bool = True
    preemph_coeff: float = 0.97
    window_type: str = "povey"
    dither: float = 0.0
    snip_edges: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    use_energy: bool = False
    use_fft_mag: bool = False
    device: str = "cuda"
    edge_rule: str = "reflect"

    def __post_init__(self):
        _check_common(self)

    def to_dict(self) -> Dict[str, Any]:
        return asdict_nonull(self)

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipSpectrogramConfig":
        return HipSpectrogramConfig(**data)


@dataclass
class HipLogSpectrogramConfig(HipSpectrogramConfig):
    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipLogSpectrogramConfig":
        return HipLogSpectrogramConfig(**data)


@dataclass
class HipFbankConfig:
    sampling_rate: int = 16000
    frame_length: Seconds = 0.025
    frame_shift: Seconds = 0.01
    round_to_power_of_two: bool = True
    remove_dc_offset: bool = True
    preemph_coeff: float = 0.97
    window_type: str = "povey"
    dither: float = 0.0
    snip_edges: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    use_energy: bool = False
    use_fft_mag: bool = False
    low_freq: float = 20.0
    high_freq: float = -400.0
    num_filters: int = 80
    num_mel_bins: Optional[int] = None  # do not use (alias kept for FbankConfig compatibility)
    norm_filters: bool = False
    torchaudio_compatible_mel_scale: bool = True
    device: str = "cuda"
    edge_rule: str = "reflect"

    def __post_init__(self):
        if self.num_mel_bins is not None:
            self.num_filters = self.num_mel_bins
            self.num_mel_bins = None
        _check_common(self)

    def to_dict(self) -> Dict[str, Any]:
        return asdict_nonull(self)

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipFbankConfig":
        return HipFbankConfig(**data)


@dataclass
class HipMfccConfig:
    sampling_rate: int = 16000
    frame_length: Seconds = 0.025
    frame_shift: Seconds = 0.01
    round_to_power_of_two: bool = True
    remove_dc_offset: bool = True
    preemph_coeff: float = 0.97
    window_type: str = "povey"
    dither: float = 0.0
    snip_edges: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    use_energy: bool = False
    use_fft_mag: bool = False
    low_freq: float = 20.0
    high_freq: float = -400.0
    num_filters: int = 23
    torchaudio_compatible_mel_scale: bool = True
    num_mel_bins: Optional[int] = None  # do not use
    norm_filters: bool = False
    num_ceps: int = 13
    cepstral_lifter: int = 22
    device: str = "cuda"
    edge_rule: str = "reflect"

    def __post_init__(self):
        if self.num_mel_bins is not None:
            self.num_filters = self.num_mel_bins
            self.num_mel_bins = None
        _check_common(self)

    def to_dict(self) -> Dict[str, Any]:
        return asdict_nonull(self)

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipMfccConfig":
        return HipMfccConfig(**data)


# --------------------------------------------------------------------------------------
# device plan (lazy, per extractor instance)
# --------------------------------------------------------------------------------------
class _Plan:
    """Owns one ``hipfeat_plan`` (constants resident in HBM + kernel selection)."""

    def __init__(self, cfg, kind: int, device: torch.device, mel_floor: float = constants.MEL_FLOOR):
        self.lib = _lib.load()
        self.handle = 0
        if device.type != "cuda":
            raise _lib.HipFeatError(1, f"Hip* extractors run on an AMD GPU ('cuda[:i]' device), got device={device}")
        if not torch.cuda.is_available():
            raise _lib.HipFeatError(2, "no HIP device is visible (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)
        if kind == KIND_LIBROSA_FBANK:  # sizes in samples; the window is the zero-padded STFT window (librosa_fbank.py:111-118)
            n = fft = int(cfg.fft_size)
            shift = int(cfg.hop_size)
            window = constants.make_stft_window(cfg.window, int(cfg.win_length or fft), fft)
        else:
            n, shift, fft = constants.frame_sizes(cfg.sampling_rate, cfg.frame_length, cfg.frame_shift, cfg.round_to_power_of_two)
            window = constants.make_window(n, cfg.window_type, getattr(cfg, "blackman_coeff", 0.42))
        self.n, self.shift, self.fft = n, shift, fft
        mel = dct = lifter = None
        num_filters = num_ceps = 0
        apply_lifter = 0
        if kind in (KIND_FBANK, KIND_MFCC, KIND_WHISPER, KIND_LIBROSA_FBANK):
            num_filters = int(cfg.num_filters)
            if kind == KIND_LIBROSA_FBANK:
                mel = constants.make_slaney_mel(num_filters, fft, cfg.sampling_rate, cfg.fmin or 0.0, cfg.fmax)
            elif kind == KIND_WHISPER:
                mel = constants.make_slaney_mel(num_filters, fft, cfg.sampling_rate)
            elif cfg.torchaudio_compatible_mel_scale:
                mel = constants.make_kaldi_mel(num_filters, fft, cfg.sampling_rate, cfg.low_freq, cfg.high_freq)
            else:
                mel = constants.make_htk_mel(num_filters, fft, cfg.sampling_rate, cfg.low_freq, cfg.high_freq, cfg.norm_filters)
        if kind == KIND_MFCC:
            num_ceps = int(cfg.num_ceps)
            dct = constants.make_dct(num_ceps, num_filters)
            apply_lifter = 1 if cfg.cepstral_lifter > 0 else 0
            lifter = constants.make_lifter(num_ceps, cfg.cepstral_lifter)
        c = np.zeros((), dtype=_lib.CONFIG_DTYPE)
        c["struct_size"] = _lib.CONFIG_DTYPE.itemsize
        c["kind"] = kind
        c["frame_length"], c["frame_shift"], c["fft_length"] = n, shift, fft
        c["num_filters"], c["num_ceps"] = num_filters, num_ceps
        c["snip_edges"] = int(cfg.snip_edges)
        c["remove_dc_offset"] = int(cfg.remove_dc_offset)
        c["use_energy"] = int(cfg.use_energy)
        c["raw_energy"] = int(cfg.raw_energy)
        c["use_fft_mag"] = int(cfg.use_fft_mag)
        c["apply_lifter"] = apply_lifter
        c["preemph_coeff"] = cfg.preemph_coeff
        c["energy_floor"] = cfg.energy_floor
        c["mel_floor"] = mel_floor
        c["log_offset"] = constants.LOG_SPEC_OFFSET
        # dither is applied by the host mirror to the packed waveform with the device RNG (torch.randn, exactly what the
        # reference does, layers.py:189-193); the library itself takes the (dithered) samples
        self.dither = float(cfg.dither)
        c["dither"] = 0.0
        # rows an item keeps of a zero-padded batch row: compute_num_frames_from_samples counts with round(), the framing
        # itself with floor() (lhotse/utils.py:424-434 vs layers.py:116)
        self.batch_hop = int(round(cfg.frame_shift * cfg.sampling_rate)) if hasattr(cfg, "frame_shift") and hasattr(cfg, "sampling_rate") else shift
        c["batch_hop"] = self.batch_hop
        cbuf = np.ascontiguousarray(c).reshape(1)
        out = np.zeros(1, dtype=np.uint64)
        self.lib.check(
            "hipfeat_plan_create",
            _lib.addr(cbuf),
            _lib.addr(window),
            _lib.addr(mel),
            _lib.addr(dct),
            _lib.addr(lifter),
            int(self.device.index),
            _lib.addr(out),
        )
        self.handle = int(out[0])
        _lib.note_plan_created()  # (the fork hazard: _lib.hip_live)
        self._pair_ok = os.environ.get("HIPFEAT_COLLATED_NO_PAIR") is None
        self.feature_dim = int(self.lib.raw("hipfeat_plan_feature_dim", self.handle))
        self.kernel_name = self.lib.string("hipfeat_plan_kernel_name", self.handle)
        self.snip_edges = int(cfg.snip_edges)

    def _dithered(self, wave: torch.Tensor) -> torch.Tensor:
        """x + dither * N(0, 1) per sample, drawn on the device (Wav2Win.forward, layers.py:189-193).  Never in place:
        the buffer may be the caller's."""
        if self.dither == 0.0:
            return wave
        with torch.cuda.device(self.device):
            return torch.randn(wave.shape, device=self.device).mul_(self.dither).add_(wave)

    def num_frames(self, num_samples: int) -> int:
        return int(self.lib.raw("hipfeat_num_frames", int(num_samples), self.n, self.shift, self.snip_edges))

    def num_frames_many(self, num_samples: np.ndarray) -> np.ndarray:
        """Vectorised ``hipfeat_num_frames`` (same integer formula; tests/test_abi.py compares the two)."""
        s = _lib.i64(num_samples)
        if self.snip_edges:
            return np.where(s < self.n, 0, 1 + (s - self.n) // self.shift).astype(np.int64)
        return (s + self.shift // 2) // self.shift

    def frame_counts(self, lengths: np.ndarray, padded: Optional[np.ndarray]) -> np.ndarray:
        """Rows every item of a batch yields (a1 of SURVEY 8a; with `padded` the rows an item keeps of a zero-padded batch row)."""
        lengths = _lib.i64(lengths)
        if padded is None:
            return self.num_frames_many(lengths)
        return np.minimum((lengths + self.batch_hop // 2) // self.batch_hop, self.num_frames_many(_lib.i64(padded)))

    def run(self, wave: torch.Tensor, offsets: np.ndarray, lengths: np.ndarray, padded: Optional[np.ndarray]) -> Tuple[torch.Tensor, np.ndarray]:
        """wave: float32 tensor on self.device holding every cut; returns the packed
        (sum T_b, F) feature matrix (same device, same stream) and the per-cut frame counts."""
        assert wave.dtype == torch.float32 and wave.is_contiguous() and wave.device == self.device
        wave = self._dithered(wave)
        lengths = _lib.i64(lengths)
        offsets = _lib.i64(offsets)
        n, shift, snip = self.n, self.shift, self.snip_edges
        if padded is None:
            frames = self.num_frames_many(lengths)
        else:
            padded = _lib.i64(padded)
            own = (lengths + self.batch_hop // 2) // self.batch_hop
            frames = np.minimum(own, self.num_frames_many(padded))
        total = int(frames.sum())
        with torch.cuda.device(self.device):
            out = torch.empty((total, self.feature_dim), dtype=torch.float32, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.lib.check(
                "hipfeat_extract",
                self.handle,
                wave.data_ptr(),
                _lib.addr(offsets),
                _lib.addr(lengths),
                _lib.addr(padded),
                int(len(lengths)),
                out.data_ptr(),
                None,
                self.feature_dim,
                int(stream),
            )
        return out, frames

    def run_collated(self, wave: torch.Tensor, offsets: np.ndarray, lengths: np.ndarray, padded: Optional[np.ndarray],
                     pad_value: float) -> Tuple[torch.Tensor, np.ndarray]:
        """As ``run`` but into a dense (B, Tmax, F) tensor whose padding rows hold ``pad_value``."""
        assert wave.dtype == torch.float32 and wave.is_contiguous() and wave.device == self.device
        wave = self._dithered(wave)
        lengths, offsets = _lib.i64(lengths), _lib.i64(offsets)
        # Round 4: the launch pair of the on-the-fly mini-batch (hipfeat_minibatch_*, a bank without resamplers) also serves the plain
        # collated extraction -- the padding rows and the descriptor tables travel with the first launch (for up to ~90 cuts in its kernel
        # arguments: no host -> device copy in front of the feature launch), where hipfeat_extract_collated stages the descriptors through
        # pinned memory and fills the padding in a third launch.  Same kernels on the same operands: bit-identical.
        if len(lengths) and self._pair_ok and (padded is None or bool((_lib.i64(padded) == int(lengths.max())).all())) and wave.ndim == 1:
            try:
                out, frames, _, _ = self._pair_bank().extract_collated(self, wave, offsets, lengths, np.full(len(lengths), -1, dtype=np.int32), wave.numel(),
                                                                       pad_value, zero_pad_batch=padded is not None)
                return out, frames
            except _lib.HipFeatError as e:
                if e.status != _lib.ERR_UNSUPPORTED:
                    raise
                self._pair_ok = False  # (Whisper / librosa plans keep the three-launch route)
        n, shift, snip = self.n, self.shift, self.snip_edges
        frames = self.num_frames_many(lengths)
        if padded is not None:
            padded = _lib.i64(padded)
            frames = np.minimum((lengths + self.batch_hop // 2) // self.batch_hop, self.num_frames_many(padded))
        tmax = int(frames.max(initial=0))
        got = np.zeros(len(lengths), dtype=np.int64)
        with torch.cuda.device(self.device):
            out = torch.empty((len(lengths), tmax, self.feature_dim), dtype=torch.float32, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.lib.check("hipfeat_extract_collated", self.handle, wave.data_ptr(), _lib.addr(offsets), _lib.addr(lengths), _lib.addr(padded),
                           int(len(lengths)), out.data_ptr(), tmax, float(pad_value), _lib.addr(got), int(stream))
        assert np.array_equal(got, frames)
        return out, frames

    def _pair_bank(self):
        bank = self.__dict__.get("_bank")
        if bank is None:
            from .augmentation import HipSpeedBank

            bank = self.__dict__["_bank"] = HipSpeedBank([], 16000, self.device)  # (no resamplers: the rate is irrelevant)
        return bank

    def close(self):
        bank = self.__dict__.pop("_bank", None)
        if bank is not None:
            bank.close()
        if self.handle:
            try:
                self.lib.raw("hipfeat_plan_destroy", self.handle)
            finally:
                self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------------------
# pinned host staging (the host<->device edge of extract_batch)
# --------------------------------------------------------------------------------------
_COPY_POOL: Optional[ThreadPoolExecutor] = None
_COPY_PIECE = 1 << 20  # floats per task (4 MiB)
_COPY_THREADS = max(2, min(12, _lib.usable_cpus() // 2))  # (affinity mask and the container's CPU quota, not the host's CPU count)
_PACK_RUN = 1 << 20    # floats per pack-and-upload run of pack_to_device, at least (4 MiB)


def _parallel_copy(dst: np.ndarray, pieces: Sequence[Tuple[int, np.ndarray]]) -> None:
    """dst[o : o + len(src)] = src for every (o, src), spread over a few host threads.  numpy releases the GIL inside the copies; one
    thread moves only ~5-10 GB/s, far below PCIe.  The pieces are dealt into ONE task per thread (runs of whole pieces, long pieces cut
    at 4 MiB): submitting a future per piece costs ~25 us each in the submitting thread, which for a 600 s mini-batch of ~40 cuts was
    more than the copies themselves."""
    global _COPY_POOL
    tasks = []
    total = 0
    for o, src in pieces:
        n = src.shape[0]
        for a in range(0, n, _COPY_PIECE):
            tasks.append((o + a, src[a : a + _COPY_PIECE]))
        total += n
    if len(tasks) <= 2 or total < (1 << 18):
        for o, src in tasks:
            dst[o : o + src.shape[0]] = src
        return
    if _COPY_POOL is None:
        _COPY_POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="hipfeat-copy")
    groups, acc, share = [[]], 0, total / min(_COPY_THREADS, len(tasks))
    for t in tasks:
        if acc >= share and len(groups) < _COPY_THREADS:
            groups.append([])
            acc = 0
        groups[-1].append(t)
        acc += t[1].shape[0]

    def run(group):
        for o, src in group:
            dst[o : o + src.shape[0]] = src

    futures = [_COPY_POOL.submit(run, g) for g in groups[1:]]
    run(groups[0])  # the calling thread takes a share too
    for f in futures:
        f.result()


class _HostStaging:
    """Grow-only pinned buffers reused across calls.

    Allocating pinned memory costs milliseconds and pageable copies run at a fraction of the
    PCIe rate, so host inputs are packed into one of two pinned input buffers (ping-pong, each
    guarded by an event recorded after its H2D copy) and device results come back through one
    pinned output buffer before being copied into the fresh array handed to the caller."""

    def __init__(self):
        self._in = [None, None]
        self._ev = [None, None]
        self._out = None
        self._turn = 0
        # one host thread at a time packs into / fetches through these buffers (extractors may be shared by the
        # reader threads of a data pipeline); held across pack + H2D issue and across D2H + copy-out
        self.lock = threading.RLock()

    def input(self, n: int) -> Tuple[torch.Tensor, int]:
        i = self._turn
        self._turn ^= 1
        if self._ev[i] is not None:
            self._ev[i].synchronize()  # the previous H2D copy out of this buffer has finished
        buf = self._in[i]
        if buf is None or buf.numel() < n:
            buf = self._in[i] = torch.empty(max(n, 1 << 20), dtype=torch.float32, pin_memory=True)
        return buf, i

    def sent(self, i: int, device: torch.device) -> None:
        ev = self._ev[i]
        if ev is None:
            ev = self._ev[i] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))

    def fetch(self, dev_tensor: torch.Tensor) -> torch.Tensor:
        """Device tensor -> fresh CPU tensor: ONE D2H copy straight into page-locked memory that torch's caching host allocator hands
        out (and takes back when the caller drops the result) -- no bounce buffer, no second host copy."""
        fresh = torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype, pin_memory=True)
        fresh.copy_(dev_tensor, non_blocking=True)
        torch.cuda.current_stream(dev_tensor.device).synchronize()
        return fresh


class _HostPipeline:
    """extract_batch on HOST inputs with HOST outputs (what lhotse's batch driver and its save thread see, cut/set.py:2393-2398): the batch
    is cut into a few chunks and every chunk goes H2D -> kernel on one side stream and D2H on a second one, so that the upload of chunk
    n+1, the launch of chunk n and the download of chunk n-1 overlap (PCIe is full duplex; the kernel is ~100x faster than either copy).
    Page-locked inputs are DMA sources as they are; pageable ones are packed chunk by chunk into the ping-pong pinned staging buffers
    while the previous chunk is on the wire.  The result is ONE fresh pinned tensor holding the packed (sum T_b, F) matrix."""

    TARGET_CHUNKS = int(os.environ.get("HIPFEAT_PIPE_CHUNKS", "4"))
    MIN_CHUNK_BYTES = 2 << 20
    MAX_CHUNK_BYTES = 48 << 20

    def __init__(self, device: torch.device):
        self.device = device
        self.s_in = torch.cuda.Stream(device=device)
        self.s_out = torch.cuda.Stream(device=device)
        self.lock = threading.Lock()  # one batch at a time per extractor: the two streams are the pipeline

    @classmethod
    def chunk_bounds(cls, nbytes: np.ndarray) -> List[Tuple[int, int]]:
        """Consecutive item ranges of roughly equal input bytes: about TARGET_CHUNKS per batch, within [MIN, MAX] bytes each."""
        total = int(nbytes.sum())
        target = min(max(total // cls.TARGET_CHUNKS, cls.MIN_CHUNK_BYTES), cls.MAX_CHUNK_BYTES)
        bounds, a, acc = [], 0, 0
        for i, b in enumerate(nbytes.tolist()):
            acc += int(b)
            if acc >= target:
                bounds.append((a, i + 1))
                a, acc = i + 1, 0
        if a < len(nbytes):
            bounds.append((a, len(nbytes)))
        return bounds

    def run(self, plan, bounds: Sequence[Tuple[int, int]], frames: np.ndarray, upload, half: bool = False) -> torch.Tensor:
        """`upload(a, b)` (called with s_in current) puts items a..b-1 on the device and returns what `plan.run` needs for them.
        `half`: the features are converted to binary16 on the device (hipfeat_float_to_half) and come back as a float16 tensor -- for
        storage backends that keep half precision; half the download."""
        rows = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
        host = torch.empty((int(rows[-1]), plan.feature_dim), dtype=torch.float16 if half else torch.float32, pin_memory=True)
        with self.lock, torch.cuda.device(self.device):
            for a, b in bounds:
                with torch.cuda.stream(self.s_in):
                    wave, offs, lens, padded = upload(a, b)
                    out, got = plan.run(wave, offs, lens, padded)
                    if half:
                        out16 = torch.empty(out.shape, dtype=torch.float16, device=out.device)
                        plan.lib.check("hipfeat_float_to_half", out.data_ptr(), out16.data_ptr(), out.numel(), int(self.s_in.cuda_stream))
                        out = out16
                    done = torch.cuda.Event()
                    done.record(self.s_in)
                assert np.array_equal(got, frames[a:b]), "frame counts of the chunk differ from the batch plan"
                with torch.cuda.stream(self.s_out):
                    self.s_out.wait_event(done)
                    host[int(rows[a]) : int(rows[b])].copy_(out, non_blocking=True)
                    out.record_stream(self.s_out)
            self.s_out.synchronize()
        return host


class PendingFeatures:
    """The packed (rows, F) feature matrix of one batch on its way to (or already in) host memory.  ``wait()`` -> the numpy matrix
    (a view of page-locked memory owned by the pipeline when ``ticket`` is set: valid until ``release()``).  The object keeps the
    batch's waveforms alive until the pipeline thread has packed them."""

    __slots__ = ("_pipe", "ticket", "_array", "frames", "_keep")

    def __init__(self, pipe, ticket, array: np.ndarray, frames: np.ndarray, keep=None):
        self._pipe, self.ticket, self._array, self.frames, self._keep = pipe, ticket, array, frames, keep

    @property
    def shape(self):
        return self._array.shape

    def wait(self) -> np.ndarray:
        if self._pipe is not None and self.ticket is not None:
            try:
                self._pipe._wait(self.ticket)
            finally:
                self._keep = None  # (the library reports a batch only after it is done with the caller's buffers, also on failure)
        return self._array

    def release(self) -> None:
        if self._pipe is not None and self.ticket is not None:
            t, self.ticket = self.ticket, None
            self._array = None
            self._pipe._release(t)  # (waits for the pipeline thread to be done with the waveforms first)
            self._keep = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class NativeHostPipeline:
    """``hipfeat_host_pipeline_*`` (include/hipfeat.h): the batch driver's extraction step as one asynchronous library call per batch --
    a persistent pool of host threads packs the cuts into page-locked staging and the chunked H2D / launch / D2H sequence is ENQUEUED,
    not waited for.  Against the Python ``_HostPipeline`` it removes ~50 interpreter-level calls (and their GIL hand-overs) per batch
    and lets batch n + 1's packing overlap batch n's download."""

    def __init__(self, plan, copy_threads: Optional[int] = None):
        self.plan, self.lib = plan, plan.lib
        # copy threads: HIPFEAT_COPY_THREADS, else a quarter of the CPUs this process may USE (affinity mask and the container's CPU quota:
        # _lib.usable_cpus), 2 ... 12 (profiles/r05_copy_threads.txt)
        env = os.environ.get("HIPFEAT_COPY_THREADS")
        threads = copy_threads or (int(env) if env else max(2, min(12, _lib.usable_cpus() // 4)))
        h = np.zeros(1, dtype=np.uint64)
        self.handle = 0
        self.lib.check("hipfeat_host_pipeline_create", plan.handle, int(threads), _lib.addr(h))
        self.handle = int(h[0])
        self.threads = int(threads)
        self._submit = self.lib.fn("hipfeat_host_pipeline_submit")
        self._waitf, self._releasef = self.lib.fn("hipfeat_host_pipeline_wait"), self.lib.fn("hipfeat_host_pipeline_release")
        # Tickets whose result buffers (page-locked memory the LIBRARY owns) are still with a caller.  close() while some are outstanding --
        # extractor.to(), a dropped plan, __del__ -- is DEFERRED until the last of them is released: a save thread that already holds the
        # array of wait() would otherwise read freed memory.  `_after_close` (the plan's close) runs behind the deferred destroy.
        self._state = threading.Lock()
        self._outstanding = set()
        self._closing = False
        self._after_close = None

    def submit(self, items: Sequence[ArrayLike], zero_pad_batch: bool = False, half: bool = False) -> PendingFeatures:
        """1-D HOST waveforms (all float32 or all int16 PCM; numpy arrays or CPU tensors) -> PendingFeatures."""
        if self._closing or not self.handle:
            raise _lib.HipFeatError(_lib.ERR_INVALID, "the host pipeline is closed (extractor moved or plan dropped)")
        B = len(items)
        pcm = _is_pcm16(items[0])
        keep, ptrs, lens = [], np.empty(B, dtype=np.uint64), np.empty(B, dtype=np.int64)
        for i, x in enumerate(items):
            if _is_pcm16(x) != pcm:
                raise TypeError("a batch must be all float32 or all int16 PCM")
            if isinstance(x, torch.Tensor):
                if not x.is_contiguous():
                    x = x.contiguous()
                ptrs[i] = x.data_ptr()
            else:
                x = np.ascontiguousarray(x)
                ptrs[i] = x.ctypes.data
            keep.append(x)  # (alive until the pipeline thread has packed them: PendingFeatures holds the list)
            lens[i] = x.shape[0]
        return self._submit_ptrs(ptrs, lens, pcm, zero_pad_batch, half, keep)

    def submit_packed(self, flat: np.ndarray, offs: np.ndarray, lens: np.ndarray, zero_pad_batch: bool = False, half: bool = False) -> PendingFeatures:
        """The cuts of a batch as they lie in ONE host buffer -- `flat` (1-D float32 or int16 PCM, C-contiguous: a slot of the ring loader),
        cut b = `flat[offs[b] : offs[b] + lens[b]]` -- without building a view per cut (60 views + 60 pointer look-ups per batch were
        half of the submitting thread's time behind the ring loader).  Same result as ``submit`` on the views."""
        if self._closing or not self.handle:
            raise _lib.HipFeatError(_lib.ERR_INVALID, "the host pipeline is closed (extractor moved or plan dropped)")
        if not isinstance(flat, np.ndarray) or flat.ndim != 1 or flat.dtype not in (np.float32, np.int16) or not flat.flags.c_contiguous:
            raise TypeError("submit_packed takes a 1-D C-contiguous float32 / int16 numpy array")
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        if offs.shape != lens.shape or offs.ndim != 1 or len(offs) == 0:
            raise ValueError("offs and lens must be 1-D, of one length, not empty")
        if int(offs.min()) < 0 or int(lens.min()) < 0 or int((offs + lens).max()) > flat.shape[0]:
            raise ValueError("a cut lies outside the buffer")
        ptrs = (offs * flat.itemsize + flat.ctypes.data).astype(np.uint64)
        return self._submit_ptrs(ptrs, lens, flat.dtype == np.int16, zero_pad_batch, half, flat)

    def _submit_ptrs(self, ptrs: np.ndarray, lens: np.ndarray, pcm: bool, zero_pad_batch: bool, half: bool, keep) -> PendingFeatures:
        import ctypes

        B = len(lens)
        frames = np.empty(B, dtype=np.int64)
        res = np.zeros(3, dtype=np.int64)  # h_out pointer, rows, ticket
        a = res.ctypes.data
        st = self._submit(self.handle, ptrs.ctypes.data, lens.ctypes.data, B, 1 if pcm else 0, 1 if zero_pad_batch else 0, 1 if half else 0,
                          frames.ctypes.data, a, a + 8, a + 16)
        if st != 0:
            msg = self.lib.last_error()
            raise (ValueError(msg) if int(st) == _lib.ERR_TOO_SHORT else _lib.HipFeatError(int(st), msg))
        rows, F = int(res[1]), int(self.plan.feature_dim)
        item = 2 if half else 4
        buf = (ctypes.c_char * (rows * F * item)).from_address(int(res[0]))
        arr = np.frombuffer(buf, dtype=np.float16 if half else np.float32).reshape(rows, F)
        with self._state:
            self._outstanding.add(int(res[2]))
        return PendingFeatures(self, int(res[2]), arr, frames, keep)

    def drain(self) -> None:
        """Block until every batch submitted so far has left the caller's waveforms alone and its result is on the host (results stay
        outstanding: their owners still ``release()`` them).  A driver calls this before it unmaps memory batches were submitted out of
        (the ring loader's slots) on a path where not every batch was collected -- an exception half-way through a run."""
        with self._state:
            tickets = sorted(self._outstanding)
        for t in tickets:
            if self.handle:
                self._waitf(self.handle, int(t))  # (a failed batch answers with its status: it was drained by the pipeline thread itself)

    def stats(self) -> Dict[str, float]:
        """The pipeline thread's own clock since creation: seconds busy / packing / waiting for PCIe + device, batches."""
        a = np.zeros(4, dtype=np.int64)
        self.lib.check("hipfeat_host_pipeline_stats", self.handle, _lib.addr(a))
        return {"busy_s": a[0] * 1e-9, "pack_s": a[1] * 1e-9, "device_backpressure_s": a[2] * 1e-9, "batches": int(a[3])}

    def _wait(self, ticket: int) -> None:
        if not self.handle:  # the extractor was moved / its plan dropped while a batch was outstanding: the result buffers went with the pipeline
            raise _lib.HipFeatError(_lib.ERR_INVALID, "the host pipeline of this batch was closed (extractor moved or plan dropped) before the batch was collected")
        st = self._waitf(self.handle, int(ticket))
        if st != 0:
            raise _lib.HipFeatError(int(st), self.lib.last_error())

    def _release(self, ticket: int) -> None:
        if self.handle:
            self._releasef(self.handle, int(ticket))
        with self._state:
            self._outstanding.discard(int(ticket))
            last = self._closing and not self._outstanding
        if last:
            self._destroy()

    def _destroy(self) -> None:
        with self._state:
            h, self.handle = self.handle, 0
            after, self._after_close = self._after_close, None
        if h:
            self.lib.raw("hipfeat_host_pipeline_destroy", h)
        if after is not None:
            after()

    def close(self, after=None) -> bool:
        """Destroy the pipeline -- now, or (tickets outstanding) when the last of them is released.  `after` runs behind the destroy
        (the owner's plan.close: the pipeline borrows the plan).  -> whether it happened now."""
        with self._state:
            if after is not None:
                prev = self._after_close
                self._after_close = after if prev is None else (lambda: (prev(), after()))
            self._closing = True
            deferred = bool(self._outstanding) and bool(self.handle)
        if deferred:
            return False
        self._destroy()
        return True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SHARED_STAGING: Dict[int, "_HostStaging"] = {}


def pack_to_device(items: Sequence[ArrayLike], device: torch.device, stage: Optional["_HostStaging"] = None) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
    """1-D float32 waveforms (host arrays / CPU tensors / tensors already on ``device``) -> one packed device buffer,
    element offsets and lengths.  Host items go through reusable pinned staging with multi-threaded copies and ONE H2D
    transfer (a pageable ``.cuda()`` per cut is several times slower); every cut starts on a 16-byte boundary."""
    lens = np.array([int(x.shape[0]) for x in items], dtype=np.int64)
    if all(isinstance(x, torch.Tensor) and x.device == device for x in items):
        if len(items) == 1:
            return items[0].contiguous(), np.zeros(1, dtype=np.int64), lens
        offs = np.zeros(len(items), dtype=np.int64)
        np.cumsum(lens[:-1], out=offs[1:])
        return torch.cat([x.contiguous() for x in items]), offs, lens
    padded = (lens + 3) & ~3
    offs = np.zeros(len(items), dtype=np.int64)
    np.cumsum(padded[:-1], out=offs[1:])
    total = int(offs[-1] + lens[-1]) if len(items) else 0
    if stage is None:
        stage = _SHARED_STAGING.setdefault(int(device.index or 0), _HostStaging())
    with stage.lock:
        host, slot = stage.input(total)
        pieces = []
        for x, o in zip(items, offs):
            if isinstance(x, torch.Tensor):
                x = x.detach().cpu().contiguous().numpy()
            pieces.append((int(o), np.ascontiguousarray(x)))
        wave = torch.empty(total, dtype=torch.float32, device=device)
        hv = host.numpy()
        # pack and upload in a few runs of items: while the DMA engine moves run i out of the pinned buffer the host threads are already
        # copying run i+1 into it (the pack, not PCIe, is the slower of the two)
        run_floats = max(_PACK_RUN, (total + 3) // 4)
        a = 0
        while a < len(pieces):
            b, first = a, pieces[a][0]
            while b < len(pieces) and pieces[b][0] + pieces[b][1].shape[0] - first <= run_floats:
                b += 1
            b = max(b, a + 1)
            _parallel_copy(hv, pieces[a:b])
            end = pieces[b - 1][0] + pieces[b - 1][1].shape[0]
            wave[first:end].copy_(host[first:end], non_blocking=True)
            a = b
        stage.sent(slot, device)
    return wave, offs, lens


# --------------------------------------------------------------------------------------
# shared extractor implementation
# --------------------------------------------------------------------------------------
def _as_1d_float(x: ArrayLike, what: str) -> ArrayLike:
    """(T,), (1,T) or (C,T) -> (T,) of channel 0, as Fbank.extract does with ``[0]``
    (extractors.py:107-110).  float32 in [-1, 1] as in the reference, or -- an extension,
    SURVEY 8f row 2 -- int16 PCM, which the device converts as x / 32768 (what the audio backends do on the
    host), so that only half the bytes cross PCIe."""
    if isinstance(x, torch.Tensor):
        if x.dtype not in (torch.float32, torch.int16):
            raise TypeError(f"{what}: expected float32 (or int16 PCM) samples, got {x.dtype}")
        if x.ndim == 1:
            return x  # (the common case of the batch drivers: 60 views per batch, and a torch-level reshape each was a third of the calling thread's time)
        return x[0] if x.ndim == 2 else x.reshape(-1)
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.int16):
        raise TypeError(f"{what}: expected float32 (or int16 PCM) samples, got {x.dtype}")
    return x[0] if x.ndim == 2 else x.reshape(-1)


def _is_pcm16(x: ArrayLike) -> bool:
    return x.dtype in (torch.int16, np.int16)


class _HipExtractor(FeatureExtractor):
    kind: int = -1
    _cpu_outputs: bool = False  # Spectrogram/LogSpectrogram.extract return .cpu() (extractors.py:338-341)

    def __init__(self, config: Optional[Any] = None):
        super().__init__(config=config)
        self._plan: Optional[_Plan] = None
        self._staging: Optional[_HostStaging] = None

    # -- lhotse surface -------------------------------------------------------------------
    @property
    def device(self) -> Union[str, torch.device]:
        return self.config.device

    @property
    def frame_shift(self) -> Seconds:
        return self.config.frame_shift

    def to(self, device: Union[str, torch.device]):
        self.config.device = device
        self._drop_plan()
        return self

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_plan"] = None  # the device handle is per process
        st["_staging"] = None
        st.pop("_lock", None)
        st.pop("_pipeline", None)
        st.pop("_native_pipeline", None)
        return st

    def _drop_plan(self):
        # the host pipeline (two side streams) and the pinned staging belong to the plan's device: they go with it, or a moved
        # extractor would record / wait on the OLD device's streams while the kernel runs on the new one
        self.__dict__.pop("_pipeline", None)
        native = self.__dict__.pop("_native_pipeline", None)
        self._staging = None
        plan, self._plan = self._plan, None
        if native is not None:
            # before the plan it borrows -- and, while results of it are still with a save thread, NOT YET: the pipeline (and behind it the
            # plan) is destroyed when the last outstanding batch is released (NativeHostPipeline.close)
            native.close(after=(plan.close if plan is not None else None))
        elif plan is not None:
            plan.close()

    @property
    def plan(self) -> _Plan:
        if self._plan is None:
            with self._lazy_lock():  # extractors may be shared between reader threads: build the plan once
                if self._plan is None:
                    self._plan = _Plan(self._plan_config(), self.kind, torch.device(self.config.device), mel_floor=self._plan_mel_floor())
        return self._plan

    def _lazy_lock(self):
        lock = self.__dict__.get("_lock")
        if lock is None:  # created on first use so that pickled / copied extractors get their own
            lock = self.__dict__.setdefault("_lock", threading.Lock())
        return lock

    def _plan_config(self):
        return self.config

    def _plan_mel_floor(self) -> float:
        return constants.MEL_FLOOR

    @property
    def kernel_name(self) -> str:
        return self.plan.kernel_name

    def _check_sr(self, sampling_rate: int):
        assert sampling_rate == self.config.sampling_rate, (
            f"{type(self).__name__} was instantiated for sampling_rate "
            f"{self.config.sampling_rate}, but sampling_rate={sampling_rate} was passed to extract(). "
            "Note you can use CutSet/RecordingSet.resample() to change the audio sampling rate."
        )

    # -- device plumbing --------------------------------------------------------------------
    def _stage(self) -> _HostStaging:
        if self._staging is None:
            with self._lazy_lock():
                if self._staging is None:
                    self._staging = _HostStaging()
        return self._staging

    def _to_host(self, dev_tensor: torch.Tensor) -> torch.Tensor:
        if dev_tensor.device.type != "cuda":
            return dev_tensor
        return self._stage().fetch(dev_tensor)

    def _pipe(self) -> "_HostPipeline":
        pipe = self.__dict__.get("_pipeline")
        if pipe is None or pipe.device != self.plan.device:
            with self._lazy_lock():
                pipe = self.__dict__.get("_pipeline")
                if pipe is None or pipe.device != self.plan.device:
                    pipe = self.__dict__["_pipeline"] = _HostPipeline(self.plan.device)
        return pipe

    def _native_pipe(self) -> "NativeHostPipeline":
        """The library-side host pipeline of this extractor's plan (created on first use; dropped with the plan)."""
        plan = self.plan  # (before the lock: creating the plan takes the same, non-reentrant, lock)
        pipe = self.__dict__.get("_native_pipeline")
        if pipe is None or pipe.plan is not plan:
            with self._lazy_lock():
                pipe = self.__dict__.get("_native_pipeline")
                if pipe is None or pipe.plan is not plan:
                    if pipe is not None:
                        pipe.close()
                    pipe = self.__dict__["_native_pipeline"] = NativeHostPipeline(plan)
        return pipe

    def submit_host_items(self, items: Sequence[ArrayLike], sampling_rate: int, half: bool = False) -> "PendingFeatures":
        """Host waveforms of one batch -> PendingFeatures (asynchronous: ``.wait()`` for the packed host matrix, ``.release()`` when done).
        The batch driver's form of ``extract_batch`` (lhotse/cut/set.py:2393-2398 hands a list of host tensors over per batch)."""
        self._check_sr(sampling_rate)
        items = [_as_1d_float(x.squeeze() if x.ndim > 1 else x, "submit_host_items()") for x in items]
        zero_pad = getattr(self.config, "edge_rule", "reflect") == "batch_zero_pad"
        if getattr(self.config, "dither", 0.0):
            raise _lib.HipFeatError(_lib.ERR_UNSUPPORTED, "dither is added by the Python host: use extract_batch")
        return self._native_pipe().submit(items, zero_pad_batch=zero_pad, half=half)

    def submit_host_packed(self, flat: np.ndarray, offs, lens, sampling_rate: int, half: bool = False) -> "PendingFeatures":
        """``submit_host_items`` for cuts that lie in ONE host buffer (a slot of the ring loader): cut b = ``flat[offs[b] : offs[b] + lens[b]]``."""
        self._check_sr(sampling_rate)
        if getattr(self.config, "dither", 0.0):
            raise _lib.HipFeatError(_lib.ERR_UNSUPPORTED, "dither is added by the Python host: use extract_batch")
        return self._native_pipe().submit_packed(flat, offs, lens, zero_pad_batch=getattr(self.config, "edge_rule", "reflect") == "batch_zero_pad", half=half)

    def _host_items_to_host(self, items: Sequence[ArrayLike], padded_len: Optional[int], half: bool = False) -> Tuple[torch.Tensor, np.ndarray]:
        """Host waveforms in, packed host feature matrix out (float32, or float16 converted on the device with `half`), through the
        chunked H2D / kernel / D2H pipeline."""
        plan = self.plan
        lens = np.array([int(x.shape[0]) for x in items], dtype=np.int64)
        padded = None if padded_len is None else np.full(len(items), padded_len, dtype=np.int64)
        frames = plan.frame_counts(lens, padded)
        if int(frames.min(initial=1)) <= 0:  # let the library word the error (and raise it) as for a single launch
            host, frames = self._extract_items_host_fallback(items, padded_len)
            return (host.to(torch.float16) if half else host), frames
        bounds = _HostPipeline.chunk_bounds(lens * (2 if _is_pcm16(items[0]) else 4))

        def upload(a, b):
            wave, offs, ln = self._pack(items[a:b])
            return wave, offs, ln, None if padded is None else padded[a:b]

        return self._pipe().run(plan, bounds, frames, upload, half=half), frames

    def _extract_items_host_fallback(self, items, padded_len):
        packed, frames = self._extract_items(items, padded_len)
        return self._to_host(packed), frames

    def _host_rows_to_host(self, wave2d: torch.Tensor, lens: np.ndarray, zero_pad: bool) -> Tuple[torch.Tensor, np.ndarray]:
        """A padded (B, Smax) HOST tensor (float32 or int16 PCM) + lengths in, packed host feature matrix out.  Page-locked tensors are
        the DMA source themselves; pageable rows are packed chunk by chunk into pinned staging while the previous chunk is on the wire."""
        plan, dev = self.plan, self.plan.device
        smax = int(wave2d.shape[1])
        padded = np.full(len(lens), smax, dtype=np.int64) if zero_pad else None
        frames = plan.frame_counts(lens, padded)
        pcm = wave2d.dtype == torch.int16
        pinned = wave2d.is_pinned()
        if int(frames.min(initial=1)) <= 0:
            bounds = [(0, len(lens))]
        else:
            bounds = _HostPipeline.chunk_bounds(np.full(len(lens), smax * (2 if pcm else 4), dtype=np.int64))
        stage = None if pinned else self._stage()

        def upload(a, b):
            src = wave2d[a:b].reshape(-1)
            n = src.numel()
            d = torch.empty(n, dtype=src.dtype, device=dev)
            if pinned:
                d.copy_(src, non_blocking=True)
            else:
                with stage.lock:
                    host, slot = stage.input(n if not pcm else (n + 1) // 2)
                    hv = host.view(torch.int16) if pcm else host
                    _parallel_copy(hv.numpy(), [(0, src.numpy())])
                    d.copy_(hv[:n], non_blocking=True)
                    stage.sent(slot, dev)
            if pcm:
                f = torch.empty(n, dtype=torch.float32, device=dev)
                plan.lib.check("hipfeat_pcm16_to_float", d.data_ptr(), f.data_ptr(), n, int(torch.cuda.current_stream(dev).cuda_stream))
                d = f
            return d, np.arange(b - a, dtype=np.int64) * smax, lens[a:b], None if padded is None else padded[a:b]

        return self._pipe().run(plan, bounds, frames, upload), frames

    def _pack(self, items: Sequence[ArrayLike]) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
        """Concatenate 1-D waveforms into one device buffer (one H2D copy for host inputs).  Every
        cut starts on a 16-byte boundary so that the kernels can use their 16-byte load path."""
        dev = self.plan.device
        lens = np.array([int(x.shape[0]) for x in items], dtype=np.int64)
        pcm = [_is_pcm16(x) for x in items]
        if any(pcm):
            if not all(pcm):
                raise TypeError("a batch must be all float32 or all int16 PCM")
            return self._pack_pcm16(items, lens)
        if dev.type != "cuda":  # only reachable with a stand-in plan (tests); no staging needed
            padded = (lens + 3) & ~3
            offs = np.zeros(len(items), dtype=np.int64)
            np.cumsum(padded[:-1], out=offs[1:])
            total = int(offs[-1] + lens[-1]) if len(items) else 0
            host = torch.zeros(total, dtype=torch.float32)
            for x, o, n in zip(items, offs, lens):
                host[o : o + n] = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
            return host, offs, lens
        return pack_to_device(items, dev, self._stage())

    def _pack_pcm16(self, items: Sequence[ArrayLike], lens: np.ndarray) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
        """int16 PCM items -> one pinned int16 buffer -> H2D (half the bytes) -> float32 on the device."""
        dev = self.plan.device
        padded = (lens + 7) & ~7  # 16-byte aligned cut starts in the int16 buffer too
        offs = np.zeros(len(items), dtype=np.int64)
        np.cumsum(padded[:-1], out=offs[1:])
        total = int(offs[-1] + lens[-1]) if len(items) else 0
        if dev.type != "cuda":  # only reachable with a stand-in plan (tests): the conversion the device does, x / 32768 (exact)
            host = torch.zeros(total, dtype=torch.float32)
            for x, o, n in zip(items, offs, lens):
                host[o : o + n] = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(torch.float32) / 32768.0
            return host, offs, lens
        if all(isinstance(x, torch.Tensor) and x.device == dev for x in items):
            pcm = torch.zeros(total, dtype=torch.int16, device=dev)
            for x, o, n in zip(items, offs, lens):
                pcm[o : o + n] = x
        else:
            stage = self._stage()
            with stage.lock:
                host, slot = stage.input((total + 1) // 2)
                hv = host.numpy().view(np.int16)
                pieces = []
                for x, o in zip(items, offs):
                    if isinstance(x, torch.Tensor):
                        x = x.detach().cpu().contiguous().numpy()
                    pieces.append((int(o), np.ascontiguousarray(x)))
                _parallel_copy(hv, pieces)
                pcm = torch.empty(total, dtype=torch.int16, device=dev)
                pcm.copy_(host.view(torch.int16)[:total], non_blocking=True)
                stage.sent(slot, dev)
        with torch.cuda.device(dev):
            wave = torch.empty(total, dtype=torch.float32, device=dev)
            self.plan.lib.check("hipfeat_pcm16_to_float", pcm.data_ptr(), wave.data_ptr(), total, int(torch.cuda.current_stream(dev).cuda_stream))
        return wave, offs, lens

    def extract_collated(
        self, samples: Sequence[ArrayLike], sampling_rate: int, padding_value: float = LOG_EPSILON
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """``extract_batch`` + ``collate_matrices(features, padding_value=LOG_EPSILON)`` in one pass
        (lhotse/dataset/input_strategies.py:441-462): returns the dense ``(B, Tmax, F)`` float32 tensor on the
        extractor's device, padded with ``padding_value``, and the int64 frame counts.  The kernels write every
        cut straight into its slot and a fill kernel writes only the padding rows -- no per-cut copies."""
        self._check_sr(sampling_rate)
        items = [_as_1d_float(x.squeeze() if x.ndim > 1 else x, "extract_collated()") for x in samples]
        if not items:
            raise ValueError("extract_collated(): empty batch")
        zero_pad = getattr(self.config, "edge_rule", "reflect") == "batch_zero_pad"
        with torch.no_grad():
            wave, offs, lens = self._pack(items)
            padded = np.full(len(items), int(lens.max()), dtype=np.int64) if zero_pad else None
            try:
                out, frames = self.plan.run_collated(wave, offs, lens, padded, float(padding_value))
            except _lib.HipFeatError as e:
                if e.status == _lib.ERR_TOO_SHORT:
                    raise ValueError(str(e)) from e
                raise
        return out, torch.from_numpy(frames)

    def _extract_items(self, items: Sequence[ArrayLike], padded_len: Optional[int] = None) -> Tuple[torch.Tensor, np.ndarray]:
        wave, offs, lens = self._pack(items)
        padded = None if padded_len is None else np.full(len(items), padded_len, dtype=np.int64)
        try:
            return self.plan.run(wave, offs, lens, padded)
        except _lib.HipFeatError as e:
            if e.status == _lib.ERR_TOO_SHORT:
                raise ValueError(str(e)) from e
            raise

    # -- extract ----------------------------------------------------------------------------
    def extract(self, samples: ArrayLike, sampling_rate: int) -> ArrayLike:
        self._check_sr(sampling_rate)
        is_numpy = not isinstance(samples, torch.Tensor)
        x = _as_1d_float(samples, "extract()")
        with torch.no_grad():
            feats, _ = self._extract_items([x])
        if is_numpy:
            return self._to_host(feats).numpy()
        return self._to_host(feats) if self._cpu_outputs else feats

    def extract_batch(
        self,
        samples: Union[np.ndarray, torch.Tensor, Sequence[np.ndarray], Sequence[torch.Tensor]],
        sampling_rate: int,
        lengths: Optional[Union[np.ndarray, torch.Tensor]] = None,
    ) -> Union[np.ndarray, torch.Tensor, List[np.ndarray], List[torch.Tensor]]:
        self._check_sr(sampling_rate)
        zero_pad = getattr(self.config, "edge_rule", "reflect") == "batch_zero_pad"
        input_is_list = False
        input_is_torch = False
        with torch.no_grad():
            if lengths is not None:
                assert isinstance(samples, torch.Tensor), "If `lengths` is provided, `samples` must be a batched and padded torch.Tensor."
                if samples.dtype not in (torch.float32, torch.int16):
                    raise TypeError(f"extract_batch(): expected float32 (or int16 PCM) samples, got {samples.dtype}")
                lens = np.asarray(lengths.cpu() if isinstance(lengths, torch.Tensor) else lengths).astype(np.int64).reshape(-1)
                assert samples.ndim == 2 and samples.shape[0] == len(lens)
                smax = int(samples.shape[1])
                assert int(lens.max(initial=0)) <= smax
                dev = self.plan.device
                wave = samples.contiguous()
                host_done = False
                try:
                    if wave.device.type == "cpu" and dev.type == "cuda":
                        packed, frames = self._host_rows_to_host(wave, lens, zero_pad)
                        host_done = True
                    else:
                        if wave.device != dev:
                            wave = wave.to(dev)
                        if wave.dtype == torch.int16:
                            with torch.cuda.device(dev):
                                f = torch.empty(wave.shape, dtype=torch.float32, device=dev)
                                self.plan.lib.check("hipfeat_pcm16_to_float", wave.data_ptr(), f.data_ptr(), wave.numel(),
                                                    int(torch.cuda.current_stream(dev).cuda_stream))
                            wave = f
                        offs = np.arange(len(lens), dtype=np.int64) * smax
                        padded = np.full(len(lens), smax, dtype=np.int64) if zero_pad else None
                        packed, frames = self.plan.run(wave.reshape(-1), offs, lens, padded)
                except _lib.HipFeatError as e:
                    if e.status == _lib.ERR_TOO_SHORT:
                        raise ValueError(str(e)) from e
                    raise
                # with `lengths` the reference always hands back numpy (extractors.py:539-540)
            else:
                if isinstance(samples, (list, tuple)):
                    input_is_list = True
                    items = list(samples)
                elif samples.ndim > 1:
                    items = list(samples)
                else:
                    items = [samples.reshape(1, -1)]
                input_is_torch = any(isinstance(x, torch.Tensor) for x in items)
                # the reference squeezes every item (extractors.py:519-522)
                items = [x if (x.ndim == 1 and x.dtype in (torch.float32, np.float32, torch.int16, np.int16)) else _as_1d_float(x.squeeze() if x.ndim > 1 else x, "extract_batch()") for x in items]
                pmax = max(int(x.shape[0]) for x in items) if zero_pad else None
                host_done = False
                to_host = not input_is_torch or self._cpu_outputs
                on_host = all(not isinstance(x, torch.Tensor) or x.device.type == "cpu" for x in items)
                if to_host and on_host and self.plan.device.type == "cuda":
                    try:
                        packed, frames = self._host_items_to_host(items, pmax)
                    except _lib.HipFeatError as e:
                        if e.status == _lib.ERR_TOO_SHORT:
                            raise ValueError(str(e)) from e
                        raise
                    host_done = True
                else:
                    packed, frames = self._extract_items(items, pmax)

            if not host_done and (not input_is_torch or self._cpu_outputs):
                packed = self._to_host(packed)
            if not input_is_torch:
                packed = packed.numpy()
            bounds = np.concatenate([[0], np.cumsum(frames)])
            result = [packed[int(bounds[i]) : int(bounds[i + 1])] for i in range(len(frames))]

        if len(result) == 1:
            # NB with `lengths` (a padded batch came in) a batch comes out, even for one item: the
            # reference returns the bare matrix here (extractors.py:543-547), which its own batch
            # driver then mis-iterates row by row (cut/set.py:2308, single-cut collated batches).
            if lengths is not None:
                return packed.reshape(1, *result[0].shape)
            return result if input_is_list else result[0]
        if all(item.shape == result[0].shape for item in result[1:]):
            # equal lengths: the packed matrix already is the stacked batch
            return packed.reshape(len(result), *result[0].shape)
        return result


def _log_mix(features_a: np.ndarray, features_b: np.ndarray, energy_scaling_factor_b: float) -> np.ndarray:
    # extractors.py:134-144
    return np.log(np.maximum(EPSILON, np.exp(features_a) + energy_scaling_factor_b * np.exp(features_b)))


# --------------------------------------------------------------------------------------
# the four registered extractors
# --------------------------------------------------------------------------------------
@register_extractor
class HipFbank(_HipExtractor):
    """Log-mel filterbank energies; drop-in for ``Fbank`` (extractors.py:66-152)."""

    name = "hip-fbank"
    config_type = HipFbankConfig
    kind = KIND_FBANK

    def feature_dim(self, sampling_rate: int) -> int:
        return self.config.num_filters

    mix = staticmethod(_log_mix)

    @staticmethod
    def compute_energy(features: np.ndarray) -> float:
        return float(np.sum(np.exp(features)))

    @staticmethod
    def scale(features: np.ndarray, energy_scaling_factor: float) -> np.ndarray:
        return features + np.log(energy_scaling_factor)


@register_extractor
class HipMfcc(_HipExtractor):
    """MFCC (log-mel -> DCT -> lifter); drop-in for ``Mfcc`` (extractors.py:200-262)."""

    name = "hip-mfcc"
    config_type = HipMfccConfig
    kind = KIND_MFCC

    def feature_dim(self, sampling_rate: int) -> int:
        return self.config.num_ceps


class _SpecMixin:
    def feature_dim(self, sampling_rate: int) -> int:
        c = self.config
        return constants.frame_sizes(c.sampling_rate, c.frame_length, c.frame_shift, c.round_to_power_of_two)[2] // 2 + 1

    @staticmethod
    def mix(features_a: np.ndarray, features_b: np.ndarray, energy_scaling_factor_b: float) -> np.ndarray:
        return features_a + energy_scaling_factor_b * features_b

    @staticmethod
    def compute_energy(features: np.ndarray) -> float:
        return float(np.sum(features))

    @staticmethod
    def scale(features: np.ndarray, energy_scaling_factor: float) -> np.ndarray:
        return energy_scaling_factor * features


@register_extractor
class HipSpectrogram(_SpecMixin, _HipExtractor):
    """Power / magnitude spectrogram; drop-in for ``Spectrogram`` (extractors.py:296-372)."""

    name = "hip-spectrogram"
    config_type = HipSpectrogramConfig
    kind = KIND_SPECTROGRAM
    _cpu_outputs = True
    log_domain = False  # power / magnitude values: not storable in binary16 (storage.py::HipArchiveF16Writer)


@register_extractor
class HipLogSpectrogram(_SpecMixin, _HipExtractor):
    """Log spectrogram; drop-in for ``LogSpectrogram`` (extractors.py:406-482)."""

    name = "hip-log-spectrogram"
    config_type = HipLogSpectrogramConfig
    kind = KIND_LOG_SPECTROGRAM
    _cpu_outputs = True
