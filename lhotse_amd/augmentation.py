"""
GPU speed perturbation / sinc resampling behind lhotse's ``AudioTransform`` interface
("next" row 1).

Mirrors lhotse/augmentation/torchaudio.py:26-140 (``Speed``, ``Resample``, ``get_or_create_resampler``) and
lhotse/augmentation/resample.py:42-142 (``Resample`` module, here ``HipResampleTensor``): same names, arguments,
output lengths and dict round trip; the arithmetic -- zero pad, strided polyphase FIR, trim -- runs in
``resample_kernel`` (lhotse_amd/csrc/kernel_resample.hpp) through the C ABI (``hipfeat_resample``).

    fn = HipSpeed(factor=1.1)                      # AudioTransform: numpy (C, T) -> numpy (C, T')
    wave = fn(samples, 16000)
    OnTheFlyFeatures(HipFbank(), wave_transforms=[...])      # unchanged; or, staying on the device:
    ys = get_or_create_resampler(17600, 16000)(x_cuda)       # torch (..., T) -> (..., T') on the same device

There is no CPU fallback: without a HIP device the call raises ``HipFeatError``.
"""
from __future__ import annotations

import threading
from dataclasses import asdict, dataclass
from decimal import ROUND_HALF_DOWN, ROUND_HALF_UP, Decimal
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, constants
from .compat import HAVE_LHOTSE, Seconds

if HAVE_LHOTSE:  # pragma: no cover - authoring container only
    from lhotse.augmentation.transform import AudioTransform  # type: ignore
else:

    class AudioTransform:  # stand-in with the same surface (lhotse/augmentation/transform.py:9-76)
        KNOWN_TRANSFORMS: Dict[str, type] = {}

        def __init_subclass__(cls, **kwargs):
            AudioTransform.KNOWN_TRANSFORMS.setdefault(cls.__name__, cls)
            super().__init_subclass__(**kwargs)

        def to_dict(self) -> dict:
            return {"name": type(self).__name__, "kwargs": asdict(self)}

        @staticmethod
        def from_dict(data: dict) -> "AudioTransform":
            assert data["name"] in AudioTransform.KNOWN_TRANSFORMS, f"Unknown transform type: {data['name']}"
            return AudioTransform.KNOWN_TRANSFORMS[data["name"]](**data["kwargs"])


def perturb_num_samples(num_samples: int, factor: float) -> int:
    """Number of samples after speed perturbation (lhotse/utils.py:649-654)."""
    rounding = ROUND_HALF_UP if factor >= 1.0 else ROUND_HALF_DOWN
    return int(Decimal(round(num_samples / factor, ndigits=8)).quantize(0, rounding=rounding))


def _compute_num_samples(duration: Seconds, sampling_rate: int) -> int:
    """lhotse/utils.py:657-673"""
    return int(Decimal(round(duration * sampling_rate, ndigits=8)).quantize(0, rounding=ROUND_HALF_UP))


class HipResampleTensor:
    """Device counterpart of the ``Resample`` nn.Module (lhotse/augmentation/resample.py:42-142): the filter
    bank lives in HBM; calling it resamples every row of a ``(..., T)`` float32 tensor."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000, lowpass_filter_width: int = 6, rolloff: float = 0.99,
                 device: Union[str, torch.device, None] = None):
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.lowpass_filter_width, self.rolloff = lowpass_filter_width, rolloff
        self.kernel, self.width, self.orig, self.new = constants.sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
        self.lib = _lib.load()
        self.handle = 0
        dev = torch.device("cuda" if device is None else device)
        if dev.type != "cuda":
            raise _lib.HipFeatError(1, f"HipResampleTensor runs on an AMD GPU ('cuda[:i]' device), got device={dev}")
        if not torch.cuda.is_available():
            raise _lib.HipFeatError(2, "no HIP device is visible (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        out = np.zeros(1, dtype=np.uint64)
        self.lib.check("hipfeat_resampler_create", self.orig, self.new, self.width, _lib.addr(self.kernel), int(self.device.index), _lib.addr(out))
        self.handle = int(out[0])

    # ---- lengths -------------------------------------------------------------------------------------------
    def output_length(self, num_samples: int) -> int:
        if self.orig == self.new:
            return int(num_samples)
        return int(self.lib.raw("hipfeat_resampled_length", int(num_samples), self.orig, self.new))

    def output_lengths(self, num_samples: np.ndarray) -> np.ndarray:
        """Vectorised ``output_length`` (same float32 rounding, resample.py:309)."""
        n = _lib.i64(num_samples)
        if self.orig == self.new:
            return n
        return np.ceil((self.new * n / self.orig).astype(np.float32)).astype(np.int64)

    # ---- packed ragged batch, device resident ---------------------------------------------------------------
    def run(self, wave: torch.Tensor, offsets: np.ndarray, lengths: np.ndarray, align: bool = True) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
        """wave: contiguous float32 on self.device holding every cut; -> (output buffer, out_offsets, out_lengths).
        With ``align`` every output cut starts on a 16-byte boundary (up to 3 floats of slack between cuts), which is
        what lets the feature kernels take their LDS-DMA load path on the result; otherwise cuts are back to back."""
        assert wave.dtype == torch.float32 and wave.is_contiguous() and wave.device == self.device
        offsets, lengths = _lib.i64(offsets), _lib.i64(lengths)
        out_lens = self.output_lengths(lengths)
        out_offs = np.zeros(len(lengths), dtype=np.int64)
        step = ((out_lens + 3) & ~3) if align else out_lens
        np.cumsum(step[:-1], out=out_offs[1:])
        total = int(out_offs[-1] + out_lens[-1]) if len(lengths) else 0
        with torch.cuda.device(self.device):
            out = torch.empty(total, dtype=torch.float32, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.lib.check("hipfeat_resample", self.handle, wave.data_ptr(), _lib.addr(offsets), _lib.addr(lengths), int(len(lengths)),
                           out.data_ptr(), _lib.addr(out_offs), int(stream))
        return out, out_offs, out_lens

    def resample_batch(self, waves: Sequence[Union[np.ndarray, torch.Tensor]]) -> List[torch.Tensor]:
        """Ragged batch of 1-D waveforms (host or device) -> list of resampled device tensors (one launch)."""
        ts = [torch.as_tensor(w).reshape(-1) for w in waves]
        for t in ts:
            if t.dtype != torch.float32:
                raise TypeError(f"expected float32 samples, got {t.dtype}")
        if not ts:
            return []
        from .extractors import pack_to_device  # pinned staging + one H2D for host inputs

        wave, offsets, lengths = pack_to_device(ts, self.device)
        if self.orig == self.new:
            return [wave[o : o + n] for o, n in zip(offsets.tolist(), lengths.tolist())]
        out, out_offs, out_lens = self.run(wave, offsets, lengths)
        return [out[o : o + n] for o, n in zip(out_offs.tolist(), out_lens.tolist())]

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        """(..., T) -> (..., T'), result on the input's device (resample.py:126-142)."""
        if not isinstance(waveform, torch.Tensor):
            raise TypeError("expected a torch.Tensor")
        if self.orig_freq == self.new_freq:
            return waveform
        if waveform.dtype != torch.float32:
            raise TypeError(f"expected float32 samples, got {waveform.dtype}")
        shape = waveform.shape
        T = int(shape[-1])
        rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
        x = waveform.reshape(rows, T).to(self.device).contiguous()
        out, _, out_lens = self.run(x.view(-1), np.arange(rows, dtype=np.int64) * T, np.full(rows, T, dtype=np.int64), align=False)
        y = out.view(shape[:-1] + (int(out_lens[0]) if rows else 0,))
        return y.to(waveform.device)  # the input's device, whichever GPU (or the host) that is

    forward = __call__

    def close(self):
        if self.handle:
            try:
                self.lib.raw("hipfeat_resampler_destroy", self.handle)
            finally:
                self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_precompiled_resamplers: Dict[Tuple[int, int, int], HipResampleTensor] = {}
_cache_lock = threading.Lock()


def get_or_create_resampler(source_sampling_rate: int, target_sampling_rate: int, device: Union[str, torch.device, None] = None) -> HipResampleTensor:
    """lhotse/augmentation/torchaudio.py:72-83, keyed per device as well."""
    dev = torch.device("cuda" if device is None else device)
    index = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    key = (int(source_sampling_rate), int(target_sampling_rate), int(index))
    with _cache_lock:
        r = _precompiled_resamplers.get(key)
        if r is None:
            r = _precompiled_resamplers[key] = HipResampleTensor(key[0], key[1], device=torch.device(dev.type, index))
        return r


@dataclass
class HipSpeed(AudioTransform):
    """Speed perturbation (``sox speed``): resample from round(sr * factor) back to sr on the GPU.
    Drop-in for ``lhotse.augmentation.Speed`` (torchaudio.py:26-68)."""

    factor: float
    device: str = "cuda"

    def __call__(self, samples: Union[np.ndarray, torch.Tensor], sampling_rate: int) -> Union[np.ndarray, torch.Tensor]:
        resampler = get_or_create_resampler(round(sampling_rate * self.factor), sampling_rate, self.device)
        if isinstance(samples, torch.Tensor):  # device-resident use: stays a tensor on its device
            return resampler(samples)
        return resampler(torch.from_numpy(np.ascontiguousarray(samples))).numpy()

    def reverse_timestamps(self, offset: Seconds, duration: Optional[Seconds], sampling_rate: int) -> Tuple[Seconds, Optional[Seconds]]:
        """Offset/duration of the original audio that yields the requested perturbed span (torchaudio.py:44-68)."""
        start_sample = perturb_num_samples(_compute_num_samples(offset, sampling_rate), 1 / self.factor)
        num_samples = None if duration is None else perturb_num_samples(_compute_num_samples(duration, sampling_rate), 1 / self.factor)
        return start_sample / sampling_rate, None if num_samples is None else num_samples / sampling_rate


@dataclass
class HipResample(AudioTransform):
    """Sampling-rate conversion (``sox rate``) on the GPU; drop-in for ``lhotse.augmentation.Resample``
    (torchaudio.py:86-164) with the sinc backend."""

    source_sampling_rate: int
    target_sampling_rate: int
    device: str = "cuda"

    def __post_init__(self):
        self.source_sampling_rate = int(self.source_sampling_rate)
        self.target_sampling_rate = int(self.target_sampling_rate)

    @property
    def resampler(self) -> HipResampleTensor:
        return get_or_create_resampler(self.source_sampling_rate, self.target_sampling_rate, self.device)

    def __call__(self, samples: Union[np.ndarray, torch.Tensor], *args, **kwargs) -> Union[np.ndarray, torch.Tensor]:
        if self.source_sampling_rate == self.target_sampling_rate:
            return samples
        if isinstance(samples, torch.Tensor):
            return self.resampler(samples)
        return self.resampler(torch.from_numpy(np.ascontiguousarray(samples))).numpy()

    def reverse_timestamps(self, offset: Seconds, duration: Optional[Seconds], sampling_rate: int) -> Tuple[Seconds, Optional[Seconds]]:
        """Timestamps do not change with the sampling rate (torchaudio.py:142-164): whole-sample snapping only."""
        if self.source_sampling_rate == self.target_sampling_rate:
            return offset, duration
        old_offset = _compute_num_samples(offset, self.source_sampling_rate) / self.source_sampling_rate
        old_duration = None if duration is None else _compute_num_samples(duration, self.source_sampling_rate) / self.source_sampling_rate
        return old_offset, old_duration


# ---- speed perturbation of a packed mini-batch, in place of the host loop of config 5 -------------------------------------------
def perturbed_tail_floats(lengths: np.ndarray, factors: Sequence[float], sampling_rate: int) -> int:
    """Floats the resampled cuts of a mini-batch need behind its inputs (every resampled cut starts on a 16-byte boundary)."""
    total = 0
    for f in sorted(set(float(x) for x in factors)):
        if f == 1.0:
            continue
        idx = np.nonzero(np.asarray(factors, dtype=np.float64) == f)[0]
        orig, new = constants.sinc_resample_kernel(round(sampling_rate * f), sampling_rate)[2:4]
        out = np.ceil((new * _lib.i64(lengths)[idx] / orig).astype(np.float32)).astype(np.int64)
        total += int(((out + 3) & ~3).sum())
    return total


def perturb_speed_in_arena(arena: torch.Tensor, offsets: np.ndarray, lengths: np.ndarray, factors: Sequence[float], sampling_rate: int,
                           tail_start: int) -> Tuple[np.ndarray, np.ndarray]:
    """Mixed-factor speed perturbation of a device-resident packed mini-batch (``PerturbSpeed`` picks one factor per cut,
    lhotse/dataset/cut_transforms/perturb_speed.py:8-47; the arithmetic is ``Speed``, lhotse/augmentation/torchaudio.py:26-42).

    ``arena`` is ONE float32 device buffer: the cuts at ``offsets`` / ``lengths`` in its front part, free space from ``tail_start`` on
    (``perturbed_tail_floats`` says how much).  Cuts with factor 1 stay where they are; the others are resampled -- one launch per
    distinct factor -- into the tail.  Returns the per-cut (offsets, lengths) of the perturbed batch inside the same arena, i.e. exactly
    what ``hipfeat_extract*`` takes next: no copy of the unperturbed cuts, no host round trip, no second buffer."""
    assert arena.dtype == torch.float32 and arena.is_contiguous() and arena.ndim == 1
    offsets, lengths = _lib.i64(offsets).copy(), _lib.i64(lengths).copy()
    fac = np.asarray(factors, dtype=np.float64)
    assert len(fac) == len(lengths)
    tail = (int(tail_start) + 3) & ~3
    dev = arena.device
    for f in sorted(set(fac.tolist())):
        if f == 1.0:
            continue
        idx = np.nonzero(fac == f)[0]
        r = get_or_create_resampler(round(sampling_rate * f), sampling_rate, dev)
        out_lens = r.output_lengths(lengths[idx])
        out_offs = np.zeros(len(idx), dtype=np.int64)
        np.cumsum(((out_lens + 3) & ~3)[:-1], out=out_offs[1:])
        out_offs += tail
        end = int(out_offs[-1] + out_lens[-1])
        if end > arena.numel():
            raise ValueError(f"arena too small: {arena.numel()} floats, the perturbed cuts need {end} (see perturbed_tail_floats)")
        in_offs, in_lens = np.ascontiguousarray(offsets[idx]), np.ascontiguousarray(lengths[idx])  # (named: they must outlive the call)
        with torch.cuda.device(dev):
            r.lib.check("hipfeat_resample", r.handle, arena.data_ptr(), _lib.addr(in_offs), _lib.addr(in_lens), int(len(idx)), arena.data_ptr(),
                        _lib.addr(out_offs), int(torch.cuda.current_stream(dev).cuda_stream))
        offsets[idx], lengths[idx] = out_offs, out_lens
        tail = (end + 3) & ~3
    return offsets, lengths


def _raw_stream(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device` as an integer (the private accessor that skips building a Stream object,
    where this torch has it: the call sits on a per-mini-batch path)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:  # pragma: no cover
        return torch.cuda.current_stream(device).cuda_stream


# ---- the same in ONE launch for all factors, fused with what else has to precede the feature launch ------------------------------
class HipSpeedBank:
    """The resamplers a mini-batch may refer to, resident on one device (``hipfeat_speed_bank``, include/hipfeat.h): mixed-factor speed
    perturbation of a packed mini-batch + the collated feature extraction as a PAIR of launches with no host -> device copy in front of
    them (``hipfeat_minibatch_plan`` / ``hipfeat_minibatch_run``).  Factors must be among 0.9 / 1.1 (the compile-time ratios of the
    mixed launch) and 1.0; anything else raises ``HipFeatError`` (UNSUPPORTED) -- use ``perturb_speed_in_arena``.

        bank = HipSpeedBank([0.9, 1.0, 1.1], 16000, "cuda:0")
        feats, frames, offs, lens = bank.extract_collated(extractor.plan, arena, offsets, lengths, bank.index_of(factors), tail_start, LOG_EPSILON)

    bit-identical to ``perturb_speed_in_arena`` + ``plan.run_collated``."""

    def __init__(self, factors: Sequence[float], sampling_rate: int, device: Union[str, torch.device, None] = None):
        self.sampling_rate = int(sampling_rate)
        self.factors = sorted({float(f) for f in factors if float(f) != 1.0})
        self.resamplers = [get_or_create_resampler(round(sampling_rate * f), sampling_rate, device) for f in self.factors]
        self.lib = _lib.load()
        if self.resamplers:
            self.device = self.resamplers[0].device
        else:  # a bank without resamplers (plain collated extraction through the launch pair): the device is the caller's
            dev = torch.device("cuda" if device is None else device)
            self.device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        handles = np.array([r.handle for r in self.resamplers], dtype=np.uint64)
        out = np.zeros(1, dtype=np.uint64)
        self.handle = 0
        self.lib.check("hipfeat_speed_bank_create", _lib.addr(handles) if len(handles) else None, len(handles), _lib.addr(out))
        self.handle = int(out[0])
        self._info = np.zeros(4, dtype=np.int64)
        self._info_addr = _lib.addr(self._info)
        self._plan_fn, self._run_fn = self.lib.fn("hipfeat_minibatch_plan"), self.lib.fn("hipfeat_minibatch_run")  # bound once: this is a per-mini-batch path
        self._lock = threading.Lock()

    def index_of(self, factors: Sequence[float]) -> np.ndarray:
        """Per-cut bank index (int32; -1 = factor 1.0 = the cut stays where it is)."""
        fac = np.asarray(factors, dtype=np.float64)
        idx = np.full(len(fac), -1, dtype=np.int32)
        for k, f in enumerate(self.factors):
            idx[fac == f] = k
        bad = (idx < 0) & (fac != 1.0)
        if bad.any():
            raise ValueError(f"factors {sorted(set(fac[bad].tolist()))} are not in this bank ({self.factors})")
        return idx

    def extract_collated(self, plan, arena: torch.Tensor, offsets: np.ndarray, lengths: np.ndarray, bank_index: np.ndarray, tail_start: int,
                         pad_value: float, max_samples: Optional[np.ndarray] = None, zero_pad_batch: bool = False,
                         stream: Optional[int] = None, group_sizes: Optional[np.ndarray] = None):
        """-> (features (B, Tmax, F) on the arena's device, frame counts, per-cut offsets and lengths of the PERTURBED batch in the arena).
        ``offsets`` / ``lengths`` int64, ``bank_index`` int32 (``index_of``), all C-contiguous numpy arrays; the arena must hold
        ``perturbed_tail_floats`` floats behind ``tail_start``.

        ``group_sizes`` (int64 array, K entries adding up to B): the batch is K mini-batches -- a prefetching loader's -- served by ONE
        pair of launches; the first result then is a LIST of K dense ``(B_k, Tmax_k, F)`` tensors (views of one allocation)."""
        B = len(lengths)
        K = 0 if group_sizes is None else len(group_sizes)
        res = np.empty(3 * B + 2 * K, dtype=np.int64)  # offsets, lengths, frames of the perturbed batch; (first row, rows per cut) per group
        a = res.__array_interface__["data"][0]
        info, fn = self._info, self._plan_fn
        with self._lock:  # (the info block is shared; the library serialises the calls anyway)
            st = fn(self.handle, plan.handle, B, offsets.__array_interface__["data"][0], lengths.__array_interface__["data"][0],
                    bank_index.__array_interface__["data"][0], None if max_samples is None else max_samples.__array_interface__["data"][0],
                    int(tail_start), 1 if zero_pad_batch else 0, K, None if K == 0 else group_sizes.__array_interface__["data"][0],
                    a, a + 8 * B, a + 16 * B, (a + 24 * B) if K else None, self._info_addr)
            if st != 0:
                raise _lib.HipFeatError(int(st), self.lib.last_error())
            ticket, tmax, rows = int(info[0]), int(info[2]), int(info[3])
        F = plan.feature_dim
        if stream is None:
            stream = _raw_stream(arena.device)
        if K > 1:
            flat = torch.empty(rows * F, dtype=torch.float32, device=arena.device)
            st = self._run_fn(self.handle, ticket, arena.data_ptr(), arena.numel(), flat.data_ptr(), -1, float(pad_value), int(stream))
            g = res[3 * B :].tolist()
            out = [flat[g[2 * k] * F : (g[2 * k] + int(group_sizes[k]) * g[2 * k + 1]) * F].view(int(group_sizes[k]), g[2 * k + 1], F) for k in range(K)]
        else:
            out = torch.empty((B, tmax, F), dtype=torch.float32, device=arena.device)
            st = self._run_fn(self.handle, ticket, arena.data_ptr(), arena.numel(), out.data_ptr(), tmax, float(pad_value), int(stream))
            if K == 1:
                out = [out]
        if st != 0:
            raise _lib.HipFeatError(int(st), self.lib.last_error())
        return out, res[2 * B : 3 * B], res[:B], res[B : 2 * B]

    def close(self):
        if self.handle:
            try:
                self.lib.raw("hipfeat_speed_bank_destroy", self.handle)
            finally:
                self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
