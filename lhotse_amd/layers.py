"""
The Kaldi feature *layers* on the HIP path: ``HipWav2Spec``, ``HipWav2LogSpec``, ``HipWav2LogFilterBank``, ``HipWav2MFCC``.

Drop-ins for the ``torch.nn.Module``s of lhotse/features/kaldi/layers.py (``Wav2Spec`` :336-402, ``Wav2LogSpec`` :405-473,
``Wav2LogFilterBank`` :476-578, ``Wav2MFCC`` :581-724) where they are used for inference: same constructor arguments and
defaults (in the same order), same attributes, ``forward(x)`` maps a ``(B, T)`` (or ``(T,)``) float32 batch of equally long
waveforms on the GPU to ``(B, num_frames, F)`` on the same device -- one fused launch instead of the ~12 tensor ops of
``Wav2Win`` + ``_rfft`` + matmul + log.  Not covered, by design: autograd (inputs that
require grad are refused), TorchScript, ``online_inference``.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _lib, constants
from . import extractors as _E
from .compat import EPSILON, Seconds

__all__ = ["HipWav2Spec", "HipWav2LogSpec", "HipWav2LogFilterBank", "HipWav2MFCC"]


class _HipLayer(torch.nn.Module):
    kind: int = -1

    def __init__(self, **opts):
        super().__init__()
        for k, v in opts.items():
            setattr(self, k, v)
        self._opts = dict(opts)
        n, shift, fft = constants.frame_sizes(opts["sampling_rate"], opts["frame_length"], opts["frame_shift"], opts["round_to_power_of_two"])
        self.fft_length = fft  # Wav2FFT.fft_length (layers.py:281-286)
        self._n, self._shift = n, shift
        self._plans = {}

    def _plan_for(self, device: torch.device):
        key = (device.type, device.index)
        plan = self._plans.get(key)
        if plan is None:
            cfg = SimpleNamespace(**self._opts)
            for name, default in (("num_filters", 0), ("num_ceps", 0), ("cepstral_lifter", 0), ("low_freq", 20.0), ("high_freq", -400.0),
                                  ("norm_filters", False), ("torchaudio_compatible_mel_scale", True), ("use_fft_mag", False)):
                if not hasattr(cfg, name):
                    setattr(cfg, name, default)
            plan = _E._Plan(cfg, self.kind, device)  # refuses anything but a 'cuda' device: there is no CPU fallback
            self._plans[key] = plan
        return plan

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_plans"] = {}  # device handles are per process
        return st

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            raise TypeError(f"{type(self).__name__}.forward expects a torch.Tensor, got {type(x).__name__}")
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError(f"{type(self).__name__} is inference-only (no autograd); call it under torch.no_grad() or detach the input")
        if x.dtype != torch.float32:
            raise TypeError(f"{type(self).__name__}: expected float32 samples, got {x.dtype}")
        squeeze = x.ndim == 1
        if squeeze:
            x = x.unsqueeze(0)
        assert x.ndim == 2, f"expected a (B, T) batch of waveforms, got shape {tuple(x.shape)}"
        plan = self._plan_for(x.device)
        B, T = x.shape
        wave = x.detach().contiguous().reshape(-1)
        offs = np.arange(B, dtype=np.int64) * T
        lens = np.full(B, T, dtype=np.int64)
        try:
            packed, frames = plan.run(wave, offs, lens, None)
        except _lib.HipFeatError as e:
            if e.status == _lib.ERR_TOO_SHORT:
                raise ValueError(str(e)) from e
            raise
        out = packed.reshape(B, int(frames[0]) if B else 0, plan.feature_dim)
        return out[0] if squeeze else out


def _common(sampling_rate, frame_length, frame_shift, round_to_power_of_two, remove_dc_offset, preemph_coeff, window_type, dither, snip_edges,
            energy_floor, raw_energy, use_energy):
    return dict(sampling_rate=sampling_rate, frame_length=frame_length, frame_shift=frame_shift, round_to_power_of_two=round_to_power_of_two,
                remove_dc_offset=remove_dc_offset, preemph_coeff=preemph_coeff, window_type=window_type, dither=dither, snip_edges=snip_edges,
                energy_floor=energy_floor, raw_energy=raw_energy, use_energy=use_energy)


class HipWav2Spec(_HipLayer):
    """Power (or magnitude) spectrum; ``use_energy`` (default True, as in the reference) puts the log-energy in bin 0."""

    kind = _E.KIND_SPECTROGRAM

    def __init__(self, sampling_rate: int = 16000, frame_length: Seconds = 0.025, frame_shift: Seconds = 0.01, round_to_power_of_two: bool = True,
                 remove_dc_offset: bool = True, preemph_coeff: float = 0.97, window_type: str = "povey", dither: float = 0.0,
                 snip_edges: bool = False, energy_floor: float = EPSILON, raw_energy: bool = True, use_energy: bool = True,
                 use_fft_mag: bool = False):
        super().__init__(**_common(sampling_rate, frame_length, frame_shift, round_to_power_of_two, remove_dc_offset, preemph_coeff, window_type,
                                   dither, snip_edges, energy_floor, raw_energy, use_energy), use_fft_mag=use_fft_mag)


class HipWav2LogSpec(_HipLayer):
    """``log(spectrum + 1e-15)``; ``use_energy`` (default True) puts the log-energy in bin 0."""

    kind = _E.KIND_LOG_SPECTROGRAM

    def __init__(self, sampling_rate: int = 16000, frame_length: Seconds = 0.025, frame_shift: Seconds = 0.01, round_to_power_of_two: bool = True,
                 remove_dc_offset: bool = True, preemph_coeff: float = 0.97, window_type: str = "povey", dither: float = 0.0,
                 snip_edges: bool = False, energy_floor: float = EPSILON, raw_energy: bool = True, use_energy: bool = True,
                 use_fft_mag: bool = False):
        super().__init__(**_common(sampling_rate, frame_length, frame_shift, round_to_power_of_two, remove_dc_offset, preemph_coeff, window_type,
                                   dither, snip_edges, energy_floor, raw_energy, use_energy), use_fft_mag=use_fft_mag)


class HipWav2LogFilterBank(_HipLayer):
    """Log-mel filterbank energies (``use_energy`` prepends the log-energy column, layers.py:575-576)."""

    kind = _E.KIND_FBANK

    def __init__(self, sampling_rate: int = 16000, frame_length: Seconds = 0.025, frame_shift: Seconds = 0.01, round_to_power_of_two: bool = True,
                 remove_dc_offset: bool = True, preemph_coeff: float = 0.97, window_type: str = "povey", dither: float = 0.0,
                 snip_edges: bool = False, energy_floor: float = EPSILON, raw_energy: bool = True, use_energy: bool = False,
                 use_fft_mag: bool = False, low_freq: float = 20.0, high_freq: float = -400.0, num_filters: int = 80,
                 norm_filters: bool = False, torchaudio_compatible_mel_scale: bool = True):
        super().__init__(**_common(sampling_rate, frame_length, frame_shift, round_to_power_of_two, remove_dc_offset, preemph_coeff, window_type,
                                   dither, snip_edges, energy_floor, raw_energy, use_energy), use_fft_mag=use_fft_mag, low_freq=low_freq,
                         high_freq=high_freq, num_filters=num_filters, norm_filters=norm_filters,
                         torchaudio_compatible_mel_scale=torchaudio_compatible_mel_scale)


class HipWav2MFCC(_HipLayer):
    """MFCCs: log-mel, DCT, lifter.  ``use_energy=True`` replaces C0 by the log-energy (the reference raises a shape error
    there; this is what it documents)."""

    kind = _E.KIND_MFCC

    def __init__(self, sampling_rate: int = 16000, frame_length: Seconds = 0.025, frame_shift: Seconds = 0.01, round_to_power_of_two: bool = True,
                 remove_dc_offset: bool = True, preemph_coeff: float = 0.97, window_type: str = "povey", dither: float = 0.0,
                 snip_edges: bool = False, energy_floor: float = EPSILON, raw_energy: bool = True, use_energy: bool = False,
                 use_fft_mag: bool = False, low_freq: float = 20.0, high_freq: float = -400.0, num_filters: int = 23,
                 norm_filters: bool = False, num_ceps: int = 13, cepstral_lifter: int = 22, torchaudio_compatible_mel_scale: bool = True):
        super().__init__(**_common(sampling_rate, frame_length, frame_shift, round_to_power_of_two, remove_dc_offset, preemph_coeff, window_type,
                                   dither, snip_edges, energy_floor, raw_energy, use_energy), use_fft_mag=use_fft_mag, low_freq=low_freq,
                         high_freq=high_freq, num_filters=num_filters, norm_filters=norm_filters, num_ceps=num_ceps,
                         cepstral_lifter=cepstral_lifter, torchaudio_compatible_mel_scale=torchaudio_compatible_mel_scale)
