"""
``HipLibrosaFbank`` -- librosa-style log-mel features on the HIP path.

Drop-in for ``LibrosaFbank`` (lhotse/features/librosa_fbank.py:139-180): same config fields and defaults
(22.05 kHz, n_fft 1024, hop 256, periodic "hann" window of ``win_length`` samples centred in the FFT frame,
80 slaney mel filters between ``fmin`` and ``fmax``), the same arithmetic -- ``librosa.stft(center=True,
pad_mode="reflect")``, ``|X|``, ``log10(max(1e-10, |X| @ mel.T))`` (:111-127) -- and the same number of rows,
``compute_num_frames`` = ``(S + hop // 2) // hop`` (:129-134; the STFT always yields at least that many, so the
reference's ``pad_or_truncate_features`` only ever truncates for cuts longer than half a window).

The arithmetic runs in libhipfeat (kind ``HIPFEAT_LIBROSA_FBANK``): ``wave_kernel`` for power-of-two FFT sizes
(512 / 1024 / 2048), ``generic_kernel`` otherwise.  Unlike the reference, no librosa is needed, and
``extract_batch`` runs the whole batch in one launch (the reference inherits the per-item loop of
``FeatureExtractor.extract_batch``, lhotse/features/base.py:152-222; same return conventions).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from .compat import EPSILON, Seconds, asdict_nonull, register_extractor
from .extractors import KIND_LIBROSA_FBANK, _HipExtractor, _log_mix


@dataclass
class HipLibrosaFbankConfig:
    # the fields and defaults of LibrosaFbankConfig (librosa_fbank.py:22-40), plus the device
    sampling_rate: int = 22050
    fft_size: int = 1024
    hop_size: int = 256
    win_length: Optional[int] = None
    window: str = "hann"
    num_mel_bins: int = 80
    fmin: int = 80
    fmax: int = 7600
    device: str = "cuda"

    def to_dict(self) -> Dict[str, Any]:
        return asdict_nonull(self)

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipLibrosaFbankConfig":
        return HipLibrosaFbankConfig(**data)


class _LibrosaPlanConfig:
    """What ``_Plan`` reads for kind ``HIPFEAT_LIBROSA_FBANK``."""

    remove_dc_offset = False
    preemph_coeff = 0.0
    dither = 0.0
    snip_edges = False
    energy_floor = EPSILON
    raw_energy = True
    use_energy = False
    use_fft_mag = True  # np.abs(x_stft), librosa_fbank.py:119

    def __init__(self, c: HipLibrosaFbankConfig):
        self.sampling_rate = int(c.sampling_rate)
        self.fft_size = int(c.fft_size)
        self.hop_size = int(c.hop_size)
        self.win_length = None if c.win_length is None else int(c.win_length)
        self.window = c.window
        self.num_filters = int(c.num_mel_bins)
        self.fmin = 0 if c.fmin is None else c.fmin  # librosa_fbank.py:121
        self.fmax = c.fmax  # None -> sampling_rate / 2 (:122)
        if self.win_length is not None and not 0 < self.win_length <= self.fft_size:
            raise ValueError(f"win_length={self.win_length} must be in (0, fft_size={self.fft_size}]")


@register_extractor
class HipLibrosaFbank(_HipExtractor):
    name = "hip-librosa-fbank"
    config_type = HipLibrosaFbankConfig
    kind = KIND_LIBROSA_FBANK

    def _plan_config(self):
        return _LibrosaPlanConfig(self.config)

    def _plan_mel_floor(self) -> float:
        return EPSILON  # np.maximum(eps, ...), librosa_fbank.py:127

    @property
    def frame_shift(self) -> Seconds:
        return self.config.hop_size / self.config.sampling_rate

    def feature_dim(self, sampling_rate: int) -> int:
        return self.config.num_mel_bins

    def extract(self, samples: Union[np.ndarray, torch.Tensor], sampling_rate: int) -> Union[np.ndarray, torch.Tensor]:
        if getattr(samples, "ndim", 1) == 2 and samples.shape[0] > 1:  # librosa_fbank.py:101-109
            raise AssertionError(f"LibrosaFbank works only with single-channel recordings (shape: {tuple(samples.shape)})")
        return super().extract(samples, sampling_rate)

    @staticmethod
    def mix(features_a: np.ndarray, features_b: np.ndarray, energy_scaling_factor_b: float) -> np.ndarray:
        return _log_mix(features_a, features_b, energy_scaling_factor_b)

    @staticmethod
    def compute_energy(features: np.ndarray) -> float:
        return float(np.sum(np.exp(features)))

    @staticmethod
    def scale(features: np.ndarray, energy_scaling_factor: float) -> np.ndarray:
        return features + np.log(energy_scaling_factor)
