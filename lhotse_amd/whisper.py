"""
``HipWhisperFbank`` -- Whisper's log-mel front end on the HIP path.

Drop-in for ``WhisperFbank`` (lhotse/features/whisper_fbank.py:88-185): same config (``num_filters``, ``device``),
16 kHz / n_fft 400 / hop 160 / periodic hann window / slaney mel filterbank / log10 with an 8-decade dynamic-range
clamp under the utterance maximum / ``(x + 4) / 4`` / zero row up to ``compute_num_frames_from_samples``.
The arithmetic runs in ``generic_kernel`` (centred "reflect" framing, 400-point direct DFT, banded mel, log) and
``whisper_norm_kernel`` (per-cut max, clamp, affine, padding row) of libhipfeat (kind ``HIPFEAT_WHISPER``).

Unlike the reference, no librosa is needed (the filterbank formula is evaluated in ``constants.make_slaney_mel``),
and ``extract_batch`` runs the whole batch in one launch (the reference inherits the per-item loop of
``FeatureExtractor.extract_batch``, lhotse/features/base.py:152-222; same return conventions).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from .compat import EPSILON, Seconds, asdict_nonull, register_extractor
from .extractors import KIND_WHISPER, _HipExtractor, _log_mix


@dataclass
class HipWhisperFbankConfig:
    num_filters: int = 80
    device: str = "cuda"

    def to_dict(self) -> Dict[str, Any]:
        return asdict_nonull(self)

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipWhisperFbankConfig":
        return HipWhisperFbankConfig(**data)


class _WhisperPlanConfig:
    """What ``_Plan`` reads, for the fixed Whisper front end (whisper_fbank.py:96-120)."""

    sampling_rate = 16000
    frame_length = 400 / 16000
    frame_shift = 160 / 16000
    round_to_power_of_two = False
    remove_dc_offset = False
    preemph_coeff = 0.0
    window_type = "hann_periodic"
    dither = 0.0
    snip_edges = False
    energy_floor = EPSILON
    raw_energy = True
    use_energy = False
    use_fft_mag = False

    def __init__(self, num_filters: int):
        self.num_filters = int(num_filters)


@register_extractor
class HipWhisperFbank(_HipExtractor):
    name = "hip-whisper-fbank"
    config_type = HipWhisperFbankConfig
    kind = KIND_WHISPER

    def __init__(self, config: Optional[HipWhisperFbankConfig] = None):
        super().__init__(config=config)
        self.sampling_rate = 16000
        self.hop_length = 160
        self.n_fft = 400
        self.num_filters = self.config.num_filters

    def _plan_config(self):
        return _WhisperPlanConfig(self.config.num_filters)

    def _plan_mel_floor(self) -> float:
        return 1e-10  # torch.clamp(mel_spec, min=1e-10), whisper_fbank.py:67

    @property
    def frame_shift(self) -> Seconds:
        return self.hop_length / self.sampling_rate

    def feature_dim(self, sampling_rate: int) -> int:
        return self.num_filters

    def _check_sr(self, sampling_rate: int):
        assert sampling_rate == self.sampling_rate, (
            f"Fbank was instantiated for sampling_rate {self.sampling_rate}, but sampling_rate={sampling_rate} was passed to extract(). "
            "Note you can use CutSet/RecordingSet.resample() to change the audio sampling rate."
        )

    def extract(self, samples: Union[np.ndarray, torch.Tensor], sampling_rate: int) -> Union[np.ndarray, torch.Tensor]:
        if getattr(samples, "ndim", 1) == 2 and samples.shape[0] > 1:
            raise ValueError("Whisper Fbank works only with single-channel recordings.")  # whisper_fbank.py:54-56
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
