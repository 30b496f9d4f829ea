"""
kaldifeat-shaped extractors on the HIP path.

lhotse's GPU feature extraction today goes through ``KaldifeatFbank`` / ``KaldifeatMfcc``
(lhotse/features/kaldifeat.py:13-263), thin wrappers that hand a list of 1-D tensors to the external
``kaldifeat`` C++/CUDA package.  The classes below keep that option surface -- ``frame_opts`` / ``mel_opts``
dataclasses with the same fields, defaults and dict layout (``samp_freq``, ``frame_shift_ms`` ...), the same
``extract`` / ``extract_batch`` input and return conventions (:78-141) -- and compute the features with the same
kernels as ``HipFbank`` / ``HipMfcc``.

kaldifeat itself is not available offline, so numerics are pinned to lhotse's torch-native Kaldi layers (the
reference's own cross-check between the two is ``assert_almost_equal(decimal=3)``,
test/features/test_kaldifeat_features.py:103-116).  Every item is framed on its own (reflected edges per item,
as Kaldi does), i.e. ``edge_rule="reflect"``.

``htk_compat=True`` (energy / C0 column last; C0 times sqrt(2) when it is not the energy) and ``use_log_fbank=False`` (linear
mel energies) are applied on the device to the kernels' output, restating Kaldi's published ``feature-fbank.cc`` /
``feature-mfcc.cc`` (the code kaldifeat wraps; not available offline, so these two are "parity unpinned" like the rest of
this module).  ``mel_opts.htk_mode=True`` / ``mel_opts.debug_mel=True`` raise ``NotImplementedError`` at construction:
the HTK variant of the mel-bin edges cannot be restated from memory reliably, and nothing would pin it.
MFCC ``use_energy=True`` follows Kaldi (the log-energy replaces C0); lhotse's own torch-native layer crashes on it.
``vtln_low`` / ``vtln_high`` only act through a VTLN warp factor, which the lhotse wrapper never sets.
``chunk_size`` is accepted and ignored (one launch handles any batch).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from functools import partial
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .compat import EPSILON, FeatureExtractor, Seconds, register_extractor
from .extractors import HipFbank, HipFbankConfig, HipMfcc, HipMfccConfig, _log_mix


@dataclass
class HipKaldifeatFrameOptions:
    """lhotse/features/kaldifeat.py:13-41"""

    sampling_rate: int = 16000
    frame_shift: Seconds = 0.01
    frame_length: Seconds = 0.025
    dither: float = 0.0
    preemph_coeff: float = 0.97
    remove_dc_offset: bool = True
    window_type: str = "povey"
    round_to_power_of_two: bool = True
    blackman_coeff: float = 0.42
    snip_edges: bool = False

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["samp_freq"] = float(d.pop("sampling_rate"))
        d["frame_shift_ms"] = d.pop("frame_shift") * 1000.0
        d["frame_length_ms"] = d.pop("frame_length") * 1000.0
        return d

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipKaldifeatFrameOptions":
        data = data.copy()
        if "samp_freq" in data:
            data["sampling_rate"] = int(data.pop("samp_freq"))
        for key in ["frame_shift_ms", "frame_length_ms"]:
            if key in data:
                data[key.replace("_ms", "")] = data.pop(key) / 1000
        return HipKaldifeatFrameOptions(**data)


@dataclass
class HipKaldifeatMelOptions:
    """lhotse/features/kaldifeat.py:44-59"""

    num_bins: int = 80
    low_freq: float = 20.0
    high_freq: float = -400.0
    vtln_low: float = 100.0
    vtln_high: float = -500.0
    debug_mel: bool = False
    htk_mode: bool = False

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipKaldifeatMelOptions":
        return HipKaldifeatMelOptions(**data)


def _device_str(device) -> str:
    return str(device) if isinstance(device, torch.device) else device


@dataclass
class HipKaldifeatFbankConfig:
    """lhotse/features/kaldifeat.py:148-175"""

    frame_opts: HipKaldifeatFrameOptions = field(default_factory=HipKaldifeatFrameOptions)
    mel_opts: HipKaldifeatMelOptions = field(default_factory=HipKaldifeatMelOptions)
    use_energy: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    htk_compat: bool = False
    use_log_fbank: bool = True
    use_power: bool = True
    device: Union[str, torch.device] = "cuda"
    chunk_size: Optional[int] = 100 * 60 * 20

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["frame_opts"] = self.frame_opts.to_dict()
        d["mel_opts"] = self.mel_opts.to_dict()
        d["device"] = _device_str(self.device)
        return d

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipKaldifeatFbankConfig":
        data = dict(data)
        frame_opts = HipKaldifeatFrameOptions.from_dict(data.pop("frame_opts"))
        mel_opts = HipKaldifeatMelOptions.from_dict(data.pop("mel_opts"))
        return HipKaldifeatFbankConfig(frame_opts=frame_opts, mel_opts=mel_opts, **data)


@dataclass
class HipKaldifeatMfccConfig:
    """lhotse/features/kaldifeat.py:217-246"""

    frame_opts: HipKaldifeatFrameOptions = field(default_factory=HipKaldifeatFrameOptions)
    mel_opts: HipKaldifeatMelOptions = field(default_factory=partial(HipKaldifeatMelOptions, num_bins=23))
    num_ceps: int = 13
    use_energy: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    cepstral_lifter: float = 22.0
    htk_compat: bool = False
    device: Union[str, torch.device] = "cuda"
    chunk_size: Optional[int] = 1000

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["frame_opts"] = self.frame_opts.to_dict()
        d["mel_opts"] = self.mel_opts.to_dict()
        d["device"] = _device_str(self.device)
        return d

    @staticmethod
    def from_dict(data: Dict[str, Any]) -> "HipKaldifeatMfccConfig":
        data = dict(data)
        frame_opts = HipKaldifeatFrameOptions.from_dict(data.pop("frame_opts"))
        mel_opts = HipKaldifeatMelOptions.from_dict(data.pop("mel_opts"))
        return HipKaldifeatMfccConfig(frame_opts=frame_opts, mel_opts=mel_opts, **data)


def _frame_kwargs(fo: HipKaldifeatFrameOptions, mo: HipKaldifeatMelOptions, what: str) -> Dict[str, Any]:
    if mo.htk_mode or mo.debug_mel:
        raise NotImplementedError(f"{what}: mel_opts.htk_mode / debug_mel are not supported by the HIP kernels")
    return dict(
        sampling_rate=int(fo.sampling_rate),
        frame_length=fo.frame_length,
        frame_shift=fo.frame_shift,
        round_to_power_of_two=fo.round_to_power_of_two,
        remove_dc_offset=fo.remove_dc_offset,
        preemph_coeff=fo.preemph_coeff,
        window_type=fo.window_type,
        dither=fo.dither,
        snip_edges=fo.snip_edges,
        low_freq=mo.low_freq,
        high_freq=mo.high_freq,
        num_filters=mo.num_bins,
        edge_rule="reflect",
    )


class _HipKaldifeatExtractor(FeatureExtractor):
    """Shared input/return conventions of ``KaldifeatExtractor`` (lhotse/features/kaldifeat.py:62-145)."""

    def __init__(self, config: Optional[Any] = None) -> None:
        super().__init__(config=config)
        self._inner = None

    def _make_inner(self):
        raise NotImplementedError

    @property
    def inner(self):
        if self._inner is None:
            self._inner = self._make_inner()
        return self._inner

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_inner"] = None  # device handles are per process
        return st

    @property
    def device(self) -> Union[str, torch.device]:
        return self.config.device

    @property
    def frame_shift(self) -> Seconds:
        return self.config.frame_opts.frame_shift

    @property
    def kernel_name(self) -> str:
        return self.inner.kernel_name

    def _post(self, packed: torch.Tensor) -> torch.Tensor:
        """Option handling on the packed (sum T, F) device tensor (column order, linear energies); identity by default."""
        return packed

    def extract_batch(self, samples, sampling_rate: int, lengths=None):
        # kaldifeat expects a list of 1-D tensors (kaldifeat.py:91-93)
        if lengths is not None:
            samples = [x[:l] for x, l in zip(samples, lengths)]
        return self.extract(samples=samples, sampling_rate=sampling_rate)

    def extract(self, samples, sampling_rate: int):
        expected_sr = self.config.frame_opts.sampling_rate
        assert sampling_rate == expected_sr, f"Mismatched sampling rate: extractor expects {expected_sr}, got {sampling_rate}"
        input_is_list = False
        if isinstance(samples, list):
            input_is_list = True
            items = list(samples)
        elif samples.ndim > 1:
            items = list(samples)
        else:
            items = [samples]
        as_numpy = any(isinstance(x, np.ndarray) for x in items)  # any numpy item -> numpy results (kaldifeat.py:124-127)
        items = [torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in items]
        items = [x.squeeze() if x.ndim == 2 else x for x in items]
        for x in items:
            if x.dtype != torch.float32:
                raise TypeError(f"extract(): expected float32 samples, got {x.dtype}")
        inner = self.inner
        with torch.no_grad():
            packed, frames = inner._extract_items(items)
            packed = self._post(packed)
            if as_numpy:
                packed = inner._to_host(packed).numpy()
        bounds = np.concatenate([[0], np.cumsum(frames)])
        result = [packed[int(bounds[i]) : int(bounds[i + 1])] for i in range(len(frames))]
        if len(result) == 1:
            return [result[0]] if input_is_list else result[0]
        if all(r.shape == result[0].shape for r in result[1:]):
            return packed.reshape(len(result), *result[0].shape)
        return result


@register_extractor
class HipKaldifeatFbank(_HipKaldifeatExtractor):
    """Drop-in for ``KaldifeatFbank`` (lhotse/features/kaldifeat.py:178-214)."""

    name = "hip-kaldifeat-fbank"
    config_type = HipKaldifeatFbankConfig

    def _make_inner(self) -> HipFbank:
        c = self.config
        kw = _frame_kwargs(c.frame_opts, c.mel_opts, self.name)
        inner = HipFbank(HipFbankConfig(use_energy=c.use_energy, energy_floor=c.energy_floor, raw_energy=c.raw_energy,
                                         use_fft_mag=not c.use_power, device=_device_str(c.device), **kw))
        inner.config.blackman_coeff = c.frame_opts.blackman_coeff
        return inner

    def _post(self, packed: torch.Tensor) -> torch.Tensor:
        """Kaldi's FbankComputer::Compute (feature-fbank.cc): without ``use_log_fbank`` the mel energies stay linear (the kernel's
        floored log is undone: energies below 1.19e-7 come back as that floor); with ``htk_compat`` the log-energy column, if
        any, goes last instead of first."""
        c = self.config
        e = 1 if c.use_energy else 0  # the kernels put the log-energy in column 0 (layers.py:575-576)
        if not c.use_log_fbank:
            packed = packed.clone()
            packed[:, e:] = torch.exp(packed[:, e:])
        if c.htk_compat and e:
            packed = torch.cat([packed[:, 1:], packed[:, :1]], dim=1)
        return packed

    def feature_dim(self, sampling_rate: int) -> int:
        return self.config.mel_opts.num_bins

    @staticmethod
    def mix(features_a: np.ndarray, features_b: np.ndarray, energy_scaling_factor_b: float) -> np.ndarray:
        return _log_mix(features_a, features_b, energy_scaling_factor_b)

    @staticmethod
    def compute_energy(features: np.ndarray) -> float:
        return float(np.sum(np.exp(features)))

    @staticmethod
    def scale(features: np.ndarray, energy_scaling_factor: float) -> np.ndarray:
        return features + np.log(energy_scaling_factor)


@register_extractor
class HipKaldifeatMfcc(_HipKaldifeatExtractor):
    """Drop-in for ``KaldifeatMfcc`` (lhotse/features/kaldifeat.py:249-263)."""

    name = "hip-kaldifeat-mfcc"
    config_type = HipKaldifeatMfccConfig

    def _make_inner(self) -> HipMfcc:
        c = self.config
        kw = _frame_kwargs(c.frame_opts, c.mel_opts, self.name)
        inner = HipMfcc(HipMfccConfig(use_energy=c.use_energy, energy_floor=c.energy_floor, raw_energy=c.raw_energy, num_ceps=c.num_ceps,
                                       cepstral_lifter=c.cepstral_lifter, device=_device_str(c.device), **kw))
        inner.config.blackman_coeff = c.frame_opts.blackman_coeff
        return inner

    def _post(self, packed: torch.Tensor) -> torch.Tensor:
        """Kaldi's MfccComputer::Compute (feature-mfcc.cc) with ``htk_compat``: the energy / C0 column goes last, and a C0 that
        is a cepstral coefficient (not the log-energy) is scaled by sqrt(2)."""
        c = self.config
        if not c.htk_compat:
            return packed
        first = packed[:, :1] if c.use_energy else packed[:, :1] * (2.0 ** 0.5)
        return torch.cat([packed[:, 1:], first], dim=1)

    def feature_dim(self, sampling_rate: int) -> int:
        return self.config.num_ceps
