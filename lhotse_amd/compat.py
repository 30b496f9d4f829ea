"""
Binding to lhotse's plugin interface for feature extractors.

When ``lhotse`` is importable the Hip* extractors subclass the real
``lhotse.features.base.FeatureExtractor`` and are registered with ``@register_extractor``
(lhotse/features/base.py:37-405), which is all that is needed for
``CutSet.compute_and_store_features[_batch]``, ``OnTheFlyFeatures``, the YAML round trip
and the ``lhotse feat`` CLI to accept them.

On machines without lhotse (e.g. the benchmark GPU box) a minimal stand-in with the same
surface is used so that the extractors, their configs and the (de)serialisation logic
behave identically.  The stand-in mirrors -- it does not re-implement -- the interface:
names, argument meaning and return conventions follow lhotse/features/base.py:64-95,
:152-222, :338-405.
"""
from __future__ import annotations

from abc import ABCMeta, abstractmethod
from dataclasses import asdict, is_dataclass
from typing import Any, Dict, Optional, Type

import numpy as np

Seconds = float
EPSILON = 1e-10  # lhotse/utils.py:49
LOG_EPSILON = -23.025850929940457  # math.log(EPSILON), lhotse/utils.py:50-51

try:  # pragma: no cover - exercised in the authoring container only
    from lhotse.features.base import (  # type: ignore
        FEATURE_EXTRACTORS,
        FeatureExtractor,
        get_extractor_type,
        register_extractor,
    )
    from lhotse.utils import compute_num_frames_from_samples  # type: ignore

    HAVE_LHOTSE = True
except Exception:  # ImportError, or a half-installed lhotse missing its own deps
    HAVE_LHOTSE = False

    FEATURE_EXTRACTORS: Dict[str, Type] = {}

    def register_extractor(cls):
        FEATURE_EXTRACTORS[cls.name] = cls
        return cls

    def get_extractor_type(name: str) -> Type:
        return FEATURE_EXTRACTORS[name]

    def compute_num_frames_from_samples(num_samples: int, frame_shift: Seconds, sampling_rate: int) -> int:
        window_hop = round(frame_shift * sampling_rate)
        return int((num_samples + window_hop // 2) // window_hop)

    class FeatureExtractor(metaclass=ABCMeta):
        """Stand-in for lhotse.features.base.FeatureExtractor (same public surface)."""

        name = None
        config_type = None

        def __init__(self, config: Optional[Any] = None):
            if config is None:
                config = self.config_type()
            assert is_dataclass(config), "The feature configuration object must be a dataclass."
            self.config = config

        @abstractmethod
        def extract(self, samples, sampling_rate: int):
            ...

        @property
        @abstractmethod
        def frame_shift(self) -> Seconds:
            ...

        @abstractmethod
        def feature_dim(self, sampling_rate: int) -> int:
            ...

        @property
        def device(self):
            return "cpu"

        @staticmethod
        def mix(features_a, features_b, energy_scaling_factor_b: float):
            raise ValueError('The feature extractor\'s "mix" operation is undefined.')

        @staticmethod
        def compute_energy(features) -> float:
            raise ValueError('The feature extractor\'s "compute_energy" operation is undefined.')

        @staticmethod
        def scale(features, energy_scaling_factor: float):
            raise ValueError('The feature extractor\'s "scale" operation is undefined.')

        @classmethod
        def from_dict(cls, data: dict) -> "FeatureExtractor":
            data = dict(data)
            feature_type = data.pop("feature_type")
            extractor_type = get_extractor_type(feature_type)
            return extractor_type(extractor_type.config_type.from_dict(data))

        def to_dict(self) -> Dict[str, Any]:
            d = self.config.to_dict()
            d["feature_type"] = self.name
            return d

        @classmethod
        def from_yaml(cls, path) -> "FeatureExtractor":
            import yaml

            with open(path) as f:
                return cls.from_dict(yaml.safe_load(f))

        def to_yaml(self, path):
            import torch
            import yaml

            data = self.to_dict()
            if "device" in data and isinstance(data["device"], torch.device):
                data["device"] = data["device"].type
            with open(path, "w") as f:
                yaml.safe_dump(data, f)


def asdict_nonull(dclass) -> Dict[str, Any]:
    """dataclass -> dict without None values (as lhotse.utils.asdict_nonull)."""
    return {k: v for k, v in asdict(dclass).items() if v is not None}
