"""
lhotse_amd -- MI355X-native batched audio feature extraction behind lhotse's
FeatureExtractor interface (see DESIGN.md / INTEGRATION.md).

Importing the package never touches the GPU; extractors create their device plan lazily.
"""
from .extractors import (  # noqa: F401
    HipFbank,
    HipFbankConfig,
    HipLogSpectrogram,
    HipLogSpectrogramConfig,
    HipMfcc,
    HipMfccConfig,
    HipSpectrogram,
    HipSpectrogramConfig,
)

from .augmentation import HipResample, HipResampleTensor, HipSpeed, HipSpeedBank, get_or_create_resampler  # noqa: F401,E402

from .kaldifeat import (  # noqa: F401,E402
    HipKaldifeatFbank,
    HipKaldifeatFbankConfig,
    HipKaldifeatFrameOptions,
    HipKaldifeatMelOptions,
    HipKaldifeatMfcc,
    HipKaldifeatMfccConfig,
)

from .input_strategies import HipOnTheFlyFeatures  # noqa: F401,E402

from .whisper import HipWhisperFbank, HipWhisperFbankConfig  # noqa: F401,E402

from .librosa_fbank import HipLibrosaFbank, HipLibrosaFbankConfig  # noqa: F401,E402

from .signal_transforms import HipGlobalMVN, HipSpecAugment  # noqa: F401,E402

from .layers import HipWav2LogFilterBank, HipWav2LogSpec, HipWav2MFCC, HipWav2Spec  # noqa: F401,E402

from .storage import HipArchiveF16Writer, HipArchiveReader, HipArchiveWriter, compute_and_store_features_batch  # noqa: F401,E402

from .sharding import compute_and_store_features_sharded  # noqa: F401,E402

__all__ = [
    "compute_and_store_features_sharded",
    "compute_and_store_features_batch",
    "HipArchiveWriter",
    "HipArchiveF16Writer",
    "HipArchiveReader",
    "HipWhisperFbank",
    "HipLibrosaFbank",
    "HipGlobalMVN",
    "HipWav2Spec",
    "HipWav2LogSpec",
    "HipWav2LogFilterBank",
    "HipWav2MFCC",
    "HipSpecAugment",
    "HipLibrosaFbankConfig",
    "HipWhisperFbankConfig",
    "HipOnTheFlyFeatures",
    "HipKaldifeatFbank",
    "HipKaldifeatFbankConfig",
    "HipKaldifeatMfcc",
    "HipKaldifeatMfccConfig",
    "HipKaldifeatFrameOptions",
    "HipKaldifeatMelOptions",
    "HipSpeed",
    "HipSpeedBank",
    "HipResample",
    "HipResampleTensor",
    "get_or_create_resampler",
    "HipFbank",
    "HipFbankConfig",
    "HipMfcc",
    "HipMfccConfig",
    "HipSpectrogram",
    "HipSpectrogramConfig",
    "HipLogSpectrogram",
    "HipLogSpectrogramConfig",
]
__version__ = "0.1.0"
