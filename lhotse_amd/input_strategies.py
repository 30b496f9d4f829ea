"""
``HipOnTheFlyFeatures`` -- lhotse's ``OnTheFlyFeatures`` input strategy (lhotse/dataset/input_strategies.py:351-476)
with the extraction + collation step fused on the GPU.

The reference computes a list of per-cut feature matrices and then ``collate_matrices(..., padding_value=LOG_EPSILON)``
copies each of them into a fresh padded tensor (lhotse/dataset/collation.py:506-535).  Here the kernels write every
cut straight into its slot of the ``(B, Tmax, F)`` tensor and only the padding rows are filled
(``hipfeat_extract_collated``); audio reading, wave transforms, ``return_audio`` / ``fault_tolerant`` outputs and the
supervision helpers are inherited unchanged.

Needs lhotse (it consumes ``CutSet``s); importing this module without lhotse works, constructing the class does not.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from .compat import HAVE_LHOTSE, LOG_EPSILON

if HAVE_LHOTSE:  # pragma: no cover - authoring container only
    from lhotse.dataset.collation import collate_vectors, read_audio_from_cuts  # type: ignore
    from lhotse.dataset.input_strategies import OnTheFlyFeatures, _get_executor  # type: ignore

    class HipOnTheFlyFeatures(OnTheFlyFeatures):
        """Same constructor as ``OnTheFlyFeatures`` plus ``return_device``: ``None`` keeps the padded feature
        tensor on the extractor's GPU (ready for the training step), ``"cpu"`` hands back a host tensor like the
        reference does."""

        def __init__(self, extractor, *args, return_device: Optional[Union[str, torch.device]] = None, **kwargs) -> None:
            if not hasattr(extractor, "extract_collated"):
                raise TypeError("HipOnTheFlyFeatures needs a Hip* extractor (with extract_collated)")
            super().__init__(extractor, *args, **kwargs)
            self.return_device = return_device

        def __call__(self, cuts, recording_field: Optional[str] = None):
            """Only the middle step differs from the parent: ``extract_batch`` + ``collate_matrices`` become ONE fused launch
            (``extract_collated``).  The parent offers no hook between reading the audio and collating the features, so the two
            ends of its pipeline are invoked here through the same public helpers it uses."""
            pool = _get_executor(self.num_workers, executor_type=self._executor_type)
            audios, cuts = read_audio_from_cuts(cuts, executor=pool, suppress_errors=self.fault_tolerant, recording_field=recording_field)
            for transform in self.wave_transforms:
                audios = [transform(a) for a in audios]
            rates = {c.sampling_rate for c in cuts}
            assert len(rates) == 1, f"one launch per batch needs a single sampling rate, got {sorted(rates)}"
            feats, feat_lens = self.extractor.extract_collated(audios, sampling_rate=rates.pop(), padding_value=LOG_EPSILON)
            result = [feats if self.return_device is None else feats.to(self.return_device), feat_lens]
            if self.return_audio:  # (B, Tmax) zero-padded samples + their lengths, as the parent returns them
                flat = [a.reshape(-1) for a in audios]
                result += [collate_vectors(flat, padding_value=0), torch.tensor([len(a) for a in flat], dtype=torch.int64)]
            if self.fault_tolerant:  # the cuts that survived audio loading
                result.append(cuts)
            return tuple(result)

else:

    class HipOnTheFlyFeatures:  # type: ignore[no-redef]
        def __init__(self, *args, **kwargs):
            raise ImportError("HipOnTheFlyFeatures consumes lhotse CutSets: install lhotse (lhotse.dataset.input_strategies.OnTheFlyFeatures)")
