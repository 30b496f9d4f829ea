"""
``HipOnTheFlyFeatures`` -- lhotse's ``OnTheFlyFeatures`` input strategy (lhotse/dataset/input_strategies.py:351-476)
with the extraction + collation step fused on the GPU and, since round 3, the speed
perturbation of ``PerturbSpeed``-ed cuts moved there too.

The reference computes a list of per-cut feature matrices and then ``collate_matrices(..., padding_value=LOG_EPSILON)``
copies each of them into a fresh padded tensor (lhotse/dataset/collation.py:506-535).  Here the kernels write every
cut straight into its slot of the ``(B, Tmax, F)`` tensor and only the padding rows are filled
(``hipfeat_extract_collated``); audio reading, wave transforms, ``return_audio`` / ``fault_tolerant`` outputs and the
supervision helpers are inherited unchanged.

Speed perturbation.  ``PerturbSpeed`` (lhotse/dataset/cut_transforms/perturb_speed.py:8-47) turns a cut into one whose
recording carries a ``Speed(factor)`` transform; the resampling itself then happens on the CPU inside
``Recording.load_audio`` (lhotse/audio/recording.py:431-490), one torch ``conv1d`` per cut, before the strategy ever
sees the samples -- it is what an on-the-fly pipeline spends its time in once the features are on the GPU.  With
``gpu_speed_perturb=True`` (default) a cut whose recording's ONLY transform is that ``Speed`` is read WITHOUT it -- the
very segment of the original file ``load_audio`` would read, through the same ``reverse_timestamps`` /
``AudioSource.load_audio`` calls -- and the batch is perturbed on the device, mixed factors and all, in the arena the
feature launch reads from (``lhotse_amd.augmentation.perturb_speed_in_arena``).  Sample counts follow
``assert_and_maybe_fix_num_samples`` (truncate; the rare cut that would need reflect-padding takes the reference's own
path), values agree with the CPU ``Speed`` to the resampler's 1e-5.  Every other cut (mixed cuts, other or several
transforms, multi-channel) is loaded exactly as before.

Needs lhotse (it consumes ``CutSet``s); importing this module without lhotse works, constructing the class does not.
"""
from __future__ import annotations

from math import gcd, isclose
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .compat import HAVE_LHOTSE, LOG_EPSILON


BANK_RATIOS = ((9, 10), (11, 10))  # orig : new of the mixed launch's compile-time resamplers (hipfeat_speed_bank_create) = speed 0.9 / 1.1


def _bank_ratio_ok(sampling_rate: int, factor: float) -> bool:
    """Speed(factor) resamples round(sr * factor) -> sr (lhotse/augmentation/torchaudio.py:37-42); the mixed launch serves the ratios
    of BANK_RATIOS after reduction by the gcd (resample.py:219-222)."""
    import math

    orig, new = int(round(sampling_rate * factor)), int(sampling_rate)
    g = math.gcd(orig, new)
    return g > 0 and (orig // g, new // g) in BANK_RATIOS


def _perturb_in_arena(arena, offsets, lengths, factors, sampling_rate, tail_start):
    """(indirection for the CPU stand-in of the tests)"""
    from .augmentation import perturb_speed_in_arena

    return perturb_speed_in_arena(arena, offsets, lengths, factors, sampling_rate, tail_start)


class FusedMiniBatch:
    """The device-facing half of ``HipOnTheFlyFeatures`` -- packing a (partly speed-perturbed) mini-batch into one arena, the launch pair
    of ``hipfeat_minibatch_*`` (or the per-factor route), the collated feature tensor -- WITHOUT any lhotse type in its interface:
    plain waveforms, factors and sample counts in, tensors out.  ``HipOnTheFlyFeatures`` inherits these methods unchanged; on a machine
    without lhotse (the GPU box of the test suite) the class is usable on its own, so that the GPU tests drive the product's code and not
    a restatement of it (tests/test_gpu_reference_drivers.py holds it to what lhotse's K2SpeechRecognitionDataset returned with the
    reference's CPU Speed + Fbank on the same files)."""

    def __init__(self, extractor, return_audio: bool = False) -> None:
        if not hasattr(extractor, "extract_collated"):
            raise TypeError("FusedMiniBatch needs a Hip* extractor (with extract_collated)")
        self.extractor = extractor
        self.return_audio = return_audio

    def features_of(self, audios: List[torch.Tensor], factors: List[float], wants: List[int], sampling_rate: int):
        """What ``HipOnTheFlyFeatures.__call__`` does between reading the audio and returning: ``audios`` as read from the files (the
        segments in front of a pending ``Speed``), ``factors`` still to be applied (1.0 = none), ``wants`` = samples each cut must end up
        with (``compute_num_samples(cut.duration)``) -> ``(feats (B, Tmax, F) on the extractor's device, feat_lens)``."""
        if any(f != 1.0 for f in factors):
            feats, feat_lens, _ = self._perturb_and_extract(audios, factors, wants, sampling_rate)
            return feats, feat_lens
        return self.extractor.extract_collated(audios, sampling_rate=sampling_rate, padding_value=LOG_EPSILON)

    def _speed_bank(self, factors, sr: int, device):
        """The bank of the factors met so far on this device (rebuilt when a new factor shows up); None if one of them is not among
        the mixed launch's ratios."""
        from ._lib import ERR_UNSUPPORTED, HipFeatError
        from .augmentation import HipSpeedBank

        need = {float(f) for f in factors if float(f) != 1.0}
        cache = self.__dict__.setdefault("_banks", {})
        key = (int(sr), str(device))
        refused = cache.setdefault("refused", set())
        if need & refused:
            return None
        bank = cache.get(key)
        if bank is None or not need <= set(bank.factors):
            have = set() if bank is None else set(bank.factors)
            # only the factors whose resampling ratio round(sr * f) : sr reduces to one of the mixed launch's compile-time ratios can be
            # served by a bank; every other factor is refused ON ITS OWN (a mini-batch with {0.9, 0.95} used to blacklist 0.9
            # as well, and every later 0.9 / 1.1 mini-batch silently fell back to the three-launch route)
            bad = {f for f in need - have if not _bank_ratio_ok(sr, f)}
            if bad:
                refused |= bad
                return None
            try:
                bank = cache[key] = HipSpeedBank(sorted(have | need), sr, device)
            except HipFeatError as e:
                if e.status != ERR_UNSUPPORTED:
                    raise
                return None  # (this mini-batch goes per factor; nothing is blacklisted on a guess)
        return bank

    def _perturb_and_extract(self, audios: List[torch.Tensor], factors: List[float], wants: List[int], sr: int):
        """Pack the (partly unperturbed) batch, resample the cuts with a pending factor into the tail of the same buffer, extract."""
        from .augmentation import perturbed_tail_floats
        from .extractors import _as_1d_float

        ex = self.extractor
        ex._check_sr(sr)
        items = [_as_1d_float(a.squeeze() if a.ndim > 1 else a, "HipOnTheFlyFeatures") for a in audios]
        with torch.no_grad():
            packed, offs, lens = ex._pack(items)
            front = int(packed.numel())
            arena = torch.empty(((front + 3) & ~3) + perturbed_tail_floats(lens, factors, sr), dtype=torch.float32, device=packed.device)
            arena[:front].copy_(packed, non_blocking=True)
            zero_pad = getattr(ex.config, "edge_rule", "reflect") == "batch_zero_pad"  # as extract_collated
            want = np.ascontiguousarray(wants, dtype=np.int64)
            bank = self._speed_bank(factors, sr, packed.device) if hasattr(ex.plan, "handle") else None
            if bank is not None:  # ONE launch for all factors + the padding rows, then the feature launch (hipfeat_minibatch_*)
                feats, frames, po, pl = bank.extract_collated(ex.plan, arena, np.ascontiguousarray(offs, dtype=np.int64),
                                                              np.ascontiguousarray(lens, dtype=np.int64), bank.index_of(factors), front,
                                                              float(LOG_EPSILON), max_samples=want, zero_pad_batch=zero_pad)
            else:  # factors outside the mixed launch's compile-time ratios: one resample launch per factor
                po, pl = _perturb_in_arena(arena, offs, lens, factors, sr, front)
                pl = np.minimum(pl, want)  # a sample or two to truncate (recording.py:1058-1060)
                padded = np.full(len(pl), int(pl.max()), dtype=np.int64) if zero_pad else None
                feats, frames = ex.plan.run_collated(arena, po, pl, padded, float(LOG_EPSILON))
        perturbed = None
        if self.return_audio:
            perturbed = [arena[int(o) : int(o) + int(n)].cpu() for o, n in zip(po, pl)]
        return feats, torch.from_numpy(np.asarray(frames, dtype=np.int64)), perturbed


if HAVE_LHOTSE:  # pragma: no cover - authoring container only
    from lhotse.audio.utils import suppress_audio_loading_errors  # type: ignore
    from lhotse.dataset.collation import collate_vectors, read_audio_from_cuts  # type: ignore
    from lhotse.dataset.input_strategies import OnTheFlyFeatures, _get_executor  # type: ignore
    from lhotse.utils import compute_num_samples  # type: ignore

    def deferred_speed_factor(cut) -> Optional[float]:
        """The factor of the cut's ``Speed`` transform if that is all that stands between the file and the samples (a mono cut over
        a recording whose transform list is exactly ``[Speed(factor)]``, no video); None = load it the reference's way."""
        if type(cut).__name__ != "MonoCut" or not cut.has_recording:
            return None
        rec = cut.recording
        tf = rec.transforms
        if not tf or len(tf) != 1 or getattr(rec, "has_video", False):
            return None
        t = tf[0]
        name = t.get("name") if isinstance(t, dict) else type(t).__name__
        if name != "Speed":
            return None
        factor = t["kwargs"]["factor"] if isinstance(t, dict) else t.factor
        return float(factor)

    def read_unperturbed(cut, factor: float) -> np.ndarray:
        """The segment of the ORIGINAL audio that ``Recording.load_audio`` reads for this cut before it applies ``Speed(factor)``
        (lhotse/audio/recording.py:412-467): same backward pass over the timestamps, same per-source reads."""
        from lhotse.augmentation import Speed  # the reference's own class: its reverse_timestamps is the contract

        rec = cut.recording
        offset, duration = cut.start, cut.duration
        if duration is not None and isclose(duration, rec.duration, abs_tol=1e-3):
            duration = None  # (recording.py:415-417)
        offset_aug, duration_aug = Speed(factor=factor).reverse_timestamps(offset=offset, duration=duration, sampling_rate=rec.sampling_rate)
        per_source = []
        for source in rec.sources:
            if cut.channel not in source.channels:
                continue
            samples = source.load_audio(offset=offset_aug, duration=duration_aug, force_opus_sampling_rate=rec.sampling_rate)
            drop = [i for i, cid in enumerate(source.channels) if cid != cut.channel]
            if drop:
                samples = np.delete(samples, drop, axis=0)
            per_source.append(samples)
        audio = rec._stack_audio_channels(per_source)
        return np.ascontiguousarray(audio.reshape(-1), dtype=np.float32)

    def _read_one(cut, gpu_speed: bool, suppress_errors: bool) -> Optional[Tuple[torch.Tensor, float, int]]:
        """(samples, factor still to be applied, samples the cut must end up with) or None when the read failed and errors are suppressed."""
        with suppress_audio_loading_errors(enabled=suppress_errors):
            factor = deferred_speed_factor(cut) if gpu_speed else None
            if factor is not None and factor != 1.0:
                raw = read_unperturbed(cut, factor)
                want = compute_num_samples(cut.duration, cut.sampling_rate)
                src, dst = round(cut.sampling_rate * factor), cut.sampling_rate
                g = gcd(src, dst)
                got = int(np.ceil(np.float32((dst // g) * len(raw) / (src // g))))  # resample.py:309
                if got >= want:  # the usual case: equal, or a sample or two to truncate (assert_and_maybe_fix_num_samples, recording.py:1032-1070)
                    return torch.from_numpy(raw), factor, want
                # (the cut would need reflect-padding: the reference's own path)
            audio = cut.load_audio()
            if audio.shape[0] == 1:
                audio = audio.squeeze(0)  # collapse channel dim if mono (collation.py:674-675)
            return torch.from_numpy(audio), 1.0, int(audio.shape[-1])
        return None

    class HipOnTheFlyFeatures(OnTheFlyFeatures, FusedMiniBatch):
        """Same constructor as ``OnTheFlyFeatures`` plus ``return_device`` (``None`` keeps the padded feature tensor on the
        extractor's GPU, ready for the training step; ``"cpu"`` hands back a host tensor like the reference does) and
        ``gpu_speed_perturb`` (see the module docstring)."""

        def __init__(self, extractor, *args, return_device: Optional[Union[str, torch.device]] = None,
                     gpu_speed_perturb: Optional[bool] = None, **kwargs) -> None:
            if not hasattr(extractor, "extract_collated"):
                raise TypeError("HipOnTheFlyFeatures needs a Hip* extractor (with extract_collated)")
            super().__init__(extractor, *args, **kwargs)
            self.return_device = return_device
            # None (default) = on the device unless `wave_transforms` are given: those run on the LOADED samples, i.e. after the
            # Speed that Recording.load_audio applies, so with them the reference order (CPU Speed inside load_audio, then the
            # transforms) is kept and the strategy behaves exactly like OnTheFlyFeatures.  An explicit True together with
            # wave_transforms is a contradiction and raises when a perturbed cut is met.
            self._gpu_speed_explicit = gpu_speed_perturb is not None
            self.gpu_speed_perturb = (not self.wave_transforms) if gpu_speed_perturb is None else bool(gpu_speed_perturb)

        def _read(self, cuts, pool, recording_field):
            """read_audio_from_cuts (lhotse/dataset/collation.py:541-600) with the Speed of eligible cuts left for the device."""
            if recording_field is not None or not self.gpu_speed_perturb or not any(deferred_speed_factor(c) not in (None, 1.0) for c in cuts):
                audios, ok = read_audio_from_cuts(cuts, executor=pool, suppress_errors=self.fault_tolerant, recording_field=recording_field)
                return audios, [1.0] * len(audios), [int(a.shape[-1]) for a in audios], ok
            from functools import partial

            from lhotse import CutSet

            cuts = list(cuts)
            map_fn = map if pool is None else pool.map
            audios, factors, wants, ok = [], [], [], []
            for cut, res in zip(cuts, map_fn(partial(_read_one, gpu_speed=True, suppress_errors=self.fault_tolerant), cuts)):
                if res is None:
                    continue
                audios.append(res[0]), factors.append(res[1]), wants.append(res[2]), ok.append(cut)
            return audios, factors, wants, CutSet.from_cuts(ok)

        def __call__(self, cuts, recording_field: Optional[str] = None):
            """Only the middle of the parent's pipeline differs: ``extract_batch`` + ``collate_matrices`` become ONE fused launch, and
            pending speed factors are applied to the packed batch on the device in front of it.  The parent offers no hook between
            reading the audio and collating the features, so the two ends of its pipeline are invoked here through the same public
            helpers it uses."""
            pool = _get_executor(self.num_workers, executor_type=self._executor_type)
            audios, factors, wants, cuts = self._read(cuts, pool, recording_field)
            for transform in self.wave_transforms:
                if any(f != 1.0 for f in factors):
                    raise ValueError("gpu_speed_perturb=True was requested together with wave_transforms: the transforms run on the loaded "
                                     "samples, before the device applies the pending speed factors; leave gpu_speed_perturb at its default "
                                     "(None: the reference's CPU Speed whenever wave_transforms are given) or pass False")
                audios = [transform(a) for a in audios]
            rates = {c.sampling_rate for c in cuts}
            assert len(rates) == 1, f"one launch per batch needs a single sampling rate, got {sorted(rates)}"
            sr = rates.pop()
            perturbed = None
            if any(f != 1.0 for f in factors):  # (FusedMiniBatch.features_of, with the perturbed samples kept for `return_audio`)
                feats, feat_lens, perturbed = self._perturb_and_extract(audios, factors, wants, sr)
            else:
                feats, feat_lens = self.extractor.extract_collated(audios, sampling_rate=sr, padding_value=LOG_EPSILON)
            result = [feats if self.return_device is None else feats.to(self.return_device), feat_lens]
            if self.return_audio:  # (B, Tmax) zero-padded samples + their lengths, as the parent returns them
                flat = [a.reshape(-1) for a in (perturbed if perturbed is not None else audios)]
                result += [collate_vectors(flat, padding_value=0), torch.tensor([len(a) for a in flat], dtype=torch.int64)]
            if self.fault_tolerant:  # the cuts that survived audio loading
                result.append(cuts)
            return tuple(result)

else:

    class HipOnTheFlyFeatures:  # type: ignore[no-redef]
        def __init__(self, *args, **kwargs):
            raise ImportError("HipOnTheFlyFeatures consumes lhotse CutSets: install lhotse (lhotse.dataset.input_strategies.OnTheFlyFeatures)")
