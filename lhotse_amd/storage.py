"""
Bulk save path for the batch feature-extraction driver.

``compute_and_store_features_batch`` below is ``CutSet.compute_and_store_features_batch`` (lhotse/cut/set.py:2197-2408)
with the same arguments, storage formats, manifests and resume semantics -- it drives the same ``SimpleCutSampler`` /
``UnsupervisedWaveformDataset`` / ``DataLoader`` and the same ``FeaturesWriter`` classes -- but with the per-cut overheads of
``_save_worker`` (:2307-2363) taken out of the way of a GPU extractor that is ~1000x faster than the CPU one it was written
for:

  * ONE device-to-host transfer per batch (the packed feature matrix) instead of ``feat_mat.cpu().numpy()`` per cut;
  * array writes of a batch go through a small thread pool when the writer stores one object per key
    (``numpy_files``: independent files) -- manifests are still emitted strictly in input order;
  * the cut manifest is flushed once per batch instead of once per cut.

Needs lhotse (it produces lhotse manifests); importing this module without lhotse works, calling the function does not.
On-GPU lossy compression in the style of lilcom is NOT provided: lilcom is a third-party codec that is not available
offline, so its bit stream cannot be pinned.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from .compat import HAVE_LHOTSE


def _batch_features_on_host(extractor, waves, sampling_rate: int, lengths) -> List[np.ndarray]:
    """Per-cut feature matrices as numpy views of ONE host array (one D2H for Hip* extractors)."""
    if lengths is None and hasattr(extractor, "_extract_items") and hasattr(extractor, "_to_host"):
        from .extractors import _as_1d_float

        extractor._check_sr(sampling_rate)
        items = [_as_1d_float(w.squeeze() if w.ndim > 1 else w, "compute_and_store_features_batch()") for w in waves]
        zero_pad = getattr(extractor.config, "edge_rule", "reflect") == "batch_zero_pad"
        pmax = max(int(x.shape[0]) for x in items) if zero_pad else None
        with torch.no_grad():
            packed, frames = extractor._extract_items(items, pmax)
            host = extractor._to_host(packed).numpy()
        bounds = np.concatenate([[0], np.cumsum(frames)])
        return [host[int(bounds[i]) : int(bounds[i + 1])] for i in range(len(frames))]
    feats = extractor.extract_batch(waves, sampling_rate=sampling_rate, lengths=lengths)
    if isinstance(feats, (np.ndarray, torch.Tensor)) and feats.ndim == 2:
        feats = [feats]
    return [f.cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f) for f in feats]


def compute_and_store_features_batch(
    cuts,
    extractor,
    storage_path,
    manifest_path=None,
    batch_duration: float = 600.0,
    num_workers: int = 4,
    collate: bool = False,
    augment_fn: Optional[Callable] = None,
    storage_type=None,
    overwrite: bool = False,
    save_threads: int = 8,
):
    """Drop-in for ``CutSet.compute_and_store_features_batch`` (same arguments plus ``save_threads``); returns the CutSet
    with ``Features`` manifests attached."""
    if not HAVE_LHOTSE:
        raise ImportError("compute_and_store_features_batch produces lhotse manifests: install lhotse")
    from lhotse import CutSet, Features, MonoCut
    from lhotse.cut import MixedCut, PaddingCut
    from lhotse.cut.data import DataCut
    from lhotse.dataset import SimpleCutSampler, UnsupervisedWaveformDataset
    from lhotse.qa import validate_features
    from lhotse.utils import fastcopy
    from torch.utils.data import DataLoader

    try:
        from lhotse.features.io import default_features_storage_backend  # newer lhotse
    except ImportError:  # pragma: no cover
        default_features_storage_backend = None
    if storage_type is None:
        if default_features_storage_backend is not None:
            storage_type = default_features_storage_backend()
        else:  # pragma: no cover
            from lhotse.features.io import NumpyFilesWriter as storage_type
    if storage_type.name == "numpy_files":
        storage_path = Path(storage_path)
        if storage_path.exists() and storage_path.is_file():
            storage_path = storage_path.with_name(f"{storage_path.name}_storage")
    frame_shift = extractor.frame_shift
    cuts_writer = CutSet.open_writer(manifest_path, overwrite=overwrite)
    sampler = SimpleCutSampler(cuts, max_duration=batch_duration)
    sampler.filter(lambda cut: cut.id not in cuts_writer.ignore_ids)
    dataset = UnsupervisedWaveformDataset(collate=collate)
    dloader = DataLoader(dataset, batch_size=None, sampler=sampler, num_workers=num_workers)
    parallel_writes = storage_type.name == "numpy_files" and save_threads > 1

    def _save_batch(batch_cuts: Sequence, feats: List[np.ndarray], pool: Optional[ThreadPoolExecutor]) -> None:
        todo = [(i, c) for i, c in enumerate(batch_cuts) if not isinstance(c, PaddingCut)]
        if pool is not None:
            keys = dict(zip((i for i, _ in todo), pool.map(lambda ic: feats_writer.write(ic[1].id, feats[ic[0]]), todo)))
        else:
            keys = {i: feats_writer.write(c.id, feats[i]) for i, c in todo}
        for i, cut in enumerate(batch_cuts):
            feat_mat = feats[i]
            if isinstance(cut, PaddingCut):
                cuts_writer.write(fastcopy(cut, num_frames=feat_mat.shape[0], num_features=feat_mat.shape[1], frame_shift=frame_shift))
                continue
            feat_manifest = Features(
                start=cut.start,
                duration=cut.duration,
                type=extractor.name,
                num_frames=feat_mat.shape[0],
                num_features=feat_mat.shape[1],
                frame_shift=frame_shift,
                sampling_rate=cut.sampling_rate,
                channels=cut.channel,
                storage_type=feats_writer.name,
                storage_path=str(feats_writer.storage_path),
                storage_key=keys[i],
            )
            validate_features(feat_manifest, feats_data=feat_mat)
            if isinstance(cut, DataCut):
                feat_manifest.recording_id = cut.recording_id
                cut = fastcopy(cut, features=feat_manifest)
            if isinstance(cut, MixedCut):
                feat_manifest.recording_id = cut.id
                cut = MonoCut(
                    id=cut.id,
                    start=0,
                    duration=cut.duration,
                    channel=0,
                    supervisions=[fastcopy(s, recording_id=cut.id, channel=0) for s in cut.supervisions],
                    features=feat_manifest,
                    recording=None,
                )
            cuts_writer.write(cut, flush=False)
        # one flush per batch (the reference flushes after every cut, cut/set.py:2363)
        if getattr(cuts_writer, "file", None) is not None:
            cuts_writer.file.flush()

    futures = []
    with cuts_writer, storage_type(storage_path, mode="w" if overwrite else "a") as feats_writer, ThreadPoolExecutor(max_workers=1) as saver:
        pool = ThreadPoolExecutor(max_workers=save_threads) if parallel_writes else None
        try:
            for batch in dloader:
                batch_cuts, waves = batch["cuts"], batch["audio"]
                wave_lens = batch["audio_lens"] if collate else None
                if len(batch_cuts) == 0:
                    continue
                assert all(c.sampling_rate == batch_cuts[0].sampling_rate for c in batch_cuts)
                if augment_fn is not None:
                    waves = [augment_fn(w, c.sampling_rate) for c, w in zip(batch_cuts, waves)]
                feats = _batch_features_on_host(extractor, waves, batch_cuts[0].sampling_rate, wave_lens)
                futures.append(saver.submit(_save_batch, list(batch_cuts), feats, pool))
            for f in futures:
                f.result()
        finally:
            if pool is not None:
                pool.shutdown()
    return cuts_writer.open_manifest()
