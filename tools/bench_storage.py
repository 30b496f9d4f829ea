#!/usr/bin/env python3
"""SURVEY section 8f row 3 micro-benchmark (CPU only, needs the real lhotse: authoring container): the batch driver's save
path with an INSTANT extractor on N one-second cuts -- lhotse's own CutSet.compute_and_store_features_batch (one .npy per cut,
per-cut Features + validate + recursive to_dict) against lhotse_amd.compute_and_store_features_batch with the packed
archive writer (one append per batch, manifest dicts from a per-recording template)."""
import argparse, os, shutil, sys, tempfile, time, wave
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import import_reference
import_reference()
import torch
import lhotse
from lhotse import CutSet, MonoCut, Recording, NumpyFilesWriter
from lhotse.audio import AudioSource
from lhotse.audio.backend import AudioBackend, set_current_audio_backend
from lhotse.features.base import FeatureExtractor, register_extractor
from dataclasses import dataclass
import lhotse_amd as LA

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=3000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--instant-audio", action="store_true", help="the audio backend returns a cached array (no file I/O, no decoding): isolates the save path")
ap.add_argument("--preloaded", action="store_true", help="both drivers get the SAME pre-loaded batches (the DataLoader is replaced): times extraction + save path only")
ap.add_argument("--workers", type=int, default=0, help="DataLoader worker processes that load the audio (both drivers)")
a = ap.parse_args()

_CACHED = (np.zeros((1, 16000), dtype=np.float32), 16000)


class StdlibWaveBackend(AudioBackend):
    def read_audio(self, path_or_fd, offset=0.0, duration=None, force_opus_sampling_rate=None):
        if a.instant_audio:
            return _CACHED
        with wave.open(str(path_or_fd), "rb") as f:
            sr, n, ch = f.getframerate(), f.getnframes(), f.getnchannels()
            start = int(round(offset * sr)); f.setpos(start)
            raw = f.readframes(n - start if duration is None else int(round(duration * sr)))
        return np.frombuffer(raw, dtype=np.int16).reshape(-1, ch).T.astype(np.float32) / 32768.0, sr
    def is_applicable(self, p): return str(p).endswith(".wav")
    handles_special_case = is_applicable
set_current_audio_backend(StdlibWaveBackend())

@dataclass
class InstantConfig:
    sampling_rate: int = 16000
    def to_dict(self): return {"sampling_rate": self.sampling_rate}
    @staticmethod
    def from_dict(d): return InstantConfig(**d)

@register_extractor
class Instant(FeatureExtractor):
    name = "instant"; config_type = InstantConfig
    @property
    def frame_shift(self): return 0.01
    def feature_dim(self, sampling_rate): return 80
    def extract(self, samples, sampling_rate):
        n = samples.shape[-1]
        return np.zeros(((n + 80) // 160, 80), dtype=np.float32)
    def extract_batch(self, samples, sampling_rate, lengths=None):
        return [self.extract(s, sampling_rate) for s in samples]

tmp = tempfile.mkdtemp(prefix="bench_storage_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
x = (np.random.RandomState(0).rand(16000) * 2 - 1)
cuts = []
for i in range(a.cuts):
    p = os.path.join(tmp, f"r{i}.wav")
    with wave.open(p, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes((x * 30000).astype(np.int16).tobytes())
    rec = Recording(id=f"rec{i}", sources=[AudioSource(type="file", channels=[0], source=p)], sampling_rate=16000, num_samples=16000, duration=1.0)
    cuts.append(MonoCut(id=f"cut{i}", start=0, duration=1.0, channel=0, recording=rec))
cs = CutSet.from_cuts(cuts)
ex = Instant()
if a.preloaded:
    import torch.utils.data as tud
    from lhotse.dataset import SimpleCutSampler, UnsupervisedWaveformDataset
    _real = tud.DataLoader
    _batches = list(_real(UnsupervisedWaveformDataset(collate=False), batch_size=None, sampler=SimpleCutSampler(cs, max_duration=600.0), num_workers=0))

    class _Preloaded:
        def __init__(self, *args, **kwargs):
            pass

        def __iter__(self):
            return iter(_batches)

    tud.DataLoader = _Preloaded
    import lhotse.cut.set as _lcs
    if hasattr(_lcs, "DataLoader"):
        _lcs.DataLoader = _Preloaded
best = {}
for rep in range(a.reps):
    for name in ("reference", "ours"):
        d = os.path.join(tmp, f"{name}{rep}")
        os.makedirs(d, exist_ok=True)
        t0 = time.perf_counter()
        if name == "reference":
            out = cs.compute_and_store_features_batch(extractor=ex, storage_path=os.path.join(d, "feats"), manifest_path=os.path.join(d, "cuts.jsonl.gz"),
                                                      batch_duration=600.0, num_workers=a.workers, storage_type=NumpyFilesWriter, overwrite=True)
        else:
            os.makedirs(d, exist_ok=True)
            out = LA.compute_and_store_features_batch(cs, extractor=ex, storage_path=os.path.join(d, "feats"), manifest_path=os.path.join(d, "cuts.jsonl.gz"),
                                                      batch_duration=600.0, num_workers=a.workers, overwrite=True)
        dt = time.perf_counter() - t0
        best[name] = min(best.get(name, 1e9), dt)
        n = 0
        for c in out:
            n += 1
            if n in (1, a.cuts):
                f = c.load_features()
                assert f.shape == (100, 80), f.shape
        assert n == a.cuts
        print(f"rep {rep} {name:10s} {dt:7.3f} s = {a.cuts / dt:8.0f} cuts/s")
print(f"best: reference {a.cuts / best['reference']:.0f} cuts/s, ours {a.cuts / best['ours']:.0f} cuts/s, ratio {best['reference'] / best['ours']:.2f}x")
shutil.rmtree(tmp, ignore_errors=True)
