"""Shared pieces of the tests that run the product's host code under the REAL lhotse (authoring container only): the oracle-backed
stand-in for the device plan, a stdlib-`wave` audio backend (soundfile is not installed here) and a small CutSet on disk.  Test
infrastructure: the product never imports this."""
import wave

import numpy as np
import torch


def import_lhotse():
    """Put /root/reference on the path (with the three stub modules it needs) and bind lhotse_amd to the real lhotse -- also when
    another test module imported lhotse_amd first (its lhotse-dependent modules are then reloaded, in dependency order)."""
    import importlib

    from oracle.make_golden import import_reference

    import_reference()
    import lhotse  # noqa: F401
    import lhotse_amd
    import lhotse_amd.compat as compat

    if not compat.HAVE_LHOTSE:  # lhotse_amd was imported before the stubs were in place
        importlib.reload(compat)
        for name in ("extractors", "augmentation", "kaldifeat", "input_strategies", "whisper", "librosa_fbank", "layers", "storage", "sharding"):
            importlib.reload(importlib.import_module(f"lhotse_amd.{name}"))
        importlib.reload(lhotse_amd)
    assert compat.HAVE_LHOTSE
    return lhotse


def make_cpu_plan():
    """An oracle-backed class with the interface of lhotse_amd.extractors._Plan (no GPU in the authoring container)."""
    from oracle.kaldi_ref import RefConfig, RefExtractor

    kinds = {0: "spectrogram", 1: "log-spectrogram", 2: "fbank", 3: "mfcc"}

    class CpuPlan:
        def __init__(self, cfg, kind, device, mel_floor=None):
            fields = {k: getattr(cfg, k) for k in RefConfig.__dataclass_fields__ if hasattr(cfg, k)}
            self.ref = RefExtractor(RefConfig(kind=kinds[kind], **fields), np.float32)
            self.device = torch.device("cpu")
            self.feature_dim = self.ref.feature_dim
            self.kernel_name = "cpu-stand-in"
            self.n, self.shift, self.snip_edges = self.ref.n, self.ref.shift, int(cfg.snip_edges)

        def run(self, wave, offsets, lengths, padded):
            outs = []
            for i, (o, l) in enumerate(zip(offsets, lengths)):
                f = self.ref.extract(wave[o : o + l].numpy(), padded_len=None if padded is None else int(padded[i]))
                if padded is not None:
                    f = f[: (int(l) + self.shift // 2) // self.shift]
                outs.append(torch.from_numpy(np.ascontiguousarray(f)))
            return torch.cat(outs), np.array([len(o) for o in outs], dtype=np.int64)

        def run_collated(self, wave, offsets, lengths, padded, pad_value):
            packed, frames = self.run(wave, offsets, lengths, padded)
            out = torch.full((len(frames), int(frames.max()), self.feature_dim), pad_value, dtype=torch.float32)
            for i, f in enumerate(packed.split(frames.tolist())):
                out[i, : len(f)] = f
            return out, frames

        def close(self):
            pass

    return CpuPlan


def install_wave_backend():
    """int16 WAV through the stdlib; returns the previous backend."""
    from lhotse.audio.backend import AudioBackend, get_current_audio_backend, set_current_audio_backend

    class StdlibWaveBackend(AudioBackend):
        def read_audio(self, path_or_fd, offset=0.0, duration=None, force_opus_sampling_rate=None):
            with wave.open(str(path_or_fd), "rb") as f:
                sr, n, ch = f.getframerate(), f.getnframes(), f.getnchannels()
                start = int(round(offset * sr))
                f.setpos(start)
                raw = f.readframes(n - start if duration is None else int(round(duration * sr)))
            return np.frombuffer(raw, dtype=np.int16).reshape(-1, ch).T.astype(np.float32) / 32768.0, sr

        def is_applicable(self, p):
            return str(p).endswith(".wav")

        handles_special_case = is_applicable

    prev = get_current_audio_backend()
    set_current_audio_backend(StdlibWaveBackend())
    return prev


def write_cutset(directory, lengths, seed=0, sampling_rate=16000):
    """One int16 WAV + MonoCut per length; returns the (eager) CutSet."""
    from lhotse import CutSet, MonoCut, Recording
    from lhotse.audio import AudioSource

    rs = np.random.RandomState(seed)
    cuts = []
    for i, n in enumerate(lengths):
        x = rs.rand(n) - 0.5
        p = directory / f"r{i}.wav"
        with wave.open(str(p), "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(sampling_rate)
            f.writeframes((x * 32767).astype(np.int16).tobytes())
        rec = Recording(id=f"rec{i}", sources=[AudioSource(type="file", channels=[0], source=str(p))], sampling_rate=sampling_rate,
                        num_samples=n, duration=n / sampling_rate)
        cuts.append(MonoCut(id=f"cut{i}", start=0, duration=rec.duration, channel=0, recording=rec))
    return CutSet.from_cuts(cuts)
