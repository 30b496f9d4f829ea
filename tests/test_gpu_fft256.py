"""GPU: fft256 fast path (8 kHz 25/10 ms frames; <= 16 ms frames at 16 kHz) against the float64 oracle and the generic kernel."""
import warnings

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from oracle import kaldi_ref as K
from _golden import ref32 as ref32_of

pytestmark = pytest.mark.gpu

CFGS = [
    ("fbank", dict(sampling_rate=8000, num_filters=40)),
    ("fbank", dict(sampling_rate=8000, num_filters=23, low_freq=60.0, high_freq=3800.0)),
    ("fbank", dict(sampling_rate=8000, num_filters=80, snip_edges=True)),
    ("fbank", dict(sampling_rate=16000, frame_length=0.016, frame_shift=0.008, num_filters=64)),  # N = 256: all 16 rows
    ("mfcc", dict(sampling_rate=8000)),
    ("mfcc", dict(sampling_rate=8000, num_filters=40, num_ceps=20, cepstral_lifter=0)),
    ("spectrogram", dict(sampling_rate=8000)),
    ("log-spectrogram", dict(sampling_rate=8000, remove_dc_offset=False, preemph_coeff=0.0, window_type="hamming")),
]
TABLE = {"fbank": (LA.HipFbank, LA.HipFbankConfig), "mfcc": (LA.HipMfcc, LA.HipMfccConfig), "spectrogram": (LA.HipSpectrogram, LA.HipSpectrogramConfig),
         "log-spectrogram": (LA.HipLogSpectrogram, LA.HipLogSpectrogramConfig)}


def _make(kind, cfg):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return TABLE[kind][0](TABLE[kind][1](**cfg))


@pytest.mark.parametrize("kind,cfg", CFGS, ids=[f"{k}-{i}" for i, (k, _) in enumerate(CFGS)])
def test_fft256_fast_path_matches_oracle_and_generic(kind, cfg, monkeypatch):
    sr = cfg["sampling_rate"]
    rng = np.random.RandomState(7)
    lens = [sr, 3 * sr + 17, 10 * sr, 2561, 2560, sr // 2 + 3, 20 * sr + 1]
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in lens]
    # low level + DC offset; without DC removal a large offset only measures float32 leakage noise (reference floor 3e-4)
    xs[1] = (xs[1] * 0.01 + (0.2 if cfg.get("remove_dc_offset", True) else 0.0)).astype(np.float32)
    fast = _make(kind, cfg)
    assert "fft256_kernel" in fast.kernel_name or "fft256c_kernel" in fast.kernel_name, fast.kernel_name
    monkeypatch.setenv("HIPFEAT_FORCE_GENERIC", "1")
    slow = _make(kind, cfg)
    assert "generic" in slow.kernel_name  # (the plan is created lazily, on first use)
    monkeypatch.delenv("HIPFEAT_FORCE_GENERIC")
    fields = {k: v for k, v in cfg.items() if k in K.RefConfig.__dataclass_fields__}
    if kind == "mfcc":
        fields.setdefault("num_filters", 23)
    ref64 = K.RefExtractor(K.RefConfig(kind=kind, **fields), np.float64)
    ref32 = ref32_of(K.RefConfig(kind=kind, **fields))
    a = fast.extract_batch(xs, sr)
    b = slow.extract_batch(xs, sr)
    for x, fa, fb in zip(xs, a, b):
        truth = ref64.extract(x)
        want = ref32.extract(x)
        assert fa.shape == fb.shape == truth.shape
        floor = np.linalg.norm(want - truth) / np.linalg.norm(truth)
        rel = np.linalg.norm(fa - truth) / np.linalg.norm(truth)
        assert rel <= max(1e-4, 3 * floor), (len(x), rel, floor)
        if kind in ("fbank", "mfcc"):
            fl = np.abs(want - truth).max()
            assert np.abs(fa - truth).max() <= max(2e-3, 3 * fl), (len(x), np.abs(fa - truth).max(), fl)
        # and the two kernels agree with each other at float32 noise level
        assert np.linalg.norm(fa - fb) / np.linalg.norm(fb) <= 2e-5 + 2 * floor


def test_fft256_tripwire_tone_and_collated():
    # known-answer style check on a two-tone signal at 8 kHz (sharp spectral lines, deep leakage floor)
    n = np.arange(8000, dtype=np.float64)
    x = (0.5 * np.sin(2 * np.pi * 440 * n / 8000) + 0.25 * np.sin(2 * np.pi * 3000 * n / 8000)).astype(np.float32)
    ex = _make("fbank", dict(sampling_rate=8000, num_filters=40))
    y = ex.extract(x, 8000)
    truth = K.RefExtractor(K.RefConfig(kind="fbank", sampling_rate=8000, num_filters=40), np.float64).extract(x)
    want = ref32_of(K.RefConfig(kind="fbank", sampling_rate=8000, num_filters=40)).extract(x)
    assert y.shape == (100, 40)
    assert np.abs(y - truth).max() <= max(2e-3, 3 * np.abs(want - truth).max())
    col, lens = ex.extract_collated([x, x[:4000]], 8000)
    assert col.shape == (2, 100, 40) and lens.tolist() == [100, 50] and torch.equal(col[0], torch.from_numpy(y).cuda())


@pytest.mark.parametrize("cfg,kernel", [(dict(sampling_rate=8000), "fft256c_kernel<13>"), (dict(sampling_rate=8000, num_filters=40), "fft256c_kernel<13>"),
                                        (dict(sampling_rate=8000, num_filters=23), "fft256_kernel<13,0>"),  # one set of 12 steps: outside the 2 x 8 schedule
                                        (dict(sampling_rate=16000, frame_length=0.016, frame_shift=0.008, num_filters=64), "fft256_kernel<16,0>"),  # 8 spans of 7 x 128 + 256 samples: over the LDS budget of two workgroups per CU
                                        (dict(sampling_rate=8000, frame_length=0.032, frame_shift=0.01, num_filters=64), "fft256c_kernel<16>")])
def test_wave_autonomous_and_tile_kernels_agree(cfg, kernel, monkeypatch):
    """fft256c (wave-autonomous, 4 x 4 x 1 matrix-core filterbank) against fft256 (32-frame tiles) on ragged batches, and the choice of kernel."""
    sr = cfg["sampling_rate"]
    rng = np.random.RandomState(3)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) * a for n, a in [(sr, 1.0), (10 * sr, 0.5), (2561, 1.0), (2560, 0.1), (30 * sr + 7, 0.9), (sr // 4, 1e-3)]]
    new = _make("fbank", cfg)
    assert new.kernel_name.startswith(kernel), new.kernel_name
    monkeypatch.setenv("HIPFEAT_FFT256_VARIANT", "b")
    old = _make("fbank", cfg)
    assert old.kernel_name.startswith("fft256_kernel")
    monkeypatch.delenv("HIPFEAT_FFT256_VARIANT")
    fields = {k: v for k, v in cfg.items() if k in K.RefConfig.__dataclass_fields__}
    ref32 = ref32_of(K.RefConfig(kind="fbank", **fields))
    for x, a, b in zip(xs, new.extract_batch([torch.from_numpy(x) for x in xs], sr), old.extract_batch(xs, sr)):
        a = a.cpu().numpy()
        want = ref32.extract(x)
        assert a.shape == b.shape == want.shape
        assert np.abs(a - b).max() <= 2e-3, (len(x), np.abs(a - b).max())
        assert np.linalg.norm(a - want) / np.linalg.norm(want) <= 1e-4, len(x)
        assert np.array_equal(new.extract(x, sr), a)  # batch == per cut, bit for bit
