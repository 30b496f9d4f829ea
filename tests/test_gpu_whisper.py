"""GPU: HipWhisperFbank (kind HIPFEAT_WHISPER: generic_kernel + whisper_norm_kernel) against goldens produced by the
reference's log_mel_spectrogram and against the float64 oracle.  Output units are log10 / 4, so the 1e-4-relative bar
of the Kaldi features (natural log) corresponds to about 1e-4 absolute here; low-energy bins inherit the float32 noise
of the reference's own STFT (|golden - float64 truth| is checked as the floor)."""
import os

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import _lib
from oracle import whisper_ref as W
from oracle.make_golden_whisper import CASES
from oracle.signals import crc, make_signal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(got, want, truth, ctx):
    floor = float(np.abs(want - truth).max())
    err = float(np.abs(got - truth).max())
    assert got.shape == want.shape, ctx
    assert err <= max(1e-4, 3 * floor), (ctx, err, floor)
    assert np.linalg.norm(got - truth) / max(np.linalg.norm(truth), 1e-30) <= max(1e-4, 3 * np.linalg.norm(want - truth) / max(np.linalg.norm(truth), 1e-30)), ctx


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_whisper_matches_reference_golden(case):
    name, n_mels, inputs = case
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ex = LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=n_mels))
    assert "whisper_kernel" in ex.kernel_name
    for i, (kind, n, seed) in enumerate(inputs):
        x = make_signal(kind, n, seed)
        assert crc(x) == int(z[f"crc{i}"])
        want = z[f"out{i}"]
        truth = W.log_mel_spectrogram(x, z["filters"], dtype=np.float64)
        got = ex.extract(x, 16000)
        assert isinstance(got, np.ndarray) and got.dtype == np.float32
        _close(got, want, truth, (name, i))
        if n % 160 >= 80:  # the zero padding row (whisper_fbank.py:70-79)
            assert np.all(got[-1] == 0.0)


def test_ragged_batch_torch_and_collated():
    rng = np.random.RandomState(0)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) * s for n, s in [(16000, 1.0), (40123, 0.1), (160000, 0.9), (201, 1.0), (8079, 0.5), (8080, 0.5)]]
    ex = LA.HipWhisperFbank()
    filters = W.slaney_mel_filters()
    outs = ex.extract_batch([torch.from_numpy(x) for x in xs], 16000)
    assert isinstance(outs, list) and all(o.is_cuda for o in outs)
    for x, o in zip(xs, outs):
        truth = W.log_mel_spectrogram(x, filters, dtype=np.float64)
        ref32 = W.log_mel_spectrogram(x, filters, dtype=np.float32)
        _close(o.cpu().numpy(), ref32, truth, len(x))
    # every cut is normalised by ITS OWN maximum: the batch result equals the per-cut result
    for x, o in zip(xs, outs):
        assert np.array_equal(ex.extract(x, 16000), o.cpu().numpy())
    col, lens = ex.extract_collated(xs, 16000)
    assert col.shape == (6, 1000, 80) and lens.tolist() == [W.num_rows(len(x)) for x in xs]
    for i, o in enumerate(outs):
        assert torch.equal(col[i, : len(o)], o) and torch.all(col[i, len(o) :] == np.float32(LA.compat.LOG_EPSILON))
    same = ex.extract_batch(np.stack([xs[0], xs[0]]), 16000)
    assert isinstance(same, np.ndarray) and same.shape == (2, 100, 80)


def test_dynamic_range_clamp_and_silence():
    ex = LA.HipWhisperFbank()
    # a loud burst followed by near silence: everything 8 decades under the peak is clamped to (max - 8 + 4) / 4
    x = np.zeros(32000, dtype=np.float32)
    x[:8000] = (np.random.RandomState(1).rand(8000).astype(np.float32) - 0.5)
    x[8000:] = 1e-7
    y = ex.extract(x, 16000)
    truth = W.log_mel_spectrogram(x, W.slaney_mel_filters(), dtype=np.float64)
    assert abs(float(y.min()) - float(truth.min())) < 1e-5 and np.abs(y - truth).max() < 2e-4
    assert np.isclose(y.min(), (truth.max() * 4 - 4 - 8 + 4) / 4, atol=1e-5)
    # digital silence: log10(1e-10) = -10 everywhere -> (-10 + 4) / 4 = -1.5
    z = ex.extract(np.zeros(16000, dtype=np.float32), 16000)
    assert np.abs(z + 1.5).max() <= 1e-6


def test_too_short_and_c_abi_validation():
    ex = LA.HipWhisperFbank()
    with pytest.raises(ValueError, match="reflect padding"):
        ex.extract(np.zeros(200, dtype=np.float32), 16000)  # torch.stft raises for pad >= length as well
    assert ex.extract(np.ones(201, dtype=np.float32) * 0.1, 16000).shape == (1, 80)
    lib = ex.plan.lib
    c = np.zeros((), dtype=_lib.CONFIG_DTYPE)
    c["struct_size"], c["kind"], c["frame_length"], c["frame_shift"], c["fft_length"], c["num_filters"] = _lib.CONFIG_DTYPE.itemsize, 4, 400, 160, 512, 80
    c["mel_floor"] = 1e-10
    win = np.ones(400, dtype=np.float32)
    mel = np.ones((257, 80), dtype=np.float32)
    h = np.zeros(1, dtype=np.uint64)
    cb = np.ascontiguousarray(c).reshape(1)
    assert lib.raw("hipfeat_plan_create", _lib.addr(cb), _lib.addr(win), _lib.addr(mel), None, None, 0, _lib.addr(h)) == _lib.ERR_INVALID
    assert "whisper" in lib.last_error()


@pytest.mark.parametrize("n_mels", [80, 128, 40])
def test_mfma_dft_kernel_agrees_with_the_generic_direct_dft(n_mels, monkeypatch):
    rng = np.random.RandomState(4)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in (16000, 4321, 160000, 201, 2559, 2560, 2561)]
    fast = LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=n_mels))
    monkeypatch.setenv("HIPFEAT_FORCE_GENERIC", "1")
    slow = LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=n_mels))
    assert "generic" in slow.kernel_name
    monkeypatch.delenv("HIPFEAT_FORCE_GENERIC")
    assert "whisper_kernel" in fast.kernel_name
    filters = W.slaney_mel_filters(16000, 400, n_mels)
    for x, a, b in zip(xs, fast.extract_batch(xs, 16000), slow.extract_batch(xs, 16000)):
        truth = W.log_mel_spectrogram(x, filters, dtype=np.float64)
        assert a.shape == b.shape == truth.shape
        assert np.abs(a - truth).max() <= 2e-4 and np.abs(b - truth).max() <= 2e-4, (len(x), np.abs(a - truth).max(), np.abs(b - truth).max())
