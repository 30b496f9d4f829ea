"""GPU: HipWhisperFbank (kind HIPFEAT_WHISPER: whisper3_kernel with the normalisation fused; whisper2_kernel / generic_kernel +
whisper_norm_kernel behind HIPFEAT_WHISPER_VARIANT=2 / HIPFEAT_FORCE_GENERIC=1) against goldens produced by the
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
    assert ex.kernel_name.startswith("whisper3_kernel")
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
    assert fast.kernel_name.startswith("whisper3_kernel"), fast.kernel_name
    filters = W.slaney_mel_filters(16000, 400, n_mels)
    for x, a, b in zip(xs, fast.extract_batch(xs, 16000), slow.extract_batch(xs, 16000)):
        truth = W.log_mel_spectrogram(x, filters, dtype=np.float64)
        assert a.shape == b.shape == truth.shape
        assert np.abs(a - truth).max() <= 2e-4 and np.abs(b - truth).max() <= 2e-4, (len(x), np.abs(a - truth).max(), np.abs(b - truth).max())


@pytest.mark.parametrize("n_mels", [80, 128])
def test_the_three_whisper_kernels_agree(n_mels, monkeypatch):
    """whisper3 (wave-autonomous, fused normalisation) vs whisper2 + whisper_norm_kernel: same FFT arithmetic, different mel
    summation order; cuts of 1 .. 30 s so that a cut spans 1 .. 12 workgroups and the ragged workgroup -> cut search runs."""
    rng = np.random.RandomState(7)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) * a for n, a in [(480000, 0.3), (201, 1.0), (40960, 1.0), (41000, 0.01), (163840, 0.7), (16000, 1e-3)]]
    xs[0][80000:320000] *= 1e-6   # 15 s more than 80 dB under the rest: whole row blocks (workgroups) under the clamp, and two partial ones
    xs[4][100000:100400] = 0.0    # digital silence inside one frame span
    new = LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=n_mels))
    monkeypatch.setenv("HIPFEAT_WHISPER_VARIANT", "2")
    old = LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=n_mels))
    assert old.kernel_name.startswith("whisper_kernel2")  # (plans are created lazily: touch it while the switch is set)
    monkeypatch.delenv("HIPFEAT_WHISPER_VARIANT")
    assert new.kernel_name.startswith("whisper3_kernel<%d>" % (2 if n_mels == 80 else 3))
    filters = W.slaney_mel_filters(16000, 400, n_mels)
    for x, a, b in zip(xs, new.extract_batch(xs, 16000), old.extract_batch(xs, 16000)):
        truth = W.log_mel_spectrogram(x, filters, dtype=np.float64)
        assert a.shape == b.shape == truth.shape
        assert np.abs(a - b).max() <= 1e-5, (len(x), np.abs(a - b).max())
        assert np.abs(a - truth).max() <= 1e-4, (len(x), np.abs(a - truth).max())


def test_fused_normalisation_rearms_itself_and_handles_strided_rows():
    """The per-cut scratch (maximum, completion counter) behind a layout is re-armed by the workgroup that used it: repeated launches of
    one layout object give identical results; rows wider than the feature dimension take the element-wise sweep."""
    rng = np.random.RandomState(11)
    lens = np.array([160000, 8000, 47999, 320000, 201], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    flat = ((rng.rand(int(lens.sum())).astype(np.float32) - 0.5) * 0.8)
    ex = LA.HipWhisperFbank()
    plan = ex.plan
    L = plan.lib
    want = [ex.extract(flat[o : o + n], 16000) for o, n in zip(offs, lens)]
    T = np.array([len(wv) for wv in want])
    d_wave = torch.from_numpy(flat).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    h = np.zeros(1, dtype=np.uint64)
    L.check("hipfeat_layout_create", plan.handle, len(lens), _lib.addr(offs), _lib.addr(lens), None, None, 80, None, _lib.addr(h))
    out = torch.empty((int(T.sum()), 80), device="cuda")
    for it in range(9):  # more launches than scratch copies: every copy is used at least twice
        out.fill_(float("nan"))
        L.check("hipfeat_extract_layout", plan.handle, int(h[0]), d_wave.data_ptr(), out.data_ptr(), stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), np.concatenate(want), err_msg=f"launch {it}")
    L.check("hipfeat_layout_destroy", int(h[0]))
    # padded (B, Tmax, 96) output with stride 96 > 80: columns 80.. and rows beyond a cut stay untouched
    Tmax = int(T.max())
    d_out = torch.full((len(lens), Tmax, 96), -7.0, device="cuda")
    rows = (np.arange(len(lens)) * Tmax).astype(np.int64)
    L.check("hipfeat_extract", plan.handle, d_wave.data_ptr(), _lib.addr(offs), _lib.addr(lens), None, len(lens), d_out.data_ptr(), _lib.addr(rows), 96, stream)
    torch.cuda.synchronize()
    o = d_out.cpu().numpy()
    for b in range(len(lens)):
        np.testing.assert_array_equal(o[b, : T[b], :80], want[b])
        assert (o[b, T[b] :, :] == -7).all() and (o[b, :, 80:] == -7).all()


def test_many_cuts_many_launches_on_two_streams():
    """4000 x 3 s cuts (2 workgroups per cut), the same layout launched alternately on two streams: the scratch copies of a layout keep
    overlapping launches apart."""
    ex = LA.HipWhisperFbank()
    plan = ex.plan
    L = plan.lib
    B, S = 1500, 48000
    offs = np.arange(B, dtype=np.int64) * S
    lens = np.full(B, S, dtype=np.int64)
    g = torch.Generator(device="cuda").manual_seed(5)
    wave = (torch.rand(B * S, device="cuda", generator=g) - 0.5) * torch.repeat_interleave(torch.logspace(-3, 0, B, device="cuda"), S)
    h = np.zeros(1, dtype=np.uint64)
    L.check("hipfeat_layout_create", plan.handle, B, _lib.addr(offs), _lib.addr(lens), None, None, 80, None, _lib.addr(h))
    ref = torch.empty(B * 300, 80, device="cuda")
    L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    # every cut is normalised by ITS OWN maximum (amplitudes rise by 3 decades over the batch = 6 in log10 power = 1.5 in output units)
    tops = ref.view(B, 300, 80).amax(dim=(1, 2))
    assert 1.3 < float(tops[-1] - tops[0]) < 1.7
    os.environ["HIPFEAT_WHISPER_VARIANT"] = "2"
    try:
        old = LA.HipWhisperFbank()
        assert old.kernel_name.startswith("whisper_kernel2")
    finally:
        del os.environ["HIPFEAT_WHISPER_VARIANT"]
    ref2, _ = old.plan.run(wave, offs, lens, None)
    assert float((ref - ref2.view_as(ref)).abs().max()) <= 5e-5  # (the quietest cuts sit 60 dB down: rounding of the mel sums)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.empty_like(ref) for _ in range(6)]
    for i, o in enumerate(outs):
        L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), o.data_ptr(), (s1 if i % 2 == 0 else s2).cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)
    L.check("hipfeat_layout_destroy", int(h[0]))
