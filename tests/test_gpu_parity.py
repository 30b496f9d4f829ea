"""
GPU parity tests (run on a real MI355X): the HIP path, called through the Hip* extractors
and the C ABI, against
  (1) the committed golden vectors produced by the reference itself,
  (2) the oracle (CPU restatement) on fresh seeded inputs,
both judged relative to float64 arithmetic.  The bar is the one north_star states, without escape clause:
     rel_l2(hip, ref) <= 1e-4        max_abs(hip, ref) <= 2e-3
for every filterbank / MFCC / spectrogram comparison on the goldens and the small seeded batches.  The HEADLINE workload (64 x 10 s of
noise = 5.1 M values per input, where a few values in a million sit next to the log(eps) clamp) is judged by the three clauses of
oracle/parity_bar.py on five inputs including bench.py's own sample -- the statement bench.py asserts in-run, word for word
(test_headline_parity_multi_seed).  Two families carry the reference's own float32 noise beyond that
(profiles/r02_parity.json records, per comparison, the achieved error, the reference's float32-vs-float64 floor and
whether a clause was needed): the LOG of single near-silent FFT bins (log-spectrogram: both clauses, `kind` says so) and
the element-wise bound of MFCCs of pure tones (a DCT row sums 23-40 such logs: abs clause only):
     rel_l2 <= max(1e-4, 3 * rel_l2(ref, float64))      max_abs <= max(2e-3, 3 * max_abs(ref, float64)).
"""
import numpy as np
import pytest
import torch

from _golden import CASES, err_stats, golden_rows, load_case, record_parity, ref32, ref_config
from oracle.kaldi_ref import RefConfig, RefExtractor

pytestmark = pytest.mark.gpu

CASE_NAMES = [c["name"] for c in CASES]
REL_TOL = 1e-4
ABS_TOL = 2e-3


def assert_parity(got, want, truth, ctx, abs_tol=ABS_TOL, suite="misc", kernel="", kind="fbank"):
    """No escape clause unless `kind` is one of the two families the module docstring names."""
    s = err_stats(got, want)
    floor = err_stats(want, truth)
    own = err_stats(got, truth)
    record_parity(suite, ctx, kernel, got, want, truth, REL_TOL, abs_tol)
    assert np.isfinite(np.asarray(got)).all(), ctx
    rel_bar = max(REL_TOL, 3 * floor["rel_l2"]) if kind == "log-spectrogram" else REL_TOL
    abs_bar = max(abs_tol, 3 * floor["max_abs"]) if kind in ("log-spectrogram", "mfcc") else abs_tol
    assert s["rel_l2"] <= rel_bar, (ctx, s, floor, own)
    assert s["max_abs"] <= abs_bar, (ctx, s, floor, own)


# log of a single near-silent FFT bin: both float32 implementations are dominated by rounding there
# (tools/fft_accuracy.py: power error / frame peak power = 3e-7 for reference arithmetic and HIP alike, which
# is 9e-4 vs 6e-3 RELATIVE on a bin 57 dB below the peak).  The norm-wise bar stays 1e-4; the element-wise
# bar for log-spectra is 1e-2 with >= 99.95 % of the elements inside rtol 1e-4 / atol 1e-3.
LOGSPEC_ABS_TOL = 1e-2


@pytest.mark.parametrize("name", CASE_NAMES)
def test_hip_matches_reference_golden(name):
    from _hip import run_case

    case, waves, z = load_case(name)
    outs, kernel = run_case(case, waves, want_kernel=True)
    ex64 = RefExtractor(ref_config(case), np.float64)
    truth = [ex64.extract(w) for w in waves] if case["mode"] == "extract" else ex64.extract_batch(waves, "batch_zero_pad")
    assert len(outs) == len(waves)
    for i, o in enumerate(outs):
        o = np.asarray(o)
        assert o.dtype == np.float32
        got, want = golden_rows(z, i, o)
        tr, _ = golden_rows(z, i, truth[i])
        assert_parity(got, want, tr, (name, i), suite="golden", kernel=kernel, kind=case["kind"])


@pytest.mark.parametrize("kind,cfg", [("fbank", {}), ("mfcc", {"num_filters": 40, "num_ceps": 40}), ("spectrogram", {}), ("log-spectrogram", {})])
def test_hip_matches_oracle_ragged_batch(kind, cfg):
    """Mixed-length batch, per-item reflect rule == every item extracted on its own."""
    from _hip import make_hip

    rs = np.random.RandomState(7)
    lens = [140, 161, 400, 1599, 1600, 1601, 16000, 31999, 48000, 7777, 160 * 33, 160 * 32 + 79]
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in lens]
    ex = make_hip(kind, cfg)
    outs = ex.extract_batch(waves, 16000)
    rc = dict(cfg)
    o32 = ref32(RefConfig(kind=kind, **rc))
    o64 = RefExtractor(RefConfig(kind=kind, **rc), np.float64)
    assert isinstance(outs, list) and len(outs) == len(waves)
    for w, o in zip(waves, outs):
        want, truth = o32.extract(w), o64.extract(w)
        assert o.shape == want.shape
        assert_parity(o, want, truth, (kind, len(w)), suite="ragged_batch", kernel=ex.kernel_name, kind=kind)
        # batch result == single-item result, bit for bit (same kernel, same arithmetic)
        np.testing.assert_array_equal(o, ex.extract(w, 16000))


def test_full_size_properties():
    """BASELINE config-2 sized cuts (10 s): properties that do not need a CPU pass over everything."""
    from _hip import make_hip

    ex = make_hip("fbank", {})
    g = torch.Generator().manual_seed(0)
    B, S = 64, 160000
    x = (torch.rand(B, S, generator=g) * 2 - 1) * 0.5
    y = ex.extract_batch(x, 16000)
    assert isinstance(y, torch.Tensor) and y.shape == (B, 1000, 80) and y.dtype == torch.float32
    assert torch.isfinite(y).all()
    # determinism
    y2 = ex.extract_batch(x, 16000)
    assert torch.equal(y, y2)
    # batch invariance: row b of a batch == the cut on its own == the cut inside a ragged batch
    y5 = ex.extract(x[5], 16000)
    assert torch.equal(y[5], y5)
    rag = ex.extract_batch([x[5], x[6][:100000], x[7][:1234]], 16000)
    assert torch.equal(rag[0], y[5])
    # time-shift covariance: shifting by k*160 samples shifts interior frames by k rows
    k = 3
    ys = ex.extract(x[0][k * 160 :], 16000)
    assert torch.allclose(ys[2:900], y[0][2 + k : 900 + k], atol=1e-5, rtol=0)
    # scaling: x -> a*x adds 2*log(a) to every log-mel above the floor
    a = 0.25
    ya = ex.extract(x[1] * a, 16000)
    assert torch.allclose(ya, y[1] + 2 * np.log(a), atol=2e-4, rtol=0)
    # (oracle parity of these and four other sets of full-size cuts: test_headline_parity_multi_seed)


# bench_rank<r>: rank r's sample in the driver's 8-GPU run; bench2000_rank1: rank 1 of `bench.py --gpus 2 --cuts 2000` (seed 1235, indices
# RandomState(4322).choice(2000)) -- the input on which the hip-vs-ref32 form of clause 2 showed ONE value of 10.2 M at 1.068
# (profiles/r05_bench_2ranks_gloo_first_attempt.txt); a permanent member of the list since round 6 (VERDICT r5 task 3)
HEADLINE_INPUTS = ["bench", "cpu0", 1, 2, 3] + [f"bench_rank{r}" for r in range(1, 8)] + ["bench2000_rank1"]


def _headline_cuts(which):
    """64 cuts of 10 s: "bench" = exactly the 64 cuts bench.py's in-run parity leg samples from its timed buffer on rank 0 (device generator
    seed 1234, 10 000 cuts filled 500 at a time, cut indices RandomState(4321)); "cpu0" = the CPU-generator input of
    test_full_size_properties (the input of rounds 1-3); an integer = that seed of the device generator."""
    B, S = 64, 160000
    if isinstance(which, str) and which.startswith("bench"):  # ("bench_rank<r>": seed 1234 + r, indices of RandomState(4321 + r))
        import bench

        cuts = 10000
        if which.startswith("bench2000_rank"):
            cuts, rank = 2000, int(which[len("bench2000_rank"):])
        else:
            rank = 0 if which == "bench" else int(which[len("bench_rank"):])
        idx = bench.fbank16k_parity_indices(cuts, rank)
        last = int(idx.max()) // bench.FILL_CHUNK * bench.FILL_CHUNK + bench.FILL_CHUNK  # whole chunks up to the last sampled cut
        wave = torch.empty((last, S), dtype=torch.float32, device="cuda")
        bench.fbank16k_fill(wave, 1234 + rank)
        return wave[torch.from_numpy(idx).cuda()].contiguous()
    if which == "cpu0":
        g = torch.Generator().manual_seed(0)
        return ((torch.rand(B, S, generator=g) * 2 - 1) * 0.5).cuda()
    g = torch.Generator(device="cuda").manual_seed(int(which))
    return torch.empty((B, S), dtype=torch.float32, device="cuda").uniform_(-0.5, 0.5, generator=g)


@pytest.mark.parametrize("which", HEADLINE_INPUTS)
def test_headline_parity_multi_seed(which):
    """The parity statement of the headline workload -- oracle/parity_bar.py, the SAME three clauses bench.py asserts in its in-run leg --
    on five independent sets of 64 full-size cuts, one of them bench.py's own sample (the suite used to meet a flat 2e-3
    element-wise bar on its own seed while the bench's seed landed at 3.8e-3).  Round 5: ref32 is the reference's real float32
    arithmetic (torch.fft.rfft in float32), K = 3; the numpy floor of rounds 1-4 is logged next to it (profiles/r05_parity.json)."""
    from _golden import PARITY_LOG
    from _hip import make_hip
    from oracle import parity_bar

    ex = make_hip("fbank", {})
    x = _headline_cuts(which)
    y = ex.extract_batch(x, 16000)
    assert y.shape == (64, 1000, 80) and torch.isfinite(y).all()
    from oracle.kaldi_torch import reference_f32

    r32 = reference_f32(RefConfig(kind="fbank"))  # ref32 = the reference's own float32 torch call sequence (bit-equal to the live reference)
    n32 = RefExtractor(RefConfig(kind="fbank"), np.float32)  # the ref32 of rounds 1-4 (float64 FFT rounded down), side by side
    o64 = RefExtractor(RefConfig(kind="fbank"), np.float64)
    yc, xc = y.cpu().numpy(), x.cpu().numpy()
    f = parity_bar.fold([parity_bar.figures(yc[b], r32.extract(xc[b]), o64.extract(xc[b]), alt32=n32.extract(xc[b])) for b in range(len(xc))])
    v = parity_bar.verdict(f)
    PARITY_LOG.append({"suite": "headline_multi_seed", "case": str(which), "kernel": ex.kernel_name.split(" ")[0], "rel_l2": f["rel_l2_max"],
                       "max_abs": f["max_abs_max"], "frac_within_rtol1e-4_atol1e-3": f["frac_within"], "floor_rel_l2": f["oracle_f32_vs_f64_rel_l2_max"],
                       "floor_max_abs": f["oracle_f32_vs_f64_max_abs"], "hip_vs_float64_max_abs": f["hip_vs_f64_max_abs"],
                       "hip_vs_float64_rms": f["hip_vs_f64_rms"], "floor_rms": f["oracle_f32_vs_f64_rms"], "K_measured": v["K_measured"],
                       "K_allowed": v["K_allowed"], "elementwise_bar": v["elementwise_bar"], "linear_domain_outside": f["lin_bad"],
                       "linear_domain_worst_share_of_tolerance": f["lin_margin_max"], "values_over_2e-3": f["n_over_2e-3"],
                       "linear_vs_f64_hip_worst_share": f["lin_own_max"], "linear_vs_f64_reference32_worst_share": f["lin_floor_max"],
                       "linear_vs_f64_bar": v["linear_bar_share_of_tolerance"], "linear_vs_f64_hip_over_1": f["lin_own_over1"],
                       "K_linear_measured": v["K_linear_measured"], "K_linear_allowed": v["K_linear_allowed"], "statement_version": v["statement_version"],
                       "linear_vs_f64_reference32_over_1": f["lin_floor_over1"],
                       "clause_needed_rel": False, "clause_needed_abs": bool(f["hip_vs_f64_max_abs"] > parity_bar.ABS_TOL),
                       "n_values": f["n_values"], "pass": v["pass"], "ref32": "oracle/kaldi_torch.reference_f32 (the reference's float32 torch calls)",
                       "numpy32_floor_max_abs": f["numpy32_vs_f64_max_abs"], "hip_vs_numpy32_max_abs": f["hip_vs_numpy32_max_abs"],
                       "numpy32_vs_ref32_max_abs": f["numpy32_vs_ref32_max_abs"], "K_against_numpy32_floor": v["K_against_numpy32_floor"]})
    assert v["pass_rel_l2"], (which, f)
    assert v["pass_linear"], (which, f)
    assert v["pass_elementwise"], (which, f, v)


def test_too_short_and_errors():
    from _hip import make_hip

    ex = make_hip("fbank", {})
    with pytest.raises(ValueError):
        ex.extract(np.zeros(139, dtype=np.float32), 16000)
    assert ex.extract(np.zeros(140, dtype=np.float32), 16000).shape == (1, 80)
    with pytest.raises(AssertionError):
        ex.extract(np.zeros(1600, dtype=np.float32), 8000)
    with pytest.raises(TypeError):
        ex.extract(np.zeros(1600, dtype=np.float64), 16000)


@pytest.mark.parametrize("raw", [True, False])
def test_mfcc_use_energy_replaces_c0(raw):
    """MFCC use_energy=True crashes in the reference (layers.py:721-722); defined here as Kaldi does and as
    that line intends: the log-energy replaces C0, everything else is unchanged."""
    from _hip import make_hip
    from oracle.kaldi_ref import RefConfig, RefExtractor

    rng = np.random.RandomState(11)
    x = rng.rand(24000).astype(np.float32) - 0.5
    cfg = {"use_energy": True, "raw_energy": raw, "energy_floor": 1e-3}
    ex = make_hip("mfcc", cfg)
    assert "wave_kernel" in ex.kernel_name or "generic" in ex.kernel_name
    got = ex.extract(x, 16000)
    plain = make_hip("mfcc", {}).extract(x, 16000)
    want = RefExtractor(RefConfig(kind="mfcc", num_filters=23, **cfg), np.float64).extract(x)
    assert got.shape == plain.shape == want.shape == (150, 13)
    assert np.abs(got - want).max() <= 2e-3
    assert np.abs(got[:, 1:] - plain[:, 1:]).max() <= 2e-3 and np.abs(got[:, 0] - plain[:, 0]).max() > 1.0
    fb = make_hip("fbank", {"use_energy": True, "raw_energy": raw, "energy_floor": 1e-3, "num_filters": 23}).extract(x, 16000)
    assert np.abs(got[:, 0] - fb[:, 0]).max() <= 1e-5  # the same log-energy Fbank prepends


def test_dither_is_gaussian_noise_on_the_waveform():
    """dither draws torch.randn on the device (layers.py:189-193) -- not bit-reproducible across
    implementations, so the check is statistical: on digital silence the features equal those of unit-variance Gaussian
    noise scaled by `dither`; the caller's buffer is not modified; torch.manual_seed makes it repeatable."""
    from _hip import make_hip
    from oracle.kaldi_ref import RefConfig, RefExtractor

    x = torch.zeros(160000, device="cuda")
    ex = make_hip("fbank", {"dither": 0.01})
    assert ex.kernel_name.startswith("fft512c_kernel")
    torch.manual_seed(3)
    a = ex.extract(x, 16000)
    torch.manual_seed(3)
    b = ex.extract(x, 16000)
    assert torch.equal(a, b) and float(x.abs().max()) == 0.0  # repeatable, input untouched
    c = ex.extract(x, 16000)
    assert not torch.equal(a, c)
    noise = (np.random.RandomState(0).randn(160000) * 0.01).astype(np.float32)
    want = RefExtractor(RefConfig(kind="fbank"), np.float64).extract(noise)
    # per-mel means over 1000 overlapping frames of log-energies: standard error ~ 0.05 for the narrowest filters
    got = a.cpu().numpy()
    assert np.abs(got.mean(axis=0) - want.mean(axis=0)).max() < 0.3
    assert abs(float(got.mean()) - float(want.mean())) < 0.05
    assert abs(float(got.std()) - float(want.std())) < 0.1
    # dither 0 on the same extractor type stays exact
    z = make_hip("fbank", {}).extract(x, 16000)
    assert torch.all(z == z[0, 0])
    # host input, batch API
    outs = ex.extract_batch([np.zeros(16000, dtype=np.float32), np.zeros(8000, dtype=np.float32)], 16000)
    assert outs[0].shape == (100, 80) and np.isfinite(outs[0]).all() and outs[0].std() > 0.1


def test_return_conventions():
    """extractors.py:485-554: list/stack/bare-item and numpy/torch conventions."""
    from _hip import make_hip

    ex = make_hip("fbank", {})
    a = np.random.RandomState(0).rand(1600).astype(np.float32)
    b = np.random.RandomState(1).rand(3200).astype(np.float32)
    r = ex.extract_batch([a], 16000)
    assert isinstance(r, list) and len(r) == 1 and isinstance(r[0], np.ndarray) and r[0].shape == (10, 80)
    r = ex.extract_batch(a, 16000)
    assert isinstance(r, np.ndarray) and r.shape == (10, 80)
    r = ex.extract_batch(np.stack([a, a]), 16000)
    assert isinstance(r, np.ndarray) and r.shape == (2, 10, 80)
    r = ex.extract_batch([a, b], 16000)
    assert isinstance(r, list) and [x.shape for x in r] == [(10, 80), (20, 80)]
    r = ex.extract_batch([torch.from_numpy(a), torch.from_numpy(b)], 16000)
    assert isinstance(r, list) and all(isinstance(x, torch.Tensor) for x in r)
    r = ex.extract_batch([torch.from_numpy(a)[None], torch.from_numpy(a)[None]], 16000)
    assert isinstance(r, torch.Tensor) and r.shape == (2, 10, 80)
    # (C, T) input: channel 0 only
    st = np.stack([a, b[:1600]])
    np.testing.assert_array_equal(ex.extract(st, 16000), ex.extract(a, 16000))
    # torch in -> torch out on the device for fbank, .cpu() for spectrogram (extractors.py:338-341)
    assert ex.extract(torch.from_numpy(a), 16000).device.type == "cuda"
    assert make_hip("spectrogram", {}).extract(torch.from_numpy(a), 16000).device.type == "cpu"
    # device tensors are accepted as they are
    r = ex.extract_batch([torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()], 16000)
    assert r[1].device.type == "cuda" and r[1].shape == (20, 80)


def test_c_abi_padded_output_and_host_form():
    """The C ABI directly: padded (B, Tmax, F) output with a row stride, and the host-pointer form."""
    from _hip import make_hip
    from lhotse_amd import _lib

    ex = make_hip("fbank", {})
    plan = ex.plan
    L = plan.lib
    rs = np.random.RandomState(3)
    lens = np.array([16000, 8000, 12000], dtype=np.int64)
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in lens]
    want = ex.extract_batch(waves, 16000)
    # host form, packed
    flat = np.concatenate(waves)
    offs = np.array([0, 16000, 24000], dtype=np.int64)
    T = np.array([100, 50, 75])
    out = np.empty((int(T.sum()), 80), dtype=np.float32)
    L.check("hipfeat_extract_host", plan.handle, _lib.addr(flat), flat.size, _lib.addr(offs), _lib.addr(lens), None, 3,
            _lib.addr(out), out.size, None, 80, None)
    np.testing.assert_array_equal(out, np.concatenate(want))
    # device form, padded output (B, Tmax, 96) with stride 96 > 80
    d_wave = torch.from_numpy(flat).cuda()
    d_out = torch.full((3, 100, 96), -1.0, device="cuda")
    rows = np.array([0, 100, 200], dtype=np.int64)
    L.check("hipfeat_extract", plan.handle, d_wave.data_ptr(), _lib.addr(offs), _lib.addr(lens), None, 3,
            d_out.data_ptr(), _lib.addr(rows), 96, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = d_out.cpu().numpy()
    for b in range(3):
        np.testing.assert_array_equal(o[b, : T[b], :80], want[b])
        assert (o[b, T[b] :, :] == -1).all() and (o[b, :, 80:] == -1).all()
    # layout object reuse
    h = np.zeros(1, dtype=np.uint64)
    L.check("hipfeat_layout_create", plan.handle, 3, _lib.addr(offs), _lib.addr(lens), None, None, 80, None, _lib.addr(h))
    assert L.raw("hipfeat_layout_total_frames", int(h[0])) == 225
    d_o2 = torch.empty((225, 80), device="cuda")
    for _ in range(3):
        L.check("hipfeat_extract_layout", plan.handle, int(h[0]), d_wave.data_ptr(), d_o2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_o2.cpu().numpy(), np.concatenate(want))
    L.check("hipfeat_layout_destroy", int(h[0]))


def test_one_very_long_cut_and_many_short_ones():
    """SURVEY section 5 'long recordings': one 20-minute cut (the kaldifeat wrapper's chunk_size case,
    lhotse/features/kaldifeat.py:160-163) is just more frame tiles; and a batch of 3000 minimal cuts
    exercises the ragged workgroup -> cut search."""
    from _hip import make_hip

    ex = make_hip("fbank", {})
    g = torch.Generator().manual_seed(3)
    S = 20 * 60 * 16000 + 77
    x = (torch.rand(S, generator=g) - 0.5)
    y = ex.extract(x, 16000)
    T = (S + 80) // 160
    assert y.shape == (T, 80) and torch.isfinite(y).all()
    o32 = ref32(RefConfig(kind="fbank"))
    o64 = RefExtractor(RefConfig(kind="fbank"), np.float64)
    # the middle minute and both ends against the oracle (frames are independent: a segment that starts
    # on a frame boundary reproduces the interior rows)
    f0 = 60000
    seg = x[f0 * 160 - 120 * 160 : (f0 + 700) * 160].numpy()
    want, truth = o32.extract(seg), o64.extract(seg)
    got = y[f0 - 120 : f0 - 120 + want.shape[0]].cpu().numpy()
    assert_parity(got[5:-5], want[5:-5], truth[5:-5], "long-middle")
    head = o32.extract(x[:32000].numpy())
    assert_parity(y[:150].cpu().numpy(), head[:150], o64.extract(x[:32000].numpy())[:150], "long-head")
    k = T - 200  # a tail segment that starts on the frame grid keeps the frame alignment and the right reflection
    tail = x[160 * k :].numpy()
    assert_parity(y[k + 5 :].cpu().numpy(), o32.extract(tail)[5:], o64.extract(tail)[5:], "long-tail")
    # many short cuts of every length 140..3139
    waves = [x[i * 100 : i * 100 + 140 + i] for i in range(3000)]
    outs = ex.extract_batch(waves, 16000)
    assert len(outs) == 3000
    for i in (0, 1, 19, 20, 21, 179, 180, 1234, 2999):
        w = waves[i].numpy()
        assert outs[i].shape == ((len(w) + 80) // 160, 80)
        assert_parity(outs[i].cpu().numpy(), o32.extract(w), o64.extract(w), ("short", i))


def test_generic_and_fast_kernels_agree(monkeypatch):
    """The wave-autonomous fft512 kernel (default for log-mel), the 16-frame-tile fft512 kernel (HIPFEAT_FFT512_VARIANT=b; the
    default for MFCC / spectrograms) and the generic kernel (HIPFEAT_FORCE_GENERIC=1) are three implementations of the same
    arithmetic."""
    from _hip import make_hip

    rs = np.random.RandomState(11)
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in (16000, 4000, 160000, 12345)]
    outs = {}
    for name, env in [("fast_c", {}), ("fast_b", {"HIPFEAT_FFT512_VARIANT": "b"}), ("generic", {"HIPFEAT_FORCE_GENERIC": "1"})]:
        for k in ("HIPFEAT_FFT512_VARIANT", "HIPFEAT_FORCE_GENERIC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ex = make_hip("fbank", {})
        outs[name] = (ex.kernel_name, ex.extract_batch(waves, 16000))
    assert outs["fast_c"][0].startswith("fft512c_kernel") and outs["fast_b"][0].startswith("fft512b_kernel") and outs["generic"][0] == "generic"
    for a, b in zip(outs["fast_c"][1], outs["generic"][1]):
        assert err_stats(a, b)["rel_l2"] < 2e-6
    for a, b in zip(outs["fast_c"][1], outs["fast_b"][1]):
        assert err_stats(a, b)["rel_l2"] < 2e-6
    # the same three for MFCC (40 filters x 40 cepstra) and for a 23-filter log-mel bank (one accumulator set of 32 steps)
    for kind, cfg in (("mfcc", {"num_filters": 40, "num_ceps": 40}), ("mfcc", {}), ("fbank", {"num_filters": 23})):
        res = {}
        for name, env in [("fast_c", {}), ("fast_b", {"HIPFEAT_FFT512_VARIANT": "b"}), ("generic", {"HIPFEAT_FORCE_GENERIC": "1"})]:
            for k in ("HIPFEAT_FFT512_VARIANT", "HIPFEAT_FORCE_GENERIC"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ex = make_hip(kind, cfg)
            res[name] = (ex.kernel_name, ex.extract_batch(waves, 16000))
        assert res["fast_c"][0].startswith("fft512c_kernel<13>") and res["fast_b"][0].startswith("fft512b_kernel") and res["generic"][0] == "generic", res["fast_c"][0]
        for a, b, g in zip(res["fast_c"][1], res["fast_b"][1], res["generic"][1]):
            assert np.abs(a - b).max() <= 2e-3 and np.abs(a - g).max() <= 2e-3, (kind, cfg, np.abs(a - b).max(), np.abs(a - g).max())


@pytest.mark.parametrize("cfg,kernel", [({}, "fft512c_kernel<13> mfcc"), ({"num_filters": 40, "num_ceps": 40}, "fft512c_kernel<13> mfcc"),
                                        ({"num_filters": 80, "num_ceps": 20, "cepstral_lifter": 0}, "fft512b_kernel<13,1> mfcc"),
                                        ({"num_filters": 40, "num_ceps": 13, "frame_length": 0.02}, "fft512c_kernel<10> mfcc"),
                                        ({"num_filters": 30, "num_ceps": 30, "cepstral_lifter": 0, "preemph_coeff": 0.0, "window_type": "hamming"}, "fft512c_kernel<13> mfcc")])
def test_mfcc_fast_path(cfg, kernel):
    """MFCC on the fft512 kernels against the oracle: the wave-autonomous kernel (filterbank on one accumulator set of 32 steps, DCT as a
    second run of 4 x 4 x 1 matrix-core blocks with its operands resident in registers) for up to 40 filters -- including the 23-filter /
    13-cepstra Kaldi default --, the 16-frame-tile kernel (DCT as a 16 x 16 x 4 GEMM) beyond."""
    from _hip import make_hip

    ex = make_hip("mfcc", cfg)
    assert ex.kernel_name.startswith(kernel), ex.kernel_name
    rs = np.random.RandomState(21)
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in (16000, 140, 31999, 160000, 5000)]
    outs = ex.extract_batch(waves, 16000)
    rc = dict(cfg)
    rc.setdefault("num_filters", 23)
    o32 = ref32(RefConfig(kind="mfcc", **rc))
    o64 = RefExtractor(RefConfig(kind="mfcc", **rc), np.float64)
    for w, o in zip(waves, outs):
        assert_parity(o, o32.extract(w), o64.extract(w), ("mfcc-fast", cfg, len(w)), suite="mfcc_fast_path", kernel=ex.kernel_name, kind="mfcc")


def test_c_abi_unaligned_offsets_and_staging_ring():
    """Cuts that start at arbitrary (not 16-byte aligned) offsets take the scalar staging path and must
    give the same bits as aligned ones; ten back-to-back asynchronous hipfeat_extract calls recycle the
    pinned descriptor ring (4 slots) without a host sync in between."""
    from _hip import make_hip
    from lhotse_amd import _lib

    ex = make_hip("fbank", {})
    plan = ex.plan
    L = plan.lib
    rs = np.random.RandomState(9)
    lens = np.array([16001, 48000, 7777, 160000], dtype=np.int64)
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in lens]
    want = np.concatenate(ex.extract_batch(waves, 16000))
    gaps = [1, 2, 3, 5]
    offs, cur, parts = [], 0, []
    for w, g in zip(waves, gaps):
        parts.append(np.zeros(g, dtype=np.float32))
        cur += g
        offs.append(cur)
        parts.append(w)
        cur += len(w)
    flat = torch.from_numpy(np.concatenate(parts)).cuda()
    offs = np.array(offs, dtype=np.int64)
    total = int(((lens + 80) // 160).sum())
    outs = [torch.empty((total, 80), device="cuda") for _ in range(10)]
    st = torch.cuda.current_stream().cuda_stream
    for o in outs:
        L.check("hipfeat_extract", plan.handle, flat.data_ptr(), _lib.addr(offs), _lib.addr(lens), None, 4, o.data_ptr(), None, 80, st)
    torch.cuda.synchronize()
    for o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want)


@pytest.mark.parametrize("kind,cfg", [("spectrogram", {}), ("log-spectrogram", {}), ("spectrogram", {"use_fft_mag": True}),
                                      ("log-spectrogram", {"use_fft_mag": True, "window_type": "hamming"})])
def test_spectrogram_fast_path(kind, cfg):
    """(log-)spectrogram on the fft512 kernel: ragged batch against the oracle, incl. the Nyquist bin."""
    from _hip import make_hip

    ex = make_hip(kind, cfg)
    assert ex.kernel_name.startswith("fft512b_kernel") and " spectrogram " in ex.kernel_name, ex.kernel_name
    rs = np.random.RandomState(31)
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in (16000, 140, 31999, 80000)]
    outs = ex.extract_batch(waves, 16000)
    o32 = ref32(RefConfig(kind=kind, **cfg))
    o64 = RefExtractor(RefConfig(kind=kind, **cfg), np.float64)
    for w, o in zip(waves, outs):
        assert o.shape[1] == 257
        want = o32.extract(w)
        assert_parity(np.asarray(o), want, o64.extract(w), (kind, cfg, len(w)), abs_tol=LOGSPEC_ABS_TOL if kind == "log-spectrogram" else ABS_TOL, suite="spectrogram_fast_path", kernel=ex.kernel_name, kind=kind)
        assert err_stats(np.asarray(o), want)["frac_within"] >= 0.9995


@pytest.mark.parametrize("cfg", [{"num_filters": 23}, {"num_filters": 128}, {"num_filters": 64, "low_freq": 0.0, "high_freq": 0.0},
                                 {"num_filters": 24, "frame_length": 0.032, "frame_shift": 0.016}])
def test_fbank_fast_path_other_filterbanks(cfg):
    from _hip import make_hip

    ex = make_hip("fbank", cfg)
    assert ex.kernel_name.startswith("fft512"), ex.kernel_name  # "c" when the filterbank fits its static schedule, else "b"
    rs = np.random.RandomState(41)
    waves = [(rs.rand(n).astype(np.float32) - 0.5) for n in (16000, 600, 31999)]
    outs = ex.extract_batch(waves, 16000)
    o32 = ref32(RefConfig(kind="fbank", **cfg))
    o64 = RefExtractor(RefConfig(kind="fbank", **cfg), np.float64)
    for w, o in zip(waves, outs):
        assert_parity(o, o32.extract(w), o64.extract(w), ("fbank-fast", cfg, len(w)), suite="fbank_other_filterbanks", kernel=ex.kernel_name)


def test_layout_launch_is_hip_graph_capturable():
    """hipfeat_extract_layout is a pure kernel launch (descriptors already resident, no allocation, no sync), so a
    fixed-shape batch can be captured into a HIP graph on the caller's stream and replayed on new audio."""
    from _hip import make_hip
    from lhotse_amd import _lib

    ex = make_hip("fbank", {})
    plan = ex.plan
    L = plan.lib
    B, S = 64, 48000
    offs = np.arange(B, dtype=np.int64) * S
    lens = np.full(B, S, dtype=np.int64)
    wave = torch.empty(B * S, device="cuda").uniform_(-0.5, 0.5)
    out = torch.zeros(B * 300, 80, device="cuda")
    h = np.zeros(1, dtype=np.uint64)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        L.check("hipfeat_layout_create", plan.handle, B, _lib.addr(offs), _lib.addr(lens), None, None, 80, side.cuda_stream, _lib.addr(h))
        layout = int(h[0])
        L.check("hipfeat_extract_layout", plan.handle, layout, wave.data_ptr(), out.data_ptr(), side.cuda_stream)  # warm-up
    side.synchronize()
    want1 = out.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        L.check("hipfeat_extract_layout", plan.handle, layout, wave.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want1)
    wave.uniform_(-0.25, 0.25)  # new audio in the same buffers
    graph.replay()
    torch.cuda.synchronize()
    ref, _ = plan.run(wave, offs, lens, None)
    assert torch.equal(out, ref) and not torch.equal(out, want1)
    L.check("hipfeat_layout_destroy", layout)


def test_plans_sharing_a_kernel_with_different_lds_footprints():
    """The dynamic-LDS limit is per FUNCTION: a later, smaller plan of the same kernel instance must not lower it
    under an earlier plan (hipfeat.hip ensure_dynamic_lds)."""
    from _hip import make_hip
    from oracle.kaldi_ref import RefConfig, RefExtractor

    x = np.random.RandomState(5).rand(48000).astype(np.float32) - 0.5
    big = make_hip("mfcc", {"num_filters": 40, "num_ceps": 40})
    a0 = big.extract(x, 16000)
    small = make_hip("mfcc", {})
    assert big.kernel_name.split(" ")[0] == small.kernel_name.split(" ")[0] and big.kernel_name != small.kernel_name
    b = small.extract(x, 16000)
    a1 = big.extract(x, 16000)  # the big plan still launches with its own footprint
    assert np.array_equal(a0, a1)
    for got, cfg in ((a1, dict(num_filters=40, num_ceps=40)), (b, dict(num_filters=23, num_ceps=13))):
        want = RefExtractor(RefConfig(kind="mfcc", **cfg), np.float64).extract(x)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4


@pytest.mark.parametrize("sr", [8000, 16000, 24000, 48000])
def test_rounds_per_workgroup_do_not_change_the_bits(sr):
    """The wave-autonomous kernels take many rounds per workgroup in long launches and few in short ones (a property of the layout):
    a cut gives the same bits alone, in a small batch and in a launch of thousands of cuts."""
    from _hip import make_hip

    ex = make_hip("fbank", {"sampling_rate": sr})
    assert "c_kernel" in ex.kernel_name, ex.kernel_name
    g = torch.Generator(device="cuda").manual_seed(sr)
    n = 3 * sr + 123
    big = torch.rand(2500, n, device="cuda", generator=g) - 0.5
    all_ = ex.extract_batch(big, sr)
    few = ex.extract_batch(big[:7], sr)
    one = ex.extract_batch(big[1234:1235], sr)
    assert torch.equal(all_[:7], few) and torch.equal(all_[1234:1235].reshape(one.shape), one)
    long_cut = torch.rand(1, 600 * sr // 10, device="cuda", generator=g) - 0.5  # one minute: one cut, many workgroups
    a = ex.extract_batch(long_cut, sr)
    b = ex.extract_batch(torch.cat([long_cut, long_cut]), sr)
    assert torch.equal(a.reshape(b[0].shape), b[0]) and torch.equal(b[0], b[1])
