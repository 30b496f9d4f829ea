"""The reference's own known-answer fixture for this path (test/fixtures/libri: a LibriSpeech utterance and the 40-dim fbank
the reference stores for it -- round-tripped through lilcom's lossy compression, which leaves them exact to 2^-6): the oracle on CPU, the HIP path on the
GPU.  tests/golden/libri_fixture.npz holds the first 3 s (oracle/make_golden_libri.py)."""
import os

import numpy as np
import pytest

from oracle.kaldi_ref import RefConfig, RefExtractor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libri_fixture.npz")
TOL = 2.0 ** -6 + 1e-4  # lossy compression of the stored fixture (measured: exactly 2^-6 against the oracle) + float32 noise


def test_oracle_reproduces_the_reference_fixture():
    z = np.load(GOLDEN)
    ref = RefExtractor(RefConfig(kind="fbank", num_filters=40), np.float32)
    y = ref.extract(z["pcm"].astype(np.float32) / 32768.0)
    assert y.shape == (300, 40) and z["feats"].shape == (290, 40)
    assert np.abs(y[:290] - z["feats"]).max() <= TOL


@pytest.mark.skipif(not os.path.isdir("/root/reference/test/fixtures/libri"), reason="reference tree not present")
def test_oracle_reproduces_the_whole_fixture_file():
    from oracle.make_golden_libri import load_reference_fixture

    pcm, feats = load_reference_fixture()
    y = RefExtractor(RefConfig(kind="fbank", num_filters=40), np.float32).extract(pcm.astype(np.float32) / 32768.0)
    assert y.shape == feats.shape == (1604, 40)  # test/features/test_kaldi_features.py:92-96
    assert np.abs(y - feats).max() <= TOL
    z = np.load(GOLDEN)
    assert np.array_equal(z["pcm"], pcm[:48000]) and np.array_equal(z["feats"], feats[:290])


@pytest.mark.gpu
@pytest.mark.parametrize("as_pcm16", [False, True])
def test_hip_fbank_reproduces_the_reference_fixture(as_pcm16):
    import lhotse_amd as LA

    z = np.load(GOLDEN)
    ex = LA.HipFbank(LA.HipFbankConfig(num_filters=40))
    x = z["pcm"] if as_pcm16 else z["pcm"].astype(np.float32) / 32768.0
    y = ex.extract(x, 16000)
    assert y.shape == (300, 40)
    assert np.abs(y[:290] - z["feats"]).max() <= TOL
    ref = RefExtractor(RefConfig(kind="fbank", num_filters=40), np.float64).extract(z["pcm"].astype(np.float64) / 32768.0)
    assert np.abs(y - ref).max() <= 2e-3 and np.linalg.norm(y - ref) / np.linalg.norm(ref) <= 1e-4


# ---- round 3: the whole utterance (256 640 samples of real speech), with the reference's own unquantised outputs ----------------------
FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libri_full.npz")
FULL_CASES = [("fbank40", RefConfig(kind="fbank", num_filters=40)), ("fbank80", RefConfig(kind="fbank")), ("mfcc13", RefConfig(kind="mfcc", num_filters=23))]


@pytest.mark.parametrize("key,rc", FULL_CASES, ids=[k for k, _ in FULL_CASES])
def test_oracle_equals_the_reference_on_the_whole_utterance(key, rc):
    z = np.load(FULL)
    y = RefExtractor(rc, np.float32).extract(z["pcm"].astype(np.float32) / 32768.0)
    assert y.shape == z[key].shape == (1604, rc.num_ceps if rc.kind == "mfcc" else rc.num_filters)
    assert np.abs(y - z[key]).max() <= (2e-3 if rc.kind == "mfcc" else 2e-4)  # same float32 arithmetic, different BLAS summation order
    assert np.abs(z["fbank40"] - z["stored_fbank40"]).max() <= TOL  # the reference's stored (lilcom) fixture


@pytest.mark.gpu
@pytest.mark.parametrize("as_pcm16", [False, True])
@pytest.mark.parametrize("key,rc", FULL_CASES, ids=[k for k, _ in FULL_CASES])
def test_hip_equals_the_reference_on_the_whole_utterance(key, rc, as_pcm16):
    import lhotse_amd as LA
    from _golden import record_parity

    z = np.load(FULL)
    if rc.kind == "mfcc":
        ex = LA.HipMfcc(LA.HipMfccConfig())
    else:
        ex = LA.HipFbank(LA.HipFbankConfig(num_filters=rc.num_filters))
    x = z["pcm"] if as_pcm16 else z["pcm"].astype(np.float32) / 32768.0
    y = ex.extract(x, 16000)
    want = z[key]
    truth = RefExtractor(rc, np.float64).extract(z["pcm"].astype(np.float64) / 32768.0)
    rec = record_parity("libri_full", (key, "pcm16" if as_pcm16 else "f32"), ex.kernel_name, y, want, truth)
    assert y.shape == want.shape
    # real speech has > 90 dB between the loudest and the quietest mel bin of the file: element-wise, both float32 implementations
    # are dominated by rounding in the quiet bins (the reference's own float32-vs-float64 max_abs is recorded next to ours)
    assert rec["rel_l2"] <= 1e-4 and rec["max_abs"] <= max(2e-3, 3 * rec["floor_max_abs"]), rec
    if key == "fbank40":
        assert np.abs(y - z["stored_fbank40"]).max() <= TOL
