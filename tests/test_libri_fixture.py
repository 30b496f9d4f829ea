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
