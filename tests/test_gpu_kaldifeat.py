"""GPU: kaldifeat-shaped extractors (rows a13/a14) run the same kernels as HipFbank/HipMfcc and follow
KaldifeatExtractor's input/return conventions (lhotse/features/kaldifeat.py:78-141)."""
import numpy as np
import pytest
import torch

import lhotse_amd as LA
from oracle import kaldi_ref as K

pytestmark = pytest.mark.gpu


def test_reference_cross_check_against_torch_native_layers():
    # the reference's own kaldifeat test (test/features/test_kaldifeat_features.py:103-116): rand(1, 32000), seed 99,
    # kaldifeat vs torch-native Fbank to 3 decimals.  Here: the HIP kaldifeat-shaped extractor vs the oracle.
    np.random.seed(99)
    x = np.random.rand(1, 32000).astype(np.float32)
    got = LA.HipKaldifeatFbank().extract(x, 16000)
    want = K.RefExtractor(K.RefConfig(kind="fbank"), np.float64).extract(x[0])
    assert got.shape == want.shape == (200, 80)
    np.testing.assert_almost_equal(got, want, decimal=3)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4
    m = LA.HipKaldifeatMfcc().extract(x, 16000)
    wm = K.RefExtractor(K.RefConfig(kind="mfcc", num_filters=23, num_ceps=13), np.float64).extract(x[0])
    assert m.shape == wm.shape == (200, 13)
    np.testing.assert_almost_equal(m, wm, decimal=3)


def test_bit_identical_to_hip_fbank_and_fast_path_selected():
    rng = np.random.RandomState(0)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in (16000, 23456, 160000)]
    kf = LA.HipKaldifeatFbank()
    fb = LA.HipFbank()
    assert "fft512c" in kf.kernel_name
    a = kf.extract([torch.from_numpy(x) for x in xs], 16000)
    b = fb.extract_batch([torch.from_numpy(x) for x in xs], 16000)
    for u, v in zip(a, b):
        assert u.is_cuda and torch.equal(u, v)


def test_input_and_return_conventions():
    rng = np.random.RandomState(1)
    kf = LA.HipKaldifeatFbank(LA.HipKaldifeatFbankConfig(mel_opts=LA.HipKaldifeatMelOptions(num_bins=40)))
    x = rng.rand(16000).astype(np.float32) - 0.5
    y = rng.rand(12000).astype(np.float32) - 0.5
    one = kf.extract(x, 16000)  # (T,) numpy -> numpy matrix
    assert isinstance(one, np.ndarray) and one.shape == (100, 40)
    assert isinstance(kf.extract([x], 16000), list)  # list in -> list out
    st = kf.extract(np.stack([x, x]), 16000)  # equal shapes -> stacked
    assert isinstance(st, np.ndarray) and st.shape == (2, 100, 40) and np.array_equal(st[0], one)
    rag = kf.extract([torch.from_numpy(x), torch.from_numpy(y)], 16000)  # ragged torch -> list of tensors
    assert isinstance(rag, list) and [tuple(r.shape) for r in rag] == [(100, 40), (75, 40)] and all(isinstance(r, torch.Tensor) for r in rag)
    # padded batch + lengths: items are trimmed first (kaldifeat.py:91-93), each framed on its own
    padded = torch.zeros(2, 16000)
    padded[0] = torch.from_numpy(x)
    padded[1, :12000] = torch.from_numpy(y)
    bl = kf.extract_batch(padded, 16000, lengths=torch.tensor([16000, 12000]))
    assert torch.equal(bl[0], rag[0]) and torch.equal(bl[1], rag[1])
    with pytest.raises(TypeError):
        kf.extract(x.astype(np.float64), 16000)


def test_other_options_route_through_the_generic_kernel():
    rng = np.random.RandomState(2)
    x = rng.rand(8000).astype(np.float32) - 0.5
    fo = LA.HipKaldifeatFrameOptions(sampling_rate=8000, window_type="blackman", blackman_coeff=0.4, snip_edges=True)
    cfg = LA.HipKaldifeatFbankConfig(frame_opts=fo, mel_opts=LA.HipKaldifeatMelOptions(num_bins=23, high_freq=3800.0), use_energy=True, use_power=False)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = LA.HipKaldifeatFbank(cfg).extract(x, 8000)
    assert got.shape == (98, 24)  # energy column first
    # same numbers as HipFbank with the equivalent flat config, except for the window coefficient
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat = LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, window_type="blackman", snip_edges=True, num_filters=23, high_freq=3800.0,
                                             use_energy=True, use_fft_mag=True)).extract(x, 8000)
    assert flat.shape == got.shape and not np.array_equal(flat, got) and np.abs(flat - got).max() < 0.2
