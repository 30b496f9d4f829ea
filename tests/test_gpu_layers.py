"""GPU: the layer modules (HipWav2Spec / LogSpec / LogFilterBank / MFCC) against the float64 oracle, row by row."""
import numpy as np
import pytest
import torch

import lhotse_amd as LA
from oracle.kaldi_ref import RefConfig, RefExtractor

pytestmark = pytest.mark.gpu

CASES = [
    ("HipWav2LogFilterBank", "fbank", {}),
    ("HipWav2LogFilterBank", "fbank", {"use_energy": True, "num_filters": 40, "snip_edges": True}),
    ("HipWav2LogFilterBank", "fbank", {"sampling_rate": 8000, "num_filters": 23}),
    ("HipWav2MFCC", "mfcc", {}),
    ("HipWav2MFCC", "mfcc", {"num_filters": 40, "num_ceps": 40}),
    ("HipWav2Spec", "spectrogram", {}),  # use_energy=True by default in the layers
    ("HipWav2Spec", "spectrogram", {"use_energy": False, "use_fft_mag": True}),
    ("HipWav2LogSpec", "log-spectrogram", {}),
    ("HipWav2LogSpec", "log-spectrogram", {"use_energy": False, "sampling_rate": 22050}),
]


@pytest.mark.parametrize("cls,kind,kw", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_layer_forward_matches_the_oracle(cls, kind, kw):
    layer = getattr(LA, cls)(**kw)
    opts = {k: getattr(layer, k) for k in RefConfig.__dataclass_fields__ if hasattr(layer, k)}
    ref = RefExtractor(RefConfig(**{**opts, "kind": kind}), np.float64)
    sr = layer.sampling_rate
    rng = np.random.RandomState(len(cls) + len(kw))
    x = (rng.rand(4, sr + 137).astype(np.float32) - 0.5)
    y = layer(torch.from_numpy(x).cuda())
    assert y.is_cuda and y.dtype == torch.float32
    got = y.cpu().numpy()
    for b in range(4):
        want = ref.extract(x[b])
        assert got[b].shape == want.shape
        if kind == "spectrogram":
            tol = np.broadcast_to(1e-4 * np.abs(want[:, 1:]).max(axis=1, keepdims=True) + 1e-7, want.shape).copy()
            if opts.get("use_energy"):
                tol[:, 0] = 1e-4  # column 0 holds the log-energy
            assert np.all(np.abs(got[b] - want) <= tol)
        elif kind == "mfcc":
            assert np.abs(got[b] - want).max() <= 1e-4 * np.abs(want).max()
            assert np.linalg.norm(got[b] - want) / np.linalg.norm(want) <= 1e-4
        else:
            err = np.abs(got[b] - want)
            # log domain: 1e-4 relative = 1e-4 absolute; bins at the float32 noise floor of their frame are bounded in the linear domain
            loud = want >= want.max(axis=1, keepdims=True) - 14.0
            assert err[loud].max() <= 2e-4, err[loud].max()
            assert np.linalg.norm(got[b] - want) / np.linalg.norm(want) <= 1e-4
    one = layer(torch.from_numpy(x[0]).cuda())
    assert torch.equal(one, y[0])
    assert layer.fft_length == ref.fft


def test_layer_argument_checks_and_pickle():
    import pickle

    layer = LA.HipWav2LogFilterBank()
    with pytest.raises(LA._lib.HipFeatError, match="no CPU fallback|'cuda"):
        layer(torch.zeros(2, 16000))
    with pytest.raises(NotImplementedError, match="inference-only"):
        layer(torch.zeros(2, 16000, device="cuda", requires_grad=True))
    with torch.no_grad():
        assert layer(torch.zeros(2, 16000, device="cuda", requires_grad=True)).shape == (2, 100, 80)
    with pytest.raises(ValueError):
        layer(torch.zeros(1, 50, device="cuda"))  # too short for one reflected frame
    again = pickle.loads(pickle.dumps(layer))
    assert again(torch.zeros(16000, device="cuda")).shape == (100, 80)
    assert layer(torch.zeros(0, 16000, device="cuda")).shape[0] == 0
