"""GPU: the instances of fft1024c / fft2048c that carry the mel schedule of the 80-filter Kaldi defaults as compile-time constants
(kernel_fft1024c.hpp, kernel_fft2048c.hpp) against the generic instances of the same kernels (HIPFEAT_NO_FIXED_SCHEDULE=1): the same
instructions on the same operands in the same order, so the outputs are bit-identical.  (Both are checked against the oracle of
Wav2LogFilterBank, lhotse/features/kaldi/layers.py:565-578, by the golden and ragged-batch tests of their sampling rates.)"""
import numpy as np
import pytest
import torch

from _golden import err_stats
from _golden import ref32
from _hip import make_hip
from oracle.kaldi_ref import RefConfig, RefExtractor

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr,kernel", [(22050, "fft1024c_kernel<20>"), (24000, "fft1024c_kernel<20>"), (32000, "fft1024c_kernel<26>"),
                                       (44100, "fft2048c_kernel<18,1>"), (48000, "fft2048c_kernel<19,0>")])
def test_fixed_schedule_instances_equal_the_generic_ones(sr, kernel, monkeypatch):
    fixed = make_hip("fbank", {"sampling_rate": sr})
    assert fixed.kernel_name.startswith(kernel) and "fixed-schedule" in fixed.kernel_name, fixed.kernel_name
    monkeypatch.setenv("HIPFEAT_NO_FIXED_SCHEDULE", "1")
    generic = make_hip("fbank", {"sampling_rate": sr})
    assert generic.kernel_name.startswith(kernel) and "fixed-schedule" not in generic.kernel_name, generic.kernel_name  # (builds the plan)
    monkeypatch.delenv("HIPFEAT_NO_FIXED_SCHEDULE")
    rng = np.random.default_rng(sr)
    waves = [torch.from_numpy((rng.standard_normal(n) * a).astype(np.float32)) for n, a in
             ((int(0.9 * sr), 0.1), (3 * sr + 17, 0.3), (sr // 20 + 1, 1e-3), (7 * sr + 5, 0.05))]
    o32, o64 = ref32(RefConfig(kind="fbank", sampling_rate=sr)), RefExtractor(RefConfig(kind="fbank", sampling_rate=sr), np.float64)
    for w, a, b in zip(waves, fixed.extract_batch(waves, sr), generic.extract_batch(waves, sr)):
        assert a.shape == b.shape and a.shape[1] == 80
        assert torch.equal(a, b)
        want, truth = o32.extract(w.numpy()), o64.extract(w.numpy())  # and both against the oracle of the reference at this rate
        st = err_stats(a.cpu().numpy(), want)
        assert a.shape == want.shape and st["rel_l2"] <= 1e-4 and st["max_abs"] <= max(2e-3, 3 * err_stats(want, truth)["max_abs"]), (sr, len(w), st)


def test_other_filterbanks_run_the_generic_instance():
    ex = make_hip("fbank", {"sampling_rate": 24000, "num_filters": 64})
    assert ex.kernel_name.startswith("fft1024c_kernel<20>") and "fixed-schedule" not in ex.kernel_name, ex.kernel_name


def test_librosa_default_runs_the_plain_fixed_schedule_instance(monkeypatch):
    """Round 4: the librosa default (n_fft 1024 @ 22.05 kHz, hop 256, 80 slaney filters; no DC removal, no pre-emphasis) has its own
    instance fft1024c_kernel<32, 16, 16, 8, PLAIN> at 3 waves/SIMD; other librosa configurations keep the generic one.  Same instructions
    on the same operands in the same order: bit-identical outputs (both are checked against the librosa oracle in test_gpu_librosa.py)."""
    import lhotse_amd as LA

    fixed = LA.HipLibrosaFbank()
    assert fixed.kernel_name.startswith("fft1024c_kernel<32>") and "fixed-schedule" in fixed.kernel_name and "waves=12" in fixed.kernel_name, fixed.kernel_name
    monkeypatch.setenv("HIPFEAT_NO_FIXED_SCHEDULE", "1")
    generic = LA.HipLibrosaFbank()
    assert generic.kernel_name.startswith("fft1024c_kernel<32>") and "fixed-schedule" not in generic.kernel_name, generic.kernel_name
    monkeypatch.delenv("HIPFEAT_NO_FIXED_SCHEDULE")
    rng = np.random.default_rng(5)
    waves = [torch.from_numpy((rng.standard_normal(n) * a).astype(np.float32)) for n, a in ((20000, 0.1), (3 * 22050 + 17, 0.3), (1500, 1e-3), (7 * 22050 + 5, 0.05), (220500, 0.2))]
    for w in waves:
        a, b = fixed.extract(w, 22050), generic.extract(w, 22050)
        assert a.shape == b.shape and a.shape[1] == 80 and torch.equal(torch.as_tensor(a), torch.as_tensor(b))
    other = LA.HipLibrosaFbank(LA.HipLibrosaFbankConfig(num_mel_bins=64))
    assert "fixed-schedule" not in other.kernel_name, other.kernel_name
