"""GPU: seeded random sweep over the option surface -- whatever kernel the plan selects (fft512b,
fft256, wave, generic radix-2 / direct DFT) must agree with the float64 oracle to the float32 noise floor of the
reference's own arithmetic (rel-L2 <= max(1e-4, 3 x floor))."""
import warnings

import numpy as np
import pytest

import lhotse_amd as LA
from oracle import kaldi_ref as K

pytestmark = pytest.mark.gpu
TABLE = {"fbank": (LA.HipFbank, LA.HipFbankConfig), "mfcc": (LA.HipMfcc, LA.HipMfccConfig), "spectrogram": (LA.HipSpectrogram, LA.HipSpectrogramConfig),
         "log-spectrogram": (LA.HipLogSpectrogram, LA.HipLogSpectrogramConfig)}


def _random_case(rng):
    kind = rng.choice(["fbank", "fbank", "mfcc", "spectrogram", "log-spectrogram"])
    sr = int(rng.choice([8000, 16000, 16000, 22050, 24000, 32000, 44100, 48000]))
    cfg = dict(sampling_rate=sr)
    cfg["frame_length"] = float(rng.choice([0.025, 0.025, 0.02, 0.032, 0.016]))
    cfg["frame_shift"] = float(rng.choice([0.01, 0.01, 0.0125, 0.008]))
    if cfg["frame_shift"] > cfg["frame_length"]:
        cfg["frame_shift"] = cfg["frame_length"] / 2
    cfg["window_type"] = str(rng.choice(["povey", "povey", "hanning", "hamming", "rectangular", "blackman"]))
    cfg["remove_dc_offset"] = bool(rng.rand() < 0.8)
    cfg["preemph_coeff"] = float(rng.choice([0.97, 0.97, 0.0, 0.9]))
    cfg["snip_edges"] = bool(rng.rand() < 0.2)
    cfg["round_to_power_of_two"] = bool(rng.rand() < 0.85)
    if kind in ("fbank", "mfcc"):
        cfg["num_filters"] = int(rng.choice([23, 40, 64, 80, 128]))
        cfg["low_freq"] = float(rng.choice([20.0, 0.0, 60.0]))
        cfg["high_freq"] = float(rng.choice([-400.0, 0.0, -100.0]))
        if rng.rand() < 0.15:
            cfg["torchaudio_compatible_mel_scale"] = False
    if kind == "mfcc":
        cfg["num_ceps"] = int(min(cfg["num_filters"], rng.choice([13, 20, 23])))
        cfg["cepstral_lifter"] = int(rng.choice([22, 0]))
    if kind != "mfcc" and rng.rand() < 0.2:
        cfg["use_energy"] = True
        cfg["raw_energy"] = bool(rng.rand() < 0.5)
        cfg["energy_floor"] = float(rng.choice([1e-10, 1e-3]))
    if kind != "mfcc" and rng.rand() < 0.15:
        cfg["use_fft_mag"] = True
    return kind, cfg


CASES = [_random_case(np.random.RandomState(1000 + i)) for i in range(160)]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_random_config_against_float64_oracle(idx):
    kind, cfg = CASES[idx]
    sr = cfg["sampling_rate"]
    rng = np.random.RandomState(idx)
    n_min = int(cfg["frame_length"] * sr) + 8
    lens = [sr, int(2.37 * sr) + 1, max(n_min, sr // 5), 4 * sr]
    xs = [(rng.rand(n).astype(np.float32) - 0.5) * s for n, s in zip(lens, (1.0, 0.05, 0.9, 0.5))]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = TABLE[kind][0](TABLE[kind][1](**cfg))
    fields = {k: v for k, v in cfg.items() if k in K.RefConfig.__dataclass_fields__}
    if kind == "mfcc":
        fields.setdefault("num_filters", 23)
    if kind in ("fbank", "mfcc") and K.window_sizes(K.RefConfig(kind=kind, **fields))[2] % 2:
        # an odd fft length has no mel filterbank in the reference either (layers.py:977 asserts)
        with pytest.raises(ValueError, match="must be even"):
            ex.extract_batch(xs, sr)
        return
    ref64 = K.RefExtractor(K.RefConfig(kind=kind, **fields), np.float64)
    ref32 = K.RefExtractor(K.RefConfig(kind=kind, **fields), np.float32)
    outs = ex.extract_batch(xs, sr)
    assert len(outs) == len(xs)
    for x, got in zip(xs, outs):
        truth = ref64.extract(x)
        want = ref32.extract(x)
        assert got.shape == truth.shape, (ex.kernel_name, kind, cfg, got.shape, truth.shape)
        den = max(np.linalg.norm(truth), 1e-30)
        floor = np.linalg.norm(want - truth) / den
        rel = np.linalg.norm(got - truth) / den
        assert rel <= max(1e-4, 3 * floor), (ex.kernel_name, kind, cfg, len(x), rel, floor)
