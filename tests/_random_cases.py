"""The seeded random sweep over the option surface shared by tests/test_gpu_random_configs.py (HIP vs ref32 / float64) and
tests/test_oracle.py (ref32 == the live reference, bit for bit, on the very same configurations)."""
import numpy as np


def random_case(rng):
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


def random_cases(n: int = 160):
    return [random_case(np.random.RandomState(1000 + i)) for i in range(n)]


def inputs_for(idx: int, cfg: dict):
    """The four inputs test_random_config_against_float64_oracle feeds configuration `idx`."""
    sr = cfg["sampling_rate"]
    rng = np.random.RandomState(idx)
    n_min = int(cfg["frame_length"] * sr) + 8
    lens = [sr, int(2.37 * sr) + 1, max(n_min, sr // 5), 4 * sr]
    return [(rng.rand(n).astype(np.float32) - 0.5) * s for n, s in zip(lens, (1.0, 0.05, 0.9, 0.5))]
