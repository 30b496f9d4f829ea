"""
TEST INFRASTRUCTURE -- the table of golden cases.

Each case = feature kind + config overrides + list of (signal kind, num_samples, seed)
+ call mode.  oracle/make_golden.py runs the REFERENCE on every case and stores
its outputs in tests/golden/<name>.npz; tests re-create the inputs from the same
table and compare the oracle restatement (CPU) and the HIP path (GPU) to them.

Call modes (reference entry point exercised):
  "extract"        Fbank.extract per item               (extractors.py:92-115)
  "batch"          Fbank.extract_batch(list)            (extractors.py:485-554, zero-padded batch)
  "batch_lengths"  Fbank.extract_batch(padded, lengths) (same, padded tensor + lengths form)
"""
from __future__ import annotations

from typing import Any, Dict, List

U = "uniform"


def _c(name: str, kind: str, cfg: Dict[str, Any], inputs: List, mode: str = "extract", rows: int = 0, sample: int = 0) -> Dict[str, Any]:
    # rows > 0: only the first/last `rows` rows of each output are stored (long inputs)
    # sample > 0: `sample` rows per output at seeded random positions (plus the first and last four) are stored
    return dict(name=name, kind=kind, cfg=cfg, inputs=[list(i) for i in inputs], mode=mode, rows=rows, sample=sample)


def _realistic(sr: int, base_seed: int) -> List:
    """20 inputs per sampling rate (VERDICT r2 task 4): speech-like signals of 0.6-3 s, a third of them 60 dB down."""
    out = []
    for i in range(20):
        kind = ("voiced", "voiced", "speechlike", "voiced_quiet", "speechlike_quiet", "voiced")[i % 6]
        secs = 0.6 + 0.12 * ((i * 7) % 21)
        out.append((kind, int(secs * sr) + 13 * i, base_seed + i))
    return out


CASES: List[Dict[str, Any]] = [
    # ---- headline configs ----
    _c("fbank80_tone", "fbank", {}, [("tone", 16000, 0)]),
    _c("fbank80_uniform", "fbank", {}, [(U, 16000, 1)]),
    _c("fbank80_gauss", "fbank", {}, [("gauss", 16000, 2)]),
    _c("fbank80_speechlike", "fbank", {}, [("speechlike", 24000, 3)]),
    _c("fbank80_10s", "fbank", {}, [(U, 160000, 4)], rows=8),
    _c("fbank40", "fbank", {"num_filters": 40}, [(U, 16000, 5)]),
    _c("fbank23", "fbank", {"num_filters": 23}, [(U, 8000, 6)]),
    _c("mfcc_default", "mfcc", {}, [(U, 16000, 7), ("tone", 16000, 0)]),
    _c("mfcc40x40", "mfcc", {"num_filters": 40, "num_ceps": 40}, [(U, 16000, 8), ("tone", 16000, 0)]),
    # NB cepstral_lifter=0 cannot be a golden: the reference crashes building it
    # (layers.py:691-692 returns the int 1, nn.Parameter(1) raises).
    _c("spectrogram", "spectrogram", {}, [(U, 4000, 10), ("tone", 4000, 0)]),
    _c("logspectrogram", "log-spectrogram", {}, [(U, 4000, 11), ("tone", 4000, 0)]),
    # ---- length edge cases (first valid length is 140: SURVEY Q6) ----
    _c(
        "fbank_lengths",
        "fbank",
        {},
        [(U, n, 20 + i) for i, n in enumerate([140, 159, 160, 161, 239, 240, 241, 399, 400, 401, 479, 480, 481, 1000, 4321])],
    ),
    _c("fbank_long", "fbank", {}, [(U, 100050, 40)], rows=6),
    # ---- window types ----
    _c("win_hamming", "fbank", {"window_type": "hamming"}, [(U, 8000, 50)]),
    _c("win_hanning", "fbank", {"window_type": "hanning"}, [(U, 8000, 51)]),
    _c("win_rect", "fbank", {"window_type": "rectangular"}, [(U, 8000, 52)]),
    _c("win_blackman", "fbank", {"window_type": "blackman"}, [(U, 8000, 53)]),
    # ---- sampling rates / fft sizes ----
    _c("sr8k", "fbank", {"sampling_rate": 8000, "num_filters": 40}, [(U, 8000, 60)]),  # N=200 fft=256
    _c("sr22k", "fbank", {"sampling_rate": 22050}, [(U, 11025, 61)]),  # N=551 fft=1024 shift=220
    _c("sr44k", "fbank", {"sampling_rate": 44100}, [(U, 22050, 62)]),  # N=1102 fft=2048 shift=441
    _c("sr48k", "fbank", {"sampling_rate": 48000, "num_filters": 128}, [(U, 24000, 63)]),  # N=1200 fft=2048
    _c("sr16k_20ms", "fbank", {"frame_length": 0.02}, [(U, 8000, 64)]),  # N=320 fft=512
    _c("sr16k_32ms_8ms", "fbank", {"frame_length": 0.032, "frame_shift": 0.008}, [(U, 8000, 65)]),  # N=512 fft=512 shift=128
    _c("nopow2", "fbank", {"round_to_power_of_two": False}, [(U, 4000, 66)]),  # fft=400 (non power of two)
    _c("nopow2_8k_mfcc", "mfcc", {"round_to_power_of_two": False, "sampling_rate": 8000}, [(U, 4000, 67)]),  # fft=200
    # ---- switches ----
    _c("snip_edges", "fbank", {"snip_edges": True}, [(U, 16000, 70), (U, 400, 71), (U, 559, 72), (U, 560, 73)]),
    _c("no_dc", "fbank", {"remove_dc_offset": False}, [("speechlike", 8000, 74)]),
    _c("no_preemph", "fbank", {"preemph_coeff": 0.0}, [(U, 8000, 75)]),
    _c("no_dc_no_preemph", "mfcc", {"remove_dc_offset": False, "preemph_coeff": 0.0}, [(U, 8000, 76)]),
    _c("fft_mag", "fbank", {"use_fft_mag": True}, [(U, 8000, 77)]),
    _c("fft_mag_spec", "spectrogram", {"use_fft_mag": True}, [(U, 4000, 78)]),
    _c("energy_raw", "fbank", {"use_energy": True}, [(U, 8000, 79), ("zeros", 4000, 0)]),
    _c("energy_windowed", "fbank", {"use_energy": True, "raw_energy": False}, [(U, 8000, 80)]),
    _c("energy_floor", "fbank", {"use_energy": True, "energy_floor": 1.0}, [(U, 8000, 81)]),
    _c("energy_spec", "spectrogram", {"use_energy": True}, [(U, 4000, 82)]),
    _c("energy_logspec", "log-spectrogram", {"use_energy": True, "raw_energy": False}, [(U, 4000, 83)]),
    _c("mel_lo0_hi0", "fbank", {"low_freq": 0.0, "high_freq": 0.0}, [(U, 8000, 84)]),
    _c("mel_hi7000", "fbank", {"low_freq": 100.0, "high_freq": 7000.0, "num_filters": 64}, [(U, 8000, 85)]),
    _c("mel_htk", "fbank", {"torchaudio_compatible_mel_scale": False}, [(U, 8000, 86)]),
    _c("mel_htk_norm", "fbank", {"torchaudio_compatible_mel_scale": False, "norm_filters": True, "num_filters": 40}, [(U, 8000, 87)]),
    # ---- degenerate signals ----
    _c("zeros", "fbank", {}, [("zeros", 4000, 0)]),
    _c("dc", "fbank", {}, [("dc", 4000, 0)]),
    _c("impulse", "fbank", {}, [("impulse", 4000, 0)]),
    _c("zeros_logspec", "log-spectrogram", {}, [("zeros", 2000, 0)]),
    # ---- batch semantics (zero-padded batch) ----
    _c("batch_list", "fbank", {}, [(U, 16000, 90), (U, 10000, 91), (U, 12345, 92), (U, 15999, 93)], mode="batch"),
    _c("batch_equal", "fbank", {}, [(U, 8000, 94), (U, 8000, 95), (U, 8000, 96)], mode="batch"),
    _c("batch_lengths", "mfcc", {"num_filters": 40, "num_ceps": 40}, [(U, 9000, 97), (U, 4000, 98), (U, 8880, 99)], mode="batch_lengths"),
    _c("batch_single", "fbank", {}, [(U, 5000, 100)], mode="batch"),
    # fractional hop: 12.5 ms @ 22.05 kHz frames with floor() = 275 samples, but compute_num_frames_from_samples slices the
    # items of a zero-padded batch with round() = 276 (lhotse/utils.py:424-434): 9213 and 15264 samples give one row less
    _c("batch_fractional_hop", "fbank", {"sampling_rate": 22050, "frame_shift": 0.0125, "num_filters": 40},
       [(U, 22050, 101), (U, 9213, 102), (U, 15264, 103)], mode="batch"),
    # ---- round 3: realistic signals through the DEFAULT-rate instance of every wave-autonomous kernel, full 10 s cuts ----
    _c("fbank80_10s_full", "fbank", {}, [(U, 160000, 110)]),                       # every row of a 10 s cut (fft512c<13>)
    _c("voiced_10s_full", "fbank", {}, [("voiced", 160000, 111)]),
    _c("voiced_quiet_10s", "fbank", {}, [("voiced_quiet", 160000, 112)], sample=96),
    _c("real16k", "fbank", {}, _realistic(16000, 200), sample=40),                  # fft512c<13>
    _c("real8k", "fbank", {"sampling_rate": 8000}, _realistic(8000, 300), sample=40),    # fft256c<13>
    _c("real24k", "fbank", {"sampling_rate": 24000}, _realistic(24000, 400), sample=40),  # fft1024c<20>
    _c("real48k", "fbank", {"sampling_rate": 48000}, _realistic(48000, 500), sample=40),  # fft2048c<19,0>
    _c("real16k_mfcc", "mfcc", {"num_filters": 40, "num_ceps": 40}, _realistic(16000, 600)[:8], sample=40),
    _c("voiced_10s_8k", "fbank", {"sampling_rate": 8000}, [("voiced", 80000, 113), ("voiced_quiet", 80000, 114)], sample=96),
    _c("voiced_10s_24k", "fbank", {"sampling_rate": 24000}, [("voiced", 240000, 115), ("speechlike_quiet", 240000, 116)], sample=96),
    _c("voiced_10s_48k", "fbank", {"sampling_rate": 48000}, [("voiced", 480000, 117), ("voiced_quiet", 480000, 118)], sample=96),
]


def case_by_name(name: str) -> Dict[str, Any]:
    for c in CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)
