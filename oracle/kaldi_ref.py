"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement (numpy) of the reference's Kaldi-style feature pipeline
(lhotse/features/kaldi/layers.py + kaldi/extractors.py + utils.py).  It exists
only so that tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline``
leg can check / time the HIP path against something that runs without
/root/reference.  Nothing under lhotse_amd/ may import this module.

Parity status: PINNED.  tests/test_oracle.py checks every function here
against tests/golden/*.npz, which were produced by importing the reference
itself (oracle/make_golden.py) in the authoring container, and against the
known-answer table of SURVEY.md section 8c.

Two arithmetic modes:
  * ``dtype=np.float64`` is the reference's algorithm in double precision: the
    "truth" both float32 implementations are measured against,
  * ``dtype=np.float32`` follows the reference's op order in float32 EXCEPT
    in the FFT: numpy has no float32 FFT, so ``np.fft.rfft`` runs in float64
    and its result is rounded to complex64 ("float64 FFT rounded to
    float32").  That is NOT the reference's arithmetic (torch.fft.rfft on a
    float32 tensor, layers.py:32-42) and sits ~4x closer to float64 than
    the reference does (4e-4 vs 1.3e-3 ... 2.1e-3 max abs on 10 s
    of noise).  It remains the general-purpose checker (every kind, window,
    edge rule; rel-L2 differences to the reference ~1e-6), but `ref32` of the
    headline parity statements is oracle/kaldi_torch.reference_f32(), which
    is bit-equal to the live reference.

All citations are path:line relative to /root/reference/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

EPSILON = 1e-10  # lhotse/utils.py:49
LOG_EPSILON = math.log(EPSILON)  # lhotse/utils.py:50


@dataclass
class RefConfig:
    """Union of FbankConfig / MfccConfig / SpectrogramConfig / LogSpectrogramConfig
    (lhotse/features/kaldi/extractors.py:23-44, 155-178, 265-280, 375-390)."""

    kind: str = "fbank"  # "fbank" | "mfcc" | "spectrogram" | "log-spectrogram"
    sampling_rate: int = 16000
    frame_length: float = 0.025
    frame_shift: float = 0.01
    round_to_power_of_two: bool = True
    remove_dc_offset: bool = True
    preemph_coeff: float = 0.97
    window_type: str = "povey"
    dither: float = 0.0
    snip_edges: bool = False
    energy_floor: float = EPSILON
    raw_energy: bool = True
    use_energy: bool = False
    use_fft_mag: bool = False
    low_freq: float = 20.0
    high_freq: float = -400.0
    num_filters: int = 80
    norm_filters: bool = False
    torchaudio_compatible_mel_scale: bool = True
    num_ceps: int = 13
    cepstral_lifter: int = 22


# --------------------------------------------------------------------------
# frame-count contract
# --------------------------------------------------------------------------
def compute_num_frames_from_samples(num_samples: int, frame_shift: float, sampling_rate: int) -> int:
    """lhotse/utils.py:424-434"""
    hop = round(frame_shift * sampling_rate)
    return int((num_samples + hop // 2) // hop)


def window_sizes(cfg: RefConfig) -> Tuple[int, int, int]:
    """(N, shift, fft_length): layers.py:114-116 and :264-265, :951-957."""
    n = int(math.floor(cfg.frame_length * cfg.sampling_rate))
    shift = int(math.floor(cfg.frame_shift * cfg.sampling_rate))
    if cfg.round_to_power_of_two:
        fft = 1 if n == 0 else 2 ** (n - 1).bit_length()
    else:
        fft = n
    return n, shift, fft


def num_frames(num_samples: int, n: int, shift: int, snip_edges: bool) -> int:
    """layers.py:747-753"""
    if snip_edges:
        if num_samples < n:
            return 0
        return 1 + (num_samples - n) // shift
    return (num_samples + shift // 2) // shift


# --------------------------------------------------------------------------
# constants
# --------------------------------------------------------------------------
def frame_window(n: int, window_type: str, dtype=np.float64, blackman_coeff: float = 0.42) -> np.ndarray:
    """layers.py:921-940.  hann(periodic=False)[i] = 0.5 - 0.5 cos(2 pi i/(n-1))."""
    i = np.arange(n, dtype=np.float64)
    if window_type == "hanning":
        w = 0.5 - 0.5 * np.cos(2 * np.pi * i / (n - 1))
    elif window_type == "hamming":
        w = 0.54 - 0.46 * np.cos(2 * np.pi * i / (n - 1))
    elif window_type == "povey":
        w = (0.5 - 0.5 * np.cos(2 * np.pi * i / (n - 1))) ** 0.85
    elif window_type == "rectangular":
        w = np.ones(n)
    elif window_type == "blackman":
        a = 2 * np.pi / n  # NB: the reference divides by n, not n-1 (layers.py:932)
        w = blackman_coeff - 0.5 * np.cos(a * i) + (0.5 - blackman_coeff) * np.cos(2 * a * i)
    else:
        raise ValueError(f"Invalid window type: {window_type}")
    return w.astype(dtype)


def lin2mel(f):
    """layers.py:943-944"""
    return 1127.0 * np.log(1 + np.asarray(f, dtype=np.float64) / 700)


def kaldi_mel_banks(num_bins: int, fft: int, sample_freq: float, low: float, high: float, dtype=np.float64) -> np.ndarray:
    """layers.py:960-1017 followed by the zero column + transpose of :553.
    Returns (fft/2+1, num_bins).  Scalars (mel_low, delta, bin width) are float64 Python
    numbers in the reference; every tensor op runs in the tensor dtype (float32 there),
    which is what ``dtype`` selects here."""
    assert num_bins > 3
    assert fft % 2 == 0
    dt = np.dtype(dtype).type
    nyq = 0.5 * sample_freq
    if high <= 0.0:
        high += nyq
    assert 0.0 <= low < nyq and 0.0 < high <= nyq and low < high
    width = sample_freq / fft
    mlo, mhi = float(lin2mel(low)), float(lin2mel(high))
    delta = (mhi - mlo) / (num_bins + 1)
    j = np.arange(num_bins).astype(dtype)[None, :]
    left = dt(mlo) + j * dt(delta)
    center = dt(mlo) + (j + dt(1.0)) * dt(delta)
    right = dt(mlo) + (j + dt(2.0)) * dt(delta)
    hz = dt(width) * np.arange(fft // 2).astype(dtype)
    mel = (dt(1127.0) * np.log(dt(1) + hz / dt(700)))[:, None]  # lin2mel on a float tensor (layers.py:943, :1008)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    fb = np.maximum(dt(0.0), np.minimum(up, down))
    fb = np.concatenate([fb, np.zeros((1, num_bins), dtype=dtype)], axis=0)
    return fb.astype(dtype)


def htk_mel_banks(num_filters: int, fft: int, sampling_rate: int, low: float, high: Optional[float], norm_filters: bool, dtype=np.float64) -> np.ndarray:
    """layers.py:873-907 (``torchaudio_compatible_mel_scale=False``).
    NB the reference maps bin j to np.linspace(0, sr, fft)[j], i.e. a bin width of
    sr/(fft-1), and builds the matrix in float32 from float64 mels."""
    if high is None or high == 0:
        high = sampling_rate / 2
    if high < 0:
        high = sampling_rate / 2 + high
    melfc = np.linspace(lin2mel(low), lin2mel(high), num_filters + 2)
    mels = lin2mel(np.linspace(0, sampling_rate, fft))
    b = np.zeros((fft // 2 + 1, num_filters), dtype=np.float64)
    mj = mels[: fft // 2]
    for k in range(num_filters):
        l, c, r = melfc[k], melfc[k + 1], melfc[k + 2]
        inside = (l < mj) & (mj < r)
        rising = inside & (mj <= c)
        falling = inside & (mj > c)
        b[: fft // 2, k][rising] = (mj[rising] - l) / (c - l)
        b[: fft // 2, k][falling] = (r - mj[falling]) / (r - c)
    b = b.astype(np.float32)  # the reference stores it in float32 (layers.py:891)
    if norm_filters:
        b = b / np.sum(b, axis=0, keepdims=True)
    return b.astype(dtype)


def mel_matrix(cfg: RefConfig, dtype=np.float64) -> np.ndarray:
    """layers.py:545-563"""
    _, _, fft = window_sizes(cfg)
    if cfg.torchaudio_compatible_mel_scale:
        return kaldi_mel_banks(cfg.num_filters, fft, cfg.sampling_rate, cfg.low_freq, cfg.high_freq, dtype)
    return htk_mel_banks(cfg.num_filters, fft, cfg.sampling_rate, cfg.low_freq, cfg.high_freq, cfg.norm_filters, dtype)


def dct_matrix(num_ceps: int, num_filters: int, dtype=np.float64) -> np.ndarray:
    """layers.py:697-706.  Shape (num_filters, num_ceps)."""
    n = np.arange(num_filters, dtype=np.float64)[:, None]
    k = np.arange(num_ceps, dtype=np.float64)[None, :]
    d = np.cos(math.pi / num_filters * (n + 0.5) * k)
    d[:, 0] *= 1.0 / math.sqrt(2.0)
    d *= math.sqrt(2.0 / num_filters)
    return d.astype(dtype)


def lifter(num_ceps: int, q: int, dtype=np.float64) -> np.ndarray:
    """layers.py:681-695"""
    if q == 0:
        return np.ones(num_ceps, dtype=dtype)
    return (1 + 0.5 * q * np.sin(math.pi * np.arange(num_ceps, dtype=np.float64) / q)).astype(dtype)


# --------------------------------------------------------------------------
# framing
# --------------------------------------------------------------------------
def frame_indices(num_samples: int, n: int, shift: int, snip_edges: bool, padded_len: Optional[int] = None) -> np.ndarray:
    """Index restatement of layers.py:727-772.

    The reference concatenates flip(x[:npad_left]) | x | flip(x[-npad_right:]) and takes
    a strided view; frame t, tap i therefore reads original index
    ``j = shift*t - npad_left + i`` with ``j<0 -> -j-1`` and ``j>=P -> 2P-1-j``,
    P being the row length the reflection was taken on.  For a single waveform
    P == num_samples.  In a zero-padded batch (extractors.py:531) P is the longest
    item's length and indices in [num_samples, P) read zeros; those are returned as -1.
    """
    p = num_samples if padded_len is None else padded_len
    if snip_edges:
        t = num_frames(p, n, shift, True)
        j = shift * np.arange(t)[:, None] + np.arange(n)[None, :]
    else:
        t = num_frames(p, n, shift, False)
        npad_left = (n - shift) // 2
        npad_right = (t - 1) * shift + n - p - npad_left
        if t > 0 and (npad_left > p or npad_right > p):
            # the reference raises here: reflection needs more samples than exist
            raise ValueError(f"waveform of {p} samples is too short for reflect padding ({npad_left}, {npad_right})")
        j = shift * np.arange(t)[:, None] - npad_left + np.arange(n)[None, :]
        j = np.where(j < 0, -j - 1, j)
        j = np.where(j >= p, 2 * p - 1 - j, j)
    j = np.where(j >= num_samples, -1, j)
    return j


def frames_from_wave(x: np.ndarray, n: int, shift: int, snip_edges: bool, padded_len: Optional[int] = None) -> np.ndarray:
    idx = frame_indices(len(x), n, shift, snip_edges, padded_len)
    out = x[np.maximum(idx, 0)]
    out = np.where(idx < 0, x.dtype.type(0), out)
    return out


def log_energy(frames: np.ndarray, energy_floor: float) -> np.ndarray:
    """layers.py:859-870"""
    dt = frames.dtype.type
    e = np.log((frames**2).sum(-1) + dt(1e-15))
    if energy_floor > 0.0:
        e = np.maximum(e, dt(math.log(energy_floor)))
    return e


def preprocess_frames(frames: np.ndarray, cfg: RefConfig, window: np.ndarray, fft: int, want_energy: bool):
    """layers.py:151-187: DC removal -> (raw log-energy) -> pre-emphasis -> window -> zero pad."""
    dt = frames.dtype.type
    if cfg.remove_dc_offset:
        frames = frames - frames.mean(axis=-1, keepdims=True, dtype=frames.dtype)
    log_e = None
    if want_energy and cfg.raw_energy:
        log_e = log_energy(frames, cfg.energy_floor)
    if cfg.preemph_coeff != 0.0:
        prev = np.concatenate([frames[..., :1], frames[..., :-1]], axis=-1)  # replicate pad (layers.py:166)
        frames = frames - dt(cfg.preemph_coeff) * prev
    frames = frames * window
    n = frames.shape[-1]
    if fft != n:
        frames = np.concatenate([frames, np.zeros(frames.shape[:-1] + (fft - n,), dtype=frames.dtype)], axis=-1)
    if want_energy and not cfg.raw_energy:
        log_e = log_energy(frames, cfg.energy_floor)
    return frames, log_e


# --------------------------------------------------------------------------
# the four feature types
# --------------------------------------------------------------------------
class RefExtractor:
    """Restates Wav2Spec / Wav2LogSpec / Wav2LogFilterBank / Wav2MFCC (layers.py:336-724)."""

    def __init__(self, cfg: RefConfig, dtype=np.float32):
        if cfg.dither != 0.0:
            raise ValueError("oracle is deterministic: dither must be 0 (layers.py:191-193 draws randn)")
        self.cfg = cfg
        self.dtype = np.dtype(dtype)
        self.n, self.shift, self.fft = window_sizes(cfg)
        self.window = frame_window(self.n, cfg.window_type, dtype)
        self.eps = self.dtype.type(np.finfo(np.float32).eps)  # layers.py:536-538
        if cfg.kind in ("fbank", "mfcc"):
            self.fb = mel_matrix(cfg, dtype)
        if cfg.kind == "mfcc":
            self.dct = dct_matrix(cfg.num_ceps, cfg.num_filters, dtype)
            self.lifter = lifter(cfg.num_ceps, cfg.cepstral_lifter, dtype)

    @property
    def feature_dim(self) -> int:
        c = self.cfg
        if c.kind == "fbank":
            return c.num_filters + (1 if c.use_energy else 0)  # layers.py:575-576
        if c.kind == "mfcc":
            return c.num_ceps
        return self.fft // 2 + 1

    def num_frames(self, num_samples: int) -> int:
        return num_frames(num_samples, self.n, self.shift, self.cfg.snip_edges)

    def _spec(self, frames: np.ndarray) -> np.ndarray:
        """layers.py:32-42: rfft then |X|^2 (or |X|).  NB float32 mode: numpy's rfft computes in float64; the result is rounded to
        complex64 (module docstring) -- closer to float64 than the reference's float32 FFT is."""
        X = np.fft.rfft(frames, axis=-1)
        if self.dtype == np.float32:
            X = X.astype(np.complex64)
        mag = np.abs(X)
        return mag if self.cfg.use_fft_mag else mag**2

    def from_frames(self, frames: np.ndarray) -> np.ndarray:
        c = self.cfg
        dt = self.dtype.type
        want_e = c.use_energy
        # NB Wav2MFCC(use_energy=True) raises a shape error upstream (layers.py:721-722 assigns a (B, T) tensor to
        # mfcc[:, 0]).  Its evident intent -- and Kaldi's definition (feature-mfcc.cc: the log-energy
        # replaces C0 after liftering) -- is restated below; this one option has no reference output to pin against.
        y, log_e = preprocess_frames(frames.astype(self.dtype), c, self.window, self.fft, want_e)
        p = self._spec(y)
        if c.kind == "spectrogram":  # layers.py:392-402
            if want_e:
                p[..., 0] = log_e
            return p
        if c.kind == "log-spectrogram":  # layers.py:461-473
            p = np.log(p + dt(1e-15))
            if want_e:
                p[..., 0] = log_e
            return p
        mel = p @ self.fb  # layers.py:571
        mel = np.log(np.maximum(mel, self.eps))  # layers.py:572
        if c.kind == "fbank":
            if want_e:
                mel = np.concatenate([log_e[..., None], mel], axis=-1)
            return mel
        out = mel @ self.dct  # layers.py:716
        if c.cepstral_lifter > 0:
            out = out * self.lifter
        if want_e:
            out = out.copy()
            out[..., 0] = log_e  # layers.py:721-722 as intended
        return out

    def extract(self, x: np.ndarray, padded_len: Optional[int] = None) -> np.ndarray:
        """One waveform (T,) -> (num_frames, feature_dim); Fbank.extract (extractors.py:92-115)."""
        x = np.asarray(x).reshape(-1).astype(self.dtype)
        frames = frames_from_wave(x, self.n, self.shift, self.cfg.snip_edges, padded_len)
        if frames.shape[0] == 0:
            return np.zeros((0, self.feature_dim), dtype=self.dtype)
        return self.from_frames(frames)

    def extract_batch(self, waves: Sequence[np.ndarray], edge_rule: str = "reflect") -> List[np.ndarray]:
        """edge_rule="reflect": every item framed on its own (Fbank.extract, kaldifeat).
        edge_rule="batch_zero_pad": extractors.py:485-554 -- items are zero padded to the
        longest, framed together, then cut to compute_num_frames_from_samples."""
        if edge_rule == "reflect":
            return [self.extract(w) for w in waves]
        assert edge_rule == "batch_zero_pad"
        pmax = max(len(np.asarray(w).reshape(-1)) for w in waves)
        out = []
        for w in waves:
            w = np.asarray(w).reshape(-1)
            f = self.extract(w, padded_len=pmax)
            t = compute_num_frames_from_samples(len(w), self.cfg.frame_shift, self.cfg.sampling_rate)
            out.append(f[:t])
        return out


def tripwire_signal(num: int = 16000, dtype=np.float32) -> np.ndarray:
    """SURVEY.md section 8c known-answer input."""
    n = np.arange(num, dtype=np.float64)
    return (0.5 * np.sin(2 * np.pi * 440 * n / 16000) + 0.25 * np.sin(2 * np.pi * 3000 * n / 16000)).astype(dtype)
