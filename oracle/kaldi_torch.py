"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

`ref32` of every parity statement of this repository (round 5) AND the CPU baseline of bench.py: the reference's CPU Fbank /
MFCC / Speed path restated with the SAME library calls the reference makes
(torch.as_strided framing on a flip/cat-padded waveform, torch.mean, F.pad(replicate) pre-emphasis, window multiply,
zero pad, torch.fft.rfft, abs()**2, matmul with the mel matrix, log) -- lhotse/features/kaldi/layers.py:151-187,
:565-578, :727-772 as called by Fbank.extract (lhotse/features/kaldi/extractors.py:92-115).  It exists because
/root/reference cannot travel to the GPU box, where "Lhotse's existing CPU Fbank path" has to be timed next to the GPU
path; being the same sequence of ATen kernels, its speed is the reference's (BASELINE.md section 2 probe: ~150 cuts/s per
single-threaded process in the authoring container).

Why this module and not oracle/kaldi_ref.py is "the reference's float32 arithmetic": numpy has no float32 FFT (np.fft.rfft
computes in float64; kaldi_ref's float32 mode rounds that result to complex64), so kaldi_ref's float32 output is CLOSER to
float64 than the reference ever is -- on 16 x 10 s of U(-0.5, 0.5): max|reference - f64| 1.3e-3 ... 2.1e-3, max|kaldi_ref32 - f64|
4e-4.  torch.fft.rfft on a float32 tensor IS what layers.py:32-42 runs; `reference_f32()` below is therefore what
bench.py's parity legs, tests/test_gpu_parity.py and __graft_entry__.smoke() take `ref32` from.  kaldi_ref's float64 mode stays
the truth both float32 implementations are measured against.

Parity status: PINNED -- tests/test_oracle.py::test_torch_baseline_equals_golden checks it against the reference's own
outputs (tests/golden/fbank_default*.npz) and tests/test_oracle.py::test_torch_ref32_is_the_live_reference_bit_for_bit
(authoring container) against the live reference on 16 full-size cuts per extractor: array_equal.
"""
from __future__ import annotations

import numpy as np
import torch

from . import kaldi_ref as K


class TorchFbank:
    """Default FbankConfig (16 kHz, 25/10 ms, povey, 80 mels, snip_edges=False) -- the BASELINE configuration."""

    def __init__(self, cfg: K.RefConfig = None, device: str = "cpu"):
        cfg = cfg or K.RefConfig(kind="fbank")
        self.device = torch.device(device)
        assert cfg.kind == "fbank" and not cfg.snip_edges and not cfg.use_energy
        self.cfg = cfg
        self.n, self.shift, self.fft = K.window_sizes(cfg)
        assert cfg.window_type == "povey"
        self.window = torch.hann_window(self.n, periodic=False).pow(0.85)  # layers.py:929 -- torch's own float32 kernels
        self.fb = torch.from_numpy(np.ascontiguousarray(K.mel_matrix(cfg, np.float32).astype(np.float32)))  # (fft/2+1, M)
        self.eps = torch.tensor(torch.finfo(torch.float32).eps)
        if self.device.type != "cpu":  # the reference's own GPU mode: the same ops on device tensors (FbankConfig(device="cuda"))
            self.window, self.fb, self.eps = self.window.to(self.device), self.fb.to(self.device), self.eps.to(self.device)

    def strided(self, x: torch.Tensor) -> torch.Tensor:
        """layers.py:727-772 (snip_edges=False): reflect by flip/cat, then an as_strided view."""
        S = x.shape[-1]
        T = (S + self.shift // 2) // self.shift
        npad_left = (self.n - self.shift) // 2
        npad_right = (T - 1) * self.shift + self.n - S - npad_left
        pad_left = torch.flip(x[:, :npad_left], (1,))
        if npad_right >= 0:
            pad_right = torch.flip(x[:, S - npad_right :], (1,))
            x = torch.cat((pad_left, x, pad_right), dim=1)
        else:
            x = torch.cat((pad_left, x[:, :npad_right]), dim=1)
        return x.as_strided((x.shape[0], T, self.n), (x.stride(0), self.shift * x.stride(1), x.stride(1)))

    @torch.no_grad()
    def forward_batch(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S) equal-length batch on self.device -> (B, T, M): Wav2LogFilterBank.forward on a batch (layers.py:322-324)."""
        c = self.cfg
        x = self.strided(x)
        if c.remove_dc_offset:
            x = x - torch.mean(x, dim=2, keepdim=True)
        if c.preemph_coeff != 0.0:
            off = torch.nn.functional.pad(x, (1, 0), mode="replicate")
            x = x - c.preemph_coeff * off[:, :, :-1]
        x = x * self.window
        if self.fft != self.n:
            x = torch.nn.functional.pad(x.unsqueeze(1), [0, self.fft - self.n], mode="constant", value=0.0).squeeze(1)
        pow_spec = torch.fft.rfft(x, dim=-1).abs() ** 2
        return torch.max(torch.matmul(pow_spec, self.fb), self.eps).log()

    @torch.no_grad()
    def extract(self, samples: np.ndarray) -> np.ndarray:
        c = self.cfg
        x = self.strided(torch.from_numpy(np.asarray(samples, dtype=np.float32)).reshape(1, -1))
        if c.remove_dc_offset:
            x = x - torch.mean(x, dim=2, keepdim=True)
        if c.preemph_coeff != 0.0:
            off = torch.nn.functional.pad(x, (1, 0), mode="replicate")
            x = x - c.preemph_coeff * off[:, :, :-1]
        x = x * self.window
        if self.fft != self.n:
            x = torch.nn.functional.pad(x.unsqueeze(1), [0, self.fft - self.n], mode="constant", value=0.0).squeeze(1)
        pow_spec = torch.fft.rfft(x, dim=-1).abs() ** 2
        mel = torch.matmul(pow_spec, self.fb)
        return torch.max(mel, self.eps).log()[0].numpy()


class TorchMfcc(TorchFbank):
    """Mfcc.extract (lhotse/features/kaldi/extractors.py:222-245) = Wav2MFCC (layers.py:708-724): the log-mel above with the MFCC's
    own filterbank, then `@ dct`, then `* lifter` -- the CPU baseline of bench.py --config mfcc40_libri.  Pinned by
    tests/test_oracle.py::test_torch_mfcc_baseline_equals_golden against the reference's own output (golden `mfcc40x40`)."""

    def __init__(self, num_filters: int = 40, num_ceps: int = 40, cepstral_lifter: int = 22, cfg: K.RefConfig = None):
        cfg = cfg or K.RefConfig(kind="mfcc", num_filters=num_filters, num_ceps=num_ceps, cepstral_lifter=cepstral_lifter)
        num_filters, num_ceps, cepstral_lifter = cfg.num_filters, cfg.num_ceps, cfg.cepstral_lifter
        assert cfg.kind == "mfcc" and not cfg.snip_edges and not cfg.use_energy and cfg.window_type == "povey"
        assert cfg.remove_dc_offset and cfg.preemph_coeff != 0.0
        self.device = torch.device("cpu")
        self.cfg = cfg
        self.n, self.shift, self.fft = K.window_sizes(cfg)
        self.window = torch.hann_window(self.n, periodic=False).pow(0.85)
        self.fb = torch.from_numpy(np.ascontiguousarray(K.mel_matrix(cfg, np.float32).astype(np.float32)))
        self.eps = torch.tensor(torch.finfo(torch.float32).eps)
        # layers.py:697-706 / :681-695 with torch's own float32 cos / sin on float32 arguments, as the reference builds them (the float64
        # tables of kaldi_ref.dct_matrix rounded to float32 differ in the last bit, which moves cepstra by up to 3e-4)
        import math

        n = torch.arange(float(num_filters)).unsqueeze(1)
        k = torch.arange(float(num_ceps))
        dct = torch.cos(math.pi / float(num_filters) * (n + 0.5) * k)  # (M, C)
        dct[:, 0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(num_filters))
        self.dct = dct
        self.lifter = 1 + 0.5 * cepstral_lifter * torch.sin(math.pi * torch.arange(num_ceps, dtype=torch.float32) / cepstral_lifter) if cepstral_lifter else None

    @torch.no_grad()
    def extract(self, samples: np.ndarray) -> np.ndarray:
        c = self.cfg
        x = self.strided(torch.from_numpy(np.asarray(samples, dtype=np.float32)).reshape(1, -1))
        x = x - torch.mean(x, dim=2, keepdim=True)
        off = torch.nn.functional.pad(x, (1, 0), mode="replicate")
        x = x - c.preemph_coeff * off[:, :, :-1]
        x = x * self.window
        x = torch.nn.functional.pad(x.unsqueeze(1), [0, self.fft - self.n], mode="constant", value=0.0).squeeze(1)
        pow_spec = torch.fft.rfft(x, dim=-1).abs() ** 2
        mel = torch.max(torch.matmul(pow_spec, self.fb), self.eps).log()
        mfcc = torch.matmul(mel, self.dct)  # layers.py:717
        if c.cepstral_lifter > 0:
            mfcc *= self.lifter             # layers.py:718-719
        return mfcc[0].numpy()


class TorchSpeed:
    """Speed.__call__ (lhotse/augmentation/torchaudio.py:26-83) -> ResampleTensor.forward (augmentation/resample.py:284-315) with the
    reference's own torch calls: F.pad, conv1d(stride=orig) with the cached sinc kernel, transpose/reshape, trim to
    ceil(new * length / orig).  The kernel values come from oracle/resample_ref.sinc_kernel (bit-identical to the reference's cached
    buffer: tests/test_resample_oracle.py); the CPU baseline of bench.py --config onthefly."""

    def __init__(self, sampling_rate: int, factor: float):
        from . import resample_ref as R

        self.orig_freq, self.new_freq = round(sampling_rate * factor), sampling_rate
        k, self.width, self.orig, self.new = R.sinc_kernel(self.orig_freq, self.new_freq)
        self.kernel = torch.from_numpy(np.ascontiguousarray(k))[:, None, :]  # (new, 1, kw)
        self._len = R.resampled_length

    @torch.no_grad()
    def __call__(self, samples: np.ndarray) -> np.ndarray:
        if self.orig_freq == self.new_freq:
            return samples
        w = torch.from_numpy(np.asarray(samples, dtype=np.float32)).reshape(1, -1)
        length = w.shape[1]
        w = torch.nn.functional.pad(w, (self.width, self.width + self.orig))
        y = torch.nn.functional.conv1d(w[:, None], self.kernel, stride=self.orig)
        y = y.transpose(1, 2).reshape(1, -1)
        return y[0, : self._len(length, self.orig, self.new)].numpy()


class TorchKaldi:
    """EVERY deterministic configuration of the reference's four Kaldi-style layers in the reference's own float32 torch calls
    (round 6, VERDICT r5 Missing #4: `ref32` outside the headline used to be kaldi_ref's float64-FFT proxy):

      Wav2Win._forward_strided      layers.py:151-187  (DC removal, raw / windowed log-energy, replicate-pad pre-emphasis, window, zero pad)
      _get_strided_batch            layers.py:727-772  (snip_edges True / False, flip + cat reflection, as_strided view)
      _get_log_energy               layers.py:859-870
      Wav2Spec / Wav2LogSpec        layers.py:392-402, :461-473   (energy overwrites bin 0)
      Wav2LogFilterBank             layers.py:565-578  (energy column prepended)
      Wav2MFCC                      layers.py:708-724  (use_energy=True raises upstream, SURVEY Q4: not restated)
      _extract_batch                kaldi/extractors.py:485-554   (`extract_batch(..., "batch_zero_pad")`: pad_sequence with zeros, ONE
                                    batched forward, rows cut to compute_num_frames_from_samples)

    Constants: windows from torch's own window kernels (layers.py:921-940), DCT / lifter with torch's float32 cos / sin (layers.py:681-706),
    mel matrices from kaldi_ref.mel_matrix(float32) -- the reference evaluates lin2mel with NUMPY's float32 log on a tensor
    (layers.py:943-944 `np.log(1 + x / 700)`), which is what kaldi_ref restates.  tests/test_oracle.py::test_torch_kaldi_is_the_live_reference_bit_for_bit
    (authoring container) asserts array_equal with the live reference layers over the random-configuration family of the GPU suite, all
    windows, energy options, snip_edges, both mel scales, the four kinds and the zero-padded batch form."""

    def __init__(self, cfg: K.RefConfig):
        import math

        if cfg.dither != 0.0:
            raise ValueError("ref32 is deterministic: dither must be 0 (layers.py:191-193 draws randn)")
        if cfg.kind == "mfcc" and cfg.use_energy:
            raise NotImplementedError("Wav2MFCC(use_energy=True) raises in the reference (layers.py:721-722, SURVEY Q4): there is no ref32 for it")
        self.cfg = cfg
        self.n, self.shift, self.fft = K.window_sizes(cfg)
        n = self.n
        wt = cfg.window_type
        if wt == "hanning":
            self.window = torch.hann_window(n, periodic=False)
        elif wt == "hamming":
            self.window = torch.hamming_window(n, periodic=False, alpha=0.54, beta=0.46)
        elif wt == "povey":
            self.window = torch.hann_window(n, periodic=False).pow(0.85)
        elif wt == "rectangular":
            self.window = torch.ones(n, dtype=torch.float32)
        elif wt == "blackman":
            a = 2 * math.pi / n
            i = torch.arange(n, dtype=torch.float32)
            self.window = 0.42 - 0.5 * torch.cos(a * i) + (0.5 - 0.42) * torch.cos(2 * a * i)
        else:
            raise ValueError(f"Invalid window type: {wt}")
        self.eps = torch.tensor(torch.finfo(torch.float32).eps)
        if cfg.kind in ("fbank", "mfcc"):
            self.fb = torch.from_numpy(np.ascontiguousarray(K.mel_matrix(cfg, np.float32).astype(np.float32)))
        if cfg.kind == "mfcc":
            m, c = cfg.num_filters, cfg.num_ceps
            nn = torch.arange(float(m)).unsqueeze(1)
            kk = torch.arange(float(c))
            dct = torch.cos(math.pi / float(m) * (nn + 0.5) * kk)
            dct[:, 0] *= 1.0 / math.sqrt(2.0)
            dct *= math.sqrt(2.0 / float(m))
            self.dct = dct
            q = cfg.cepstral_lifter
            self.lifter = 1 + 0.5 * q * torch.sin(math.pi * torch.arange(c, dtype=torch.float32) / q) if q else None

    @property
    def feature_dim(self) -> int:
        c = self.cfg
        if c.kind == "fbank":
            return c.num_filters + (1 if c.use_energy else 0)
        return c.num_ceps if c.kind == "mfcc" else self.fft // 2 + 1

    def _strided(self, x: torch.Tensor) -> torch.Tensor:
        S = x.shape[-1]
        if self.cfg.snip_edges:
            if S < self.n:
                return torch.empty((0, 0, 0))
            T = 1 + (S - self.n) // self.shift
        else:
            T = (S + self.shift // 2) // self.shift
            npad = (T - 1) * self.shift + self.n - S
            npad_left = int((self.n - self.shift) // 2)
            npad_right = npad - npad_left
            pad_left = torch.flip(x[:, :npad_left], (1,))
            if npad_right >= 0:
                pad_right = torch.flip(x[:, -npad_right:], (1,))  # NB npad_right == 0 -> x[:, -0:] is the WHOLE row, as upstream (layers.py:761)
            else:
                pad_right = torch.zeros(0, dtype=x.dtype)
            x = torch.cat((pad_left, x, pad_right), dim=1)
        return x.as_strided([x.shape[0], T, self.n], (x.stride(0), self.shift * x.stride(1), x.stride(1)))

    def _log_energy(self, x: torch.Tensor) -> torch.Tensor:
        import math

        e = (x.pow(2).sum(-1) + 1e-15).log()
        if self.cfg.energy_floor > 0.0:
            e = torch.max(e, torch.tensor(math.log(self.cfg.energy_floor), dtype=e.dtype))
        return e

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S) float32 -> (B, T, F): layer.forward(x) of the reference."""
        c = self.cfg
        x = self._strided(x)
        if x.numel() == 0:
            return torch.zeros((x.shape[0] if x.dim() else 0, 0, self.feature_dim))
        # Wav2Win._forward_strided
        if c.remove_dc_offset:
            x = x - torch.mean(x, dim=2, keepdim=True)
        log_e = None
        if c.use_energy and c.raw_energy:
            log_e = self._log_energy(x)
        if c.preemph_coeff != 0.0:
            off = torch.nn.functional.pad(x, (1, 0), mode="replicate")
            x = x - c.preemph_coeff * off[:, :, :-1]
        x = x * self.window
        if self.fft != self.n:
            x = torch.nn.functional.pad(x.unsqueeze(1), [0, self.fft - self.n], mode="constant", value=0.0).squeeze(1)
        if c.use_energy and not c.raw_energy:
            log_e = self._log_energy(x)
        # the subclass' _forward_strided
        X = torch.fft.rfft(x, dim=-1)
        spec = X.abs() if c.use_fft_mag else X.abs() ** 2
        if c.kind == "spectrogram":
            if log_e is not None:
                spec[:, :, 0] = log_e
            return spec
        if c.kind == "log-spectrogram":
            spec = (spec + 1e-15).log()
            if log_e is not None:
                spec[:, :, 0] = log_e
            return spec
        mel = torch.max(torch.matmul(spec, self.fb), self.eps).log()
        if c.kind == "fbank":
            if log_e is not None:
                mel = torch.cat((log_e.unsqueeze(-1), mel), dim=-1)
            return mel
        mfcc = torch.matmul(mel, self.dct)
        if c.cepstral_lifter > 0:
            mfcc *= self.lifter
        return mfcc

    def extract(self, samples: np.ndarray) -> np.ndarray:
        """<Extractor>.extract (kaldi/extractors.py:92-115): (S,) -> (T, F)."""
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(samples, dtype=np.float32).reshape(1, -1)))
        return self.forward(x)[0].numpy()

    def extract_batch(self, waves, edge_rule: str = "reflect"):
        if edge_rule == "reflect":
            return [self.extract(w) for w in waves]
        assert edge_rule == "batch_zero_pad"
        items = [torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype=np.float32).reshape(-1))) for w in waves]
        lens = [K.compute_num_frames_from_samples(len(w), self.cfg.frame_shift, self.cfg.sampling_rate) for w in items]
        feats = self.forward(torch.nn.utils.rnn.pad_sequence(items, batch_first=True))
        return [feats[i, : lens[i]].numpy() for i in range(len(items))]


def reference_f32(cfg: K.RefConfig = None):
    """`ref32` of the parity statements: an object with `.extract(samples) -> (T, F) float32` (and `.extract_batch(waves, edge_rule)`)
    that runs the reference's own float32 torch call sequence for `cfg` -- every kind, window, energy option, edge rule (round 6:
    `TorchKaldi`; the BASELINE configurations keep their round-5 classes, which are the CPU baseline of bench.py)."""
    cfg = cfg or K.RefConfig(kind="fbank")
    plain = not cfg.snip_edges and not cfg.use_energy and cfg.window_type == "povey" and not cfg.use_fft_mag
    if cfg.kind == "fbank" and plain:
        return TorchFbank(cfg)
    if cfg.kind == "mfcc" and plain and cfg.remove_dc_offset and cfg.preemph_coeff != 0.0:
        return TorchMfcc(cfg=cfg)
    return TorchKaldi(cfg)


REF32_NAME = ("oracle/kaldi_torch.py (the reference's own float32 torch call sequence: as_strided framing, torch.fft.rfft, abs()**2, matmul, log; "
              "array_equal to the live reference on full-size cuts, tests/test_oracle.py::test_torch_ref32_is_the_live_reference_bit_for_bit)")
