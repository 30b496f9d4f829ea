#!/usr/bin/env python3
"""The reference's OWN GPU mode on this MI355X, beside the HIP path: lhotse's Fbank(FbankConfig(device="cuda")) runs
Wav2LogFilterBank as ~12 full-tensor torch ops (hipFFT for the rFFT, rocBLAS for the mel matmul).  /root/reference cannot
travel to the GPU box, so the same op sequence is taken from oracle/kaldi_torch.py (TEST INFRASTRUCTURE, bit-identical to
the reference on the CPU goldens).  Device-resident equal-length batches of 10 s cuts, as in bench.py.  One JSON line."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
from oracle.kaldi_torch import TorchFbank

res = {}
tf = TorchFbank(device="cuda")
ex = LA.HipFbank()
for B in (60, 512):
    x = torch.empty(B, 160000, device="cuda").uniform_(-0.5, 0.5)
    ref = tf.forward_batch(x); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); ref = tf.forward_batch(x); b.record()
    torch.cuda.synchronize()
    ms_t = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    offs = np.arange(B, dtype=np.int64) * 160000; lens = np.full(B, 160000, dtype=np.int64)
    out, _ = ex.plan.run(x.view(-1), offs, lens, None); torch.cuda.synchronize()
    for a, b in evs:
        a.record(); out, _ = ex.plan.run(x.view(-1), offs, lens, None); b.record()
    torch.cuda.synchronize()
    ms_h = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    err = float((out.view(B, 1000, 80) - ref).abs().max())
    res[f"batch_{B}"] = {"torch_ops_ms": round(ms_t, 3), "torch_ops_cuts_per_s": round(B / ms_t * 1e3, 1), "hip_ms": round(ms_h, 3),
                         "hip_cuts_per_s": round(B / ms_h * 1e3, 1), "speedup": round(ms_t / ms_h, 1), "max_abs_diff": err,
                         "torch_peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
print(json.dumps({"workload": "B x 10 s @ 16 kHz -> 80-dim fbank, device resident: the reference's torch-op sequence on the GPU vs libhipfeat", **res}))

# ---- speed perturbation: the reference's ResampleTensor on the GPU is F.pad + conv1d(stride=orig) + reshape + trim
#      (lhotse/augmentation/resample.py:284-315); whisper: torch.stft + matmul + log10 (lhotse/features/whisper_fbank.py:62-80)
from lhotse_amd import augmentation as A, constants

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); r = fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs])), r

extra = {}
B = 512
x = torch.empty(B, 160000, device="cuda").uniform_(-0.5, 0.5)
for factor in (0.9, 1.1):
    k, width, orig, new = constants.sinc_resample_kernel(round(16000 * factor), 16000)
    kt = torch.from_numpy(k).cuda()[:, None, :]
    def torch_resample():
        w = torch.nn.functional.pad(x, (width, width + orig))
        y = torch.nn.functional.conv1d(w[:, None], kt, stride=orig)
        y = y.transpose(1, 2).reshape(B, -1)
        return y[..., : int(np.ceil(np.float32(new * 160000 / orig)))]
    r = A.get_or_create_resampler(round(16000 * factor), 16000)
    ms_t, yt = timed(torch_resample)
    ms_h, yh = timed(lambda: r(x))
    extra[f"speed_{factor}"] = {"torch_conv1d_ms": round(ms_t, 3), "hip_ms": round(ms_h, 3), "speedup": round(ms_t / ms_h, 1),
                                "max_abs_diff": float((yt - yh).abs().max())}
filters = torch.from_numpy(np.ascontiguousarray(constants.make_slaney_mel(80, 400, 16000).T)).cuda()
win = torch.hann_window(400, device="cuda")
def torch_whisper():
    st = torch.stft(x, 400, 160, window=win, return_complex=True)
    mag = st[..., :-1].abs() ** 2
    ls = torch.clamp(filters @ mag, min=1e-10).log10()
    ls = torch.maximum(ls, ls.amax(dim=(1, 2), keepdim=True) - 8.0)
    return ((ls + 4.0) / 4.0).transpose(1, 2)
wh = LA.HipWhisperFbank()
offs = np.arange(B, dtype=np.int64) * 160000; lens = np.full(B, 160000, dtype=np.int64)
ms_t, wt = timed(torch_whisper)
ms_h, (wo, _) = timed(lambda: wh.plan.run(x.view(-1), offs, lens, None))
extra["whisper"] = {"torch_stft_ms": round(ms_t, 3), "hip_ms": round(ms_h, 3), "speedup": round(ms_t / ms_h, 1),
                    "max_abs_diff": float((wt.contiguous().view(-1, 80) - wo).abs().max())}
print(json.dumps({"workload": f"{B} x 10 s: reference torch-op GPU paths vs libhipfeat (speed perturbation, whisper log-mel)", **extra}))
