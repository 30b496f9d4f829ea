#!/usr/bin/env python3
"""librosa-style log-mel throughput, device resident: C x 10 s cuts @ 22.05 kHz -> (861, 80).  One JSON line; with
--torch also times the same arithmetic as torch ops on the GPU (torch.stft + abs + matmul + log10)."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
from lhotse_amd import constants

ap = argparse.ArgumentParser()
ap.add_argument("--cuts", type=int, default=2000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--torch", action="store_true")
a = ap.parse_args()
ex = LA.HipLibrosaFbank()
plan = ex.plan
S = 220500
wave = torch.empty(a.cuts * S, device="cuda").uniform_(-0.5, 0.5)
offs = np.arange(a.cuts, dtype=np.int64) * S
lens = np.full(a.cuts, S, dtype=np.int64)


def timed(fn):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return float(np.mean([s.elapsed_time(e) for s, e in evs]))


ms = timed(lambda: plan.run(wave, offs, lens, None))
rows = (S + 128) // 256
line = {"workload": f"{a.cuts} x 10 s cuts @ 22.05 kHz -> librosa log-mel ({rows}, 80), device resident", "kernel": plan.kernel_name,
        "ms_per_launch": round(ms, 3), "cuts_per_s": round(a.cuts / ms * 1e3, 1), "audio_seconds_per_s": round(a.cuts * 10 / ms * 1e3, 1),
        "algorithmic_GBps": round(a.cuts * (S * 4 + rows * 80 * 4) / ms / 1e6, 1)}
if a.torch:
    win = torch.hann_window(1024, device="cuda")
    mel = torch.from_numpy(constants.make_slaney_mel(80, 1024, 22050, 80, 7600)).cuda()
    x2 = wave.view(a.cuts, S)
    chunk = 250

    def torch_path():
        for i in range(0, a.cuts, chunk):
            st = torch.stft(x2[i : i + chunk], 1024, 256, window=win, center=True, pad_mode="reflect", return_complex=True)
            torch.log10(torch.clamp(st.abs().transpose(1, 2) @ mel, min=1e-10))[:, :rows]

    tms = timed(torch_path)
    line["torch_gpu_ms"] = round(tms, 3)
    line["torch_gpu_cuts_per_s"] = round(a.cuts / tms * 1e3, 1)
print(json.dumps(line))
