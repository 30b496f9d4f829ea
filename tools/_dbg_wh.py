import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import lhotse_amd as LA
from oracle import whisper_ref as W
ex = LA.HipWhisperFbank()
F = W.slaney_mel_filters()
rng = np.random.RandomState(0)
xs = [(rng.rand(n).astype(np.float32) - 0.5) * s for n, s in [(16000, 1.0), (40123, 0.1), (160000, 0.9), (201, 1.0), (8079, 0.5), (8080, 0.5)]]
outs = ex.extract_batch([torch.from_numpy(x) for x in xs], 16000)
for x, o in zip(xs, outs):
    y = o.cpu().numpy()
    t = W.log_mel_spectrogram(x, F, dtype=np.float64)
    e = np.abs(y - t).max(axis=1)
    bad = np.nonzero(e > 1e-3)[0]
    print(len(x), y.shape, "max", e.max(), "bad frames", bad[:8], bad[-3:], len(bad), "ymax", y.max(), t.max(), "ymin", y.min(), t.min())
y = ex.extract(xs[2], 16000); t = W.log_mel_spectrogram(xs[2], F, dtype=np.float64); print("single", np.abs(y - t).max())
