import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from _hip import make_hip
from oracle.kaldi_ref import RefConfig, RefExtractor
w = (np.random.RandomState(31).rand(16000).astype(np.float32) - 0.5)
t64 = RefExtractor(RefConfig(kind="spectrogram"), np.float64).extract(w)
o32 = RefExtractor(RefConfig(kind="spectrogram"), np.float32).extract(w)
def rep(name, y):
    d = np.abs(y - t64); rel = d / t64.max(axis=1, keepdims=True)
    i = np.unravel_index(np.argmax(d / t64), d.shape)
    print(f"{name:10s} max abs err / frame max power = {rel.max():.3e}   mean = {rel.mean():.3e}   worst bin rel err = {(d / t64).max():.3e} at (frame, bin) {i}; rel err by bin class: DC {(d / t64)[:, 0].max():.2e} Nyquist {(d / t64)[:, 256].max():.2e} bin128 {(d / t64)[:, 128].max():.2e} others {np.delete(d / t64, [0, 128, 256], axis=1).max():.2e}")
rep("oracle32", o32)
for env, name in [({}, "fast"), ({"HIPFEAT_FORCE_GENERIC": "1"}, "generic")]:
    for k in ("HIPFEAT_FORCE_GENERIC", "HIPFEAT_FFT512_VARIANT"): os.environ.pop(k, None)
    os.environ.update(env)
    ex = make_hip("spectrogram", {})
    rep(name + ":" + ex.kernel_name.split()[0], ex.extract(w, 16000))
