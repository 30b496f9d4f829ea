#!/usr/bin/env python3
"""The on-the-fly training front end end to end on one GPU, device resident after the H2D copy: a 600 s mini-batch of cuts of
U(1, 30) s held as host float32 arrays -> pack + H2D -> speed perturbation (0.9 / 1.0 / 1.1) -> 80-dim fbank collated to
(B, Tmax, 80) -> GlobalMVN -> SpecAugment (defaults).  HIP path against the same chain built from the reference's torch-op
implementations restated in oracle/ (kaldi_torch.TorchFbank batch forward, resample_ref via torch conv1d, specaug_torch) on the same
GPU.  One JSON line.

    python tools/bench_pipeline.py [--batches 20]
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lhotse_amd as LA
from lhotse_amd import augmentation as A, constants
from lhotse_amd.extractors import pack_to_device
from oracle.kaldi_torch import TorchFbank
from oracle.specaug_torch import TorchSpecAugment

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=20)
a = ap.parse_args()
rng = np.random.RandomState(0)
batches = []
for b in range(a.batches):
    lens, tot = [], 0.0
    while True:
        d = rng.uniform(1.0, 30.0)
        if tot + d > 600.0:
            break
        lens.append(int(d * 16000)); tot += d
    batches.append([(rng.rand(n).astype(np.float32) - 0.5, float(rng.choice([0.9, 1.0, 1.1]))) for n in lens])
dev = torch.device("cuda", 0)
means, stds = torch.randn(80) - 8, torch.rand(80) * 2 + 1

# ---- HIP path ------------------------------------------------------------------------------------------------------
ex = LA.HipFbank()
res = {f: A.get_or_create_resampler(round(16000 * f), 16000) for f in (0.9, 1.1)}
mvn = LA.HipGlobalMVN(80); mvn.load_state_dict({"norm_means": means, "norm_stds": stds})
aug = LA.HipSpecAugment()


def hip_batch(batch):
    packed, offs, lens_ = pack_to_device([x for x, _ in batch], dev)
    cuts = [packed[o : o + n] for o, n in zip(offs.tolist(), lens_.tolist())]
    out = list(cuts)
    for f, r in res.items():
        idx = [i for i, (_, ff) in enumerate(batch) if ff == f]
        if idx:
            for i, y in zip(idx, r.resample_batch([cuts[i] for i in idx])):
                out[i] = y
    feats, lens = ex.extract_collated(out, 16000)
    return aug(mvn(feats)), lens


# ---- the reference's op sequences on the same GPU -------------------------------------------------------------------
tf = TorchFbank(device="cuda")
taug = TorchSpecAugment()
kern = {}
for f in (0.9, 1.1):
    k, width, orig, new = constants.sinc_resample_kernel(round(16000 * f), 16000)
    kern[f] = (torch.from_numpy(k).to(dev)[:, None, :], width, orig, new)
tm, ts = means.to(dev), stds.to(dev)


def torch_resample(x, f):  # ResampleTensor.forward (lhotse/augmentation/resample.py:284-315)
    k, width, orig, new = kern[f]
    n = x.shape[-1]
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(x[None, None], (width, width + orig)), k, stride=orig)
    y = y.transpose(1, 2).reshape(1, -1)
    return y[0, : int(np.ceil(np.float32(new * n / orig)))]


def torch_batch(batch):
    cuts = [torch.from_numpy(x).to(dev, non_blocking=True) for x, _ in batch]
    cuts = [c if f == 1.0 else torch_resample(c, f) for c, (_, f) in zip(cuts, batch)]
    # _extract_batch (lhotse/features/kaldi/extractors.py:485-554): zero-pad to the longest, one batched forward, slice
    padded = torch.nn.utils.rnn.pad_sequence(cuts, batch_first=True)
    full = tf.forward_batch(padded)
    lens = torch.tensor([(len(c) + 80) // 160 for c in cuts])
    feats = [full[i, :n] for i, n in enumerate(lens.tolist())]
    col = torch.nn.utils.rnn.pad_sequence(feats, batch_first=True, padding_value=float(LA.compat.LOG_EPSILON))  # collate_matrices
    return taug((col - tm) / ts), lens


def timed(fn):
    fn(batches[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        fn(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(batches) * 1e3


hip_ms = timed(hip_batch)
try:
    torch_first_ms = timed(torch_batch)  # every new waveform length makes MIOpen search for a conv1d kernel
    torch_ms = timed(torch_batch)        # the same batches again: shapes already known
except Exception as e:  # noqa: BLE001
    torch_ms = None
    print("torch chain failed:", repr(e), file=sys.stderr)
ncuts = sum(len(b) for b in batches) / len(batches)
line = {"workload": "600 s mini-batch (cuts U(1,30) s, host float32) -> H2D -> speed 0.9/1.0/1.1 -> fbank-80 collated -> GlobalMVN -> SpecAugment", "cuts_per_batch": round(ncuts, 1),
        "hip_ms_per_batch": round(hip_ms, 3), "hip_batches_per_s": round(1e3 / hip_ms, 1), "hip_audio_seconds_per_s": round(600e3 / hip_ms, 0)}
if torch_ms:
    line.update({"torch_gpu_ms_per_batch_first_pass": round(torch_first_ms, 3), "torch_gpu_ms_per_batch_shapes_seen": round(torch_ms, 3),
                 "speedup_vs_shapes_seen": round(torch_ms / hip_ms, 1)})
print(json.dumps(line))
