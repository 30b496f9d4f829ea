"""TEST INFRASTRUCTURE -- one single-threaded host process of bench.py's CPU baseline.

    python oracle/cpu_worker.py <seconds> <seed>

Loops the CPU Fbank path (oracle/kaldi_torch.py: the reference's own sequence of torch calls, bit-identical to the
reference on the goldens; one 10 s cut per call with torch.set_num_threads(1), as CutSet.compute_and_store_features /
`lhotse feat extract` run the reference extractor) for <seconds> and prints "<cuts> <elapsed>".
`mfcc40` as third argument loops the 40 x 40 MFCC on LibriSpeech-like lengths (bench.py --config mfcc40_libri) and prints
"<cuts> <elapsed> <audio seconds>"; `onthefly` loops speed perturbation (0.9 / 1.0 / 1.1) + Fbank on U(1, 30) s cuts (--config onthefly).
`numpy` as third argument times the numpy restatement (oracle/kaldi_ref.py) instead; `batched` times the batched forward on
batches of 60 cuts with torch's DEFAULT intra-op threads, as Fbank.extract_batch runs it (lhotse/features/kaldi/extractors.py:485-554),
and prints "<cuts> <elapsed> <threads>".
"""
import os
import sys
import time

BATCHED = len(sys.argv) > 3 and sys.argv[3] == "batched"
if not BATCHED:
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from oracle.kaldi_ref import RefConfig, RefExtractor  # noqa: E402
from oracle.signals import make_signal  # noqa: E402


def main():
    seconds, seed = float(sys.argv[1]), int(sys.argv[2])
    if BATCHED:
        import torch

        from oracle.kaldi_torch import TorchFbank

        ex = TorchFbank()
        x = torch.from_numpy(np.stack([make_signal("uniform", 160000, seed + s) for s in range(60)]))
        ex.forward_batch(x)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            ex.forward_batch(x)
            n += 60
        print(n, time.perf_counter() - t0, torch.get_num_threads(), flush=True)
        return
    if len(sys.argv) > 3 and sys.argv[3] in ("mfcc40", "onthefly"):
        import torch

        from oracle.kaldi_torch import TorchFbank, TorchMfcc, TorchSpeed

        torch.set_num_threads(1)
        rs = np.random.RandomState(seed)
        if sys.argv[3] == "mfcc40":
            ex = TorchMfcc(40, 40, 22)
            lens = np.round(np.clip(np.exp(rs.randn(16) * 0.45 + 2.42), 1.0, 35.0) * 16000).astype(int)
            pool = [(make_signal("uniform", int(n), seed + i), None) for i, n in enumerate(lens)]
        else:
            ex = TorchFbank()
            speeds = {f: TorchSpeed(16000, f) for f in (0.9, 1.0, 1.1)}
            lens = (rs.uniform(1.0, 30.0, size=16) * 16000).astype(int)
            pool = [(make_signal("uniform", int(n), seed + i), speeds[(0.9, 1.0, 1.1)[i % 3]]) for i, n in enumerate(lens)]
        ex.extract(pool[0][0])
        n, secs, t0 = 0, 0.0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            x, sp = pool[n % len(pool)]
            ex.extract(sp(x) if sp is not None else x)
            secs += len(x) / 16000.0
            n += 1
        print(n, time.perf_counter() - t0, secs, flush=True)
        return
    if len(sys.argv) > 3 and sys.argv[3] == "numpy":
        ex = RefExtractor(RefConfig(kind="fbank"), np.float32)
    else:
        import torch

        from oracle.kaldi_torch import TorchFbank

        torch.set_num_threads(1)  # lhotse/bin/modes/features.py:25-32
        ex = TorchFbank()
    pool = [make_signal("uniform", 160000, seed + s) for s in range(4)]
    ex.extract(pool[0])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ex.extract(pool[n % len(pool)])
        n += 1
    print(n, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
