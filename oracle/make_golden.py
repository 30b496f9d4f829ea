#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE -- generates tests/golden/*.npz by running the REFERENCE itself.

Run in the authoring container only (needs /root/reference):

    python oracle/make_golden.py

The reference package needs `soundfile`, `intervaltree` and `cytoolz` at import time
(lhotse/audio/source.py:11, lhotse/cut/base.py:7); none of them is used by the
feature-extraction path, so permissive stub modules are injected when they are missing.
Outputs are the reference's float32 results (torch CPU) -- the committed fixtures are
what pins oracle/kaldi_ref.py and the HIP path on machines without the reference.
"""
from __future__ import annotations

import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.golden_cases import CASES  # noqa: E402
from oracle.signals import crc, make_signal  # noqa: E402


def import_reference(path: str = "/root/reference"):
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    for name in ("soundfile", "intervaltree", "cytoolz"):
        try:
            __import__(name)
        except ImportError:
            m = types.ModuleType(name)
            m.__file__ = f"<stub {name}>"

            def _ga(attr, _A=_Any):
                if attr.startswith("__"):
                    raise AttributeError(attr)
                return _A

            m.__getattr__ = _ga
            sys.modules[name] = m
    if path not in sys.path:
        sys.path.insert(0, path)
    import lhotse  # noqa: F401
    from lhotse.features.kaldi import extractors

    return extractors


def build(ex_mod, kind: str, cfg: dict):
    table = {
        "fbank": (ex_mod.Fbank, ex_mod.FbankConfig),
        "mfcc": (ex_mod.Mfcc, ex_mod.MfccConfig),
        "spectrogram": (ex_mod.Spectrogram, ex_mod.SpectrogramConfig),
        "log-spectrogram": (ex_mod.LogSpectrogram, ex_mod.LogSpectrogramConfig),
    }
    cls, ccls = table[kind]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return cls(ccls(**cfg))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ex_mod = import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    index = {}
    total = 0
    only = set(sys.argv[1:])  # optional: regenerate just the named cases (the index keeps the others' entries)
    if only and os.path.exists(os.path.join(out_dir, "index.json")):
        with open(os.path.join(out_dir, "index.json")) as f:
            index = json.load(f)["cases"]
    for case in CASES:
        if only and case["name"] not in only:
            continue
        ex = build(ex_mod, case["kind"], case["cfg"])
        sr = ex.config.sampling_rate
        waves = [make_signal(k, n, seed, sr) for k, n, seed in case["inputs"]]
        arrays = {}
        if case["mode"] == "extract":
            outs = [ex.extract(w, sr) for w in waves]
        elif case["mode"] == "batch":
            res = ex.extract_batch([w for w in waves], sr)
            outs = list(res) if not isinstance(res, np.ndarray) or res.ndim == 3 else [res]
        elif case["mode"] == "batch_lengths":
            lens = torch.tensor([len(w) for w in waves], dtype=torch.int32)
            padded = torch.zeros(len(waves), int(lens.max()))
            for i, w in enumerate(waves):
                padded[i, : len(w)] = torch.from_numpy(w)
            res = ex.extract_batch(padded, sr, lengths=lens)
            outs = list(res)
        else:
            raise ValueError(case["mode"])
        outs = [np.asarray(o, dtype=np.float32) for o in outs]
        assert len(outs) == len(waves), (case["name"], len(outs), len(waves))
        rows = case["rows"]
        for i, o in enumerate(outs):
            arrays[f"shape{i}"] = np.array(o.shape, dtype=np.int64)
            arrays[f"sum{i}"] = np.array(o.astype(np.float64).sum())
            sample = case.get("sample", 0)
            if sample and o.shape[0] > sample + 8:
                pick = np.random.RandomState(1000 + i).choice(o.shape[0] - 8, size=sample, replace=False) + 4
                sel = np.unique(np.concatenate([np.arange(4), pick, np.arange(o.shape[0] - 4, o.shape[0])]))
                arrays[f"rows{i}"] = sel.astype(np.int64)
                arrays[f"sel{i}"] = o[sel]
            elif rows and o.shape[0] > 2 * rows:
                arrays[f"head{i}"] = o[:rows]
                arrays[f"tail{i}"] = o[-rows:]
            else:
                arrays[f"out{i}"] = o
            arrays[f"crc{i}"] = np.array(crc(waves[i]), dtype=np.uint64)
        # constants of the reference module (bit-exactness target for lhotse_amd/constants.py)
        mod = ex.extractor
        arrays["window"] = mod.wav2win._window.detach().numpy()
        arrays["fft_length"] = np.array(mod.fft_length)
        if hasattr(mod, "_fb"):
            arrays["fb"] = mod._fb.detach().numpy()
        if hasattr(mod, "_dct"):
            arrays["dct"] = mod._dct.detach().numpy()
            lf = mod._lifter
            arrays["lifter"] = lf.detach().numpy() if isinstance(lf, torch.Tensor) else np.array(lf, dtype=np.float32)
        path = os.path.join(out_dir, case["name"] + ".npz")
        np.savez_compressed(path, **arrays)
        sz = os.path.getsize(path)
        total += sz
        index[case["name"]] = {"bytes": sz, "num_items": len(outs)}
        print(f"{case['name']:24s} items={len(outs):2d} {sz/1024:8.1f} KiB")
    meta = {
        "generator": "oracle/make_golden.py",
        "reference": "lhotse (reference tree mounted at /root/reference), lhotse/features/kaldi/extractors.py",
        "torch": torch.__version__,
        "numpy": np.__version__,
        "cases": index,
    }
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(f"total {total/1024:.1f} KiB in {len(CASES)} cases")


if __name__ == "__main__":
    main()
