#!/usr/bin/env python3
"""hipfeat_pcm16_to_float streaming rate (int16 in, float32 out), device resident."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lhotse_amd import _lib
L = _lib.load()
n = 4000 * 160000
pcm = torch.randint(-32768, 32767, (n,), dtype=torch.int16, device="cuda")
out = torch.empty(n, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L.check("hipfeat_pcm16_to_float", pcm.data_ptr(), out.data_ptr(), n, st); torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in evs:
    a.record(); L.check("hipfeat_pcm16_to_float", pcm.data_ptr(), out.data_ptr(), n, st); b.record()
torch.cuda.synchronize()
ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
print(json.dumps({"samples": n, "ms": round(ms, 3), "GBps": round(n * 6 / ms / 1e6, 1), "frac_of_8TBps": round(n * 6 / ms / 1e6 / 8000, 3)}))
