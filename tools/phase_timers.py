#!/usr/bin/env python3
"""Run the bench workload on an experiment build with -DHIPFEAT_PHASE_TIMERS and print per-phase clocks per tile per wave.
usage (GPU box): HIPFEAT_LIB=lhotse_amd/_lib/var_<name>.so python tools/phase_timers.py [cuts]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd
from lhotse_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ex = lhotse_amd.HipFbank(); plan = ex.plan; L = plan.lib
dll = L.backend.dll
dll.hipfeat_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
wave = (torch.rand(C, 160000, device="cuda") - 0.5)
out = torch.empty(C * 1000, 80, device="cuda")
offs = np.arange(C, dtype=np.int64) * 160000; lens = np.full(C, 160000, dtype=np.int64)
h = np.zeros(1, dtype=np.uint64)
L.check("hipfeat_layout_create", plan.handle, C, _lib.addr(offs), _lib.addr(lens), None, None, 80, None, _lib.addr(h))
nblocks = C * 16
buf = torch.zeros(nblocks * 4 * 8, dtype=torch.int64, device="cuda")
assert dll.hipfeat_debug_set_phase_buffer(buf.data_ptr()) == 0
for it in range(3):
    L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
v = buf.view(-1, 8).double().sum(0).cpu().numpy()
tiles = v[5]
names = ["S1 work", "barrier1", "S3", "barrier2", "S5"]
print(plan.kernel_name, "wave-tiles:", int(tiles))
tot = v[:5].sum()
for n, x in zip(names, v[:5]):
    print(f"  {n:10s} {x / tiles:9.0f} clk per wave-tile  ({100 * x / tot:5.1f} %)")
print(f"  total      {tot / tiles:9.0f} clk per wave-tile")
