#!/usr/bin/env python3
"""Per-phase shader clocks per wave-round of whisper3_kernel on 10 s cuts (experiment build -DHIPFEAT_PHASE_TIMERS).
usage (GPU box): HIPFEAT_LIB=lhotse_amd/_lib/var_<name>.so python tools/phase_timers_c.py [cuts]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd
from lhotse_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ex = lhotse_amd.HipWhisperFbank(); plan = ex.plan; L = plan.lib
dll = L.backend.dll
dll.hipfeat_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
wave = (torch.rand(C, 160000, device="cuda") - 0.5)
out = torch.empty(C * 1000, 80, device="cuda")
offs = np.arange(C, dtype=np.int64) * 160000; lens = np.full(C, 160000, dtype=np.int64)
h = np.zeros(1, dtype=np.uint64)
L.check("hipfeat_layout_create", plan.handle, C, _lib.addr(offs), _lib.addr(lens), None, None, 80, None, _lib.addr(h))
nblocks = int(L.raw("hipfeat_layout_total_frames", int(h[0])) // 1000 * 8)  # upper bound
buf = torch.zeros(nblocks * 8 * 8, dtype=torch.int64, device="cuda")
assert dll.hipfeat_debug_set_phase_buffer(buf.data_ptr()) == 0
for it in range(3):
    L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
v = buf.view(-1, 8).double().sum(0).cpu().numpy()
rounds = v[7]
names = ["sample + window reads", "DMA issue, window, 25-point DFTs", "twiddles, transpose", "fft16, power rows", "wait span (vmcnt)", "mel, log10, stores", "-"]
print(plan.kernel_name, "wave-rounds:", int(rounds))
tot = v[:6].sum()
for n, x in zip(names[:6], v[:6]):
    print(f"  {n:38s} {x / rounds:9.0f} clk per wave-round  ({100 * x / tot:5.1f} %)")
print(f"  total {tot / rounds:9.0f} clk per wave-round")
