import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd
from lhotse_amd import _lib
C=4000
ex = lhotse_amd.HipFbank(); plan = ex.plan; L = plan.lib
dll = L.backend.dll
dll.hipfeat_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
wave = (torch.rand(C, 160000, device="cuda") - 0.5)
out = torch.empty(C * 1000, 80, device="cuda")
offs = np.arange(C, dtype=np.int64) * 160000; lens = np.full(C, 160000, dtype=np.int64)
h = np.zeros(1, dtype=np.uint64)
L.check("hipfeat_layout_create", plan.handle, C, _lib.addr(offs), _lib.addr(lens), None, None, 80, None, _lib.addr(h))
nb = int(np.ceil(1000/256))*C
buf = torch.zeros(nb * 4 * 8, dtype=torch.int64, device="cuda")
assert dll.hipfeat_debug_set_phase_buffer(buf.data_ptr()) == 0
for it in range(2):
    L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
b = buf.view(-1, 4, 8).double()
names = ["vmcnt wait","barrier1","S3","barrier2","S5","(tiles)","S5: weights wait","S5: seg0 DMA+Pread+MFMA"]
print(plan.kernel_name)
for w in range(4):
    v = b[:, w, :].sum(0).cpu().numpy(); tiles = v[5]
    print(f" wave {w}: " + "  ".join(f"{n}={x/tiles:.0f}" for n, x in zip(names, v) if n != "(tiles)"))
