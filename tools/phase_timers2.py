import ctypes, os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/lhotse_amd') else os.getcwd())
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
buf = torch.zeros(C*16 * 4 * 8, dtype=torch.int64, device="cuda")
assert dll.hipfeat_debug_set_phase_buffer(buf.data_ptr()) == 0
for it in range(2):
    L.check("hipfeat_extract_layout", plan.handle, int(h[0]), wave.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
v = buf.view(-1, 8).double().sum(0).cpu().numpy()
tiles = v[5]
names = ["reads+sum","mean+preproc","fft1","twiddle+exwrite","exread","(tiles)","fft2","mirror+split+Pwrite"]
print(plan.kernel_name)
for n,x in zip(names,v):
    print(f"  {n:22s} {x/tiles:8.0f} clk per wave-tile")
