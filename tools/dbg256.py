import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd as LA
rs = np.random.RandomState(0)
x = (rs.rand(8000).astype(np.float32) - 0.5)
ex = LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, device="cuda:0"))
y = ex.extract(x, 8000)
os.environ["HIPFEAT_FORCE_GENERIC"] = "1"
g = LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, device="cuda:0"))
z = g.extract(x, 8000)
d = np.abs(y - z); bad = d > 1e-3
print(os.environ.get("HIPFEAT_LIB", "")[-14:], ex.kernel_name.split()[0], "bad rows:", np.nonzero(bad.any(axis=1))[0][:40], "ncols", int(bad.any(axis=0).sum()))
