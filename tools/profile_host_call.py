import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import torch, numpy as np
import lhotse_amd as LA
ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
B = 60
x = (torch.rand(B, 160000) - 0.5).pin_memory()
lens = torch.full((B,), 160000, dtype=torch.int32)
for _ in range(20): ex.extract_batch(x, 16000, lengths=lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): ex.extract_batch(x, 16000, lengths=lens)
dt = (time.perf_counter() - t0) / 200
print(f"{dt*1e3:.3f} ms per call = {B/dt:.0f} cuts/s")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): ex.extract_batch(x, 16000, lengths=lens)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3800])
