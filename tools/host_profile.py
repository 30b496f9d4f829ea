import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lhotse_amd
ex = lhotse_amd.HipFbank()
B = 60
x = (torch.rand(B, 160000) - 0.5)
xs = [x[i] for i in range(B)]
ex.extract_batch(xs, 16000); torch.cuda.synchronize()
def run():
    for _ in range(20):
        r = ex.extract_batch(xs, 16000)
        r = [t.cpu() for t in r] if isinstance(r, list) else r.cpu()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
print("threads", torch.get_num_threads())
