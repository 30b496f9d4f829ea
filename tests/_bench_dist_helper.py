"""Run under `python -m torch.distributed.run` by tests/test_sharding.py: bench.py's group initialisation on a box where RCCL cannot come up
(no GPU here) -- the ranks must meet on gloo through the fallback's own TCPStore, next to the launcher's agent store."""
import importlib.util
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

mode = sys.argv[1]
if mode == "rccl-raises":  # pretend there is one GPU per rank: init_process_group("nccl") is attempted and fails
    torch.cuda.device_count = lambda: 64
    dev = torch.device("cuda:0")
else:  # fewer GPUs than ranks: gloo without trying RCCL
    dev = torch.device("cpu")
dist, used = bench.init_dist("nccl", dev)
t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
if dist.get_rank() == 0:
    print(f"RESULT world={dist.get_world_size()} sum={t.item()} used={used}", flush=True)
dist.destroy_process_group()
