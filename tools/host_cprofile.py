"""cProfile of the host thread over a few cfg5 steps (tottime: where Python itself spends time;
the blocking downloads show up as `.cpu()` / `numpy` of CUDA tensors)."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parsenet_codebase_amd import dp, workloads

dp.limit_host_threads()       # as bench.py and the trainer run: one intra-op CPU thread

dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.warm_paths()
for _ in range(4):
    step.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
