"""Where does the HOST spend a cfg2 / cfg3 step (the steps are bound by its launch rate)?
python tools/probes/spline_host_profile.py [closed]"""
import cProfile
import os
import pstats
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parsenet_codebase_amd import dp, workloads  # noqa: E402

dp.limit_host_threads()
dev = torch.device("cuda:0")
s = workloads.SplineNetStep(dev, closed=len(sys.argv) > 1 and sys.argv[1] == "closed")
for _ in range(10):
    s.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    s.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(22)
