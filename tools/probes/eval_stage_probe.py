"""Where does the stage-wise evaluation spend its time?  cProfile of the host + per-kernel-family event timers."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from parsenet_codebase_amd import _lib, synthetic
from parsenet_codebase_amd.encoders import DGCNNControlPoints
from parsenet_codebase_amd.fitting import Evaluation

from parsenet_codebase_amd import dp
dp.limit_host_threads()
dev = torch.device("cuda:0")
B, N = 4, 10000
pts, nrm, lab, prim = synthetic.make_batch(2000, B, N)
torch.manual_seed(0)
ev = Evaluation(closed_path=DGCNNControlPoints(20, num_points=10, mode=1).eval().to(dev),
                open_path=DGCNNControlPoints(20, num_points=10, mode=0).eval().to(dev))
g = torch.Generator().manual_seed(5)
code = torch.nn.functional.normalize(torch.randn(64, 128, generator=g), dim=1)
emb = torch.nn.functional.normalize(code[torch.from_numpy(lab)] + 0.02 * torch.randn(B, N, 128, generator=g), dim=2).to(dev)
logp = torch.log_softmax(8.0 * torch.nn.functional.one_hot(torch.from_numpy(prim), 10).float().permute(0, 2, 1), 1).to(dev)
P, Nr = torch.from_numpy(pts).to(dev), torch.from_numpy(nrm).to(dev)
kw = dict(quantile=0.025, iterations=10, lamb=0.1)
fn = lambda: ev.fitting_losses_eval(emb, P, Nr, lab, prim, logp, **kw)
for _ in range(2):
    fn()
torch.cuda.synchronize()
_lib.prof_enable(True)
_lib.prof_reset()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    fn()
torch.cuda.synchronize()
pr.disable()
res = _lib.prof_results()
_lib.prof_enable(False)
for k, (t, c) in sorted(res.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%-28s %8.3f ms per call of the stage  (%d launches)" % (k, t / 3, c / 3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
