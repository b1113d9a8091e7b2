"""Share of tile pairs the block-sparse mean-shift plans keep as a function of the relative bound
(PLAN_REL_EPS) on the benchmark's held-out embedding, and what the iterates lose: max |difference| of
the ten-iteration result against dense launches.  python tools/plan_eps_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARSENET_MS_STATS"] = "1"
import torch
from parsenet_codebase_amd import workloads
from parsenet_codebase_amd import mean_shift as MSM
from parsenet_codebase_amd.fitting_batch import bandwidth_batch

dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.model.eval()
for start in (0, 4, 8, 12):
    step.select(start)
    with torch.no_grad():
        emb, _, _ = step.model(step.x, step.labels, True)
        e = torch.nn.functional.normalize(emb.permute(0, 2, 1), dim=2).contiguous()
        bw, _ = bandwidth_batch(e, 0.025)
    print("batch at %d: bandwidths %s" % (start, [round(float(x), 4) for x in bw]))
    MSM.SPARSE = False
    with torch.no_grad():
        dense = MSM.mean_shift_iterations(e, bw, 10)
    MSM.SPARSE = True
    for eps in (1e-9, 1e-8, 1e-7, 5e-8, 1e-6, 1e-5, 1e-4):
        MSM.PLAN_REL_EPS = eps
        x = e.clone().requires_grad_(True)
        out = MSM.mean_shift_iterations(x, bw, 10)
        st = MSM.LAST_PLAN_STATS
        pairs = sum(t[0] for t in st) / len(st)
        err = float((out.detach() - dense).abs().max())
        print("  rel_eps %.0e: pairs kept %.4f (it0 %.4f it9 %.4f)  max |iterate - dense| %.2e"
              % (eps, pairs, st[0][0], st[-1][0], err))
