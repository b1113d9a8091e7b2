"""How many deterministic seg-only steps (triplet + NLL on the bench batch, lr 1e-2, train mode)
until mean-shift (quantile 0.025, 10 iterations) finds a handful of clusters per shape; also the
number of occupied centres the non-maximum suppression sees.  python tools/pretrain_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from parsenet_codebase_amd import workloads
from parsenet_codebase_amd.fitting_batch import bandwidth_batch, nms_batch
from parsenet_codebase_amd.mean_shift import mean_shift_iterations

dev = torch.device("cuda:0")
step = workloads.ParsenetSegStep(dev, batch=4, num_points=10000)
np.random.seed(1000)
done = 0
for target in (0, 20, 50, 100, 150, 200, 300, 500, 800, 1200):
    step.model.train()
    while done < target:
        step.step()
        done += 1
    step.model.eval()
    with torch.no_grad():
        emb, _, l = step.model(step.x, step.labels, True)
        e = torch.nn.functional.normalize(emb.permute(0, 2, 1), dim=2)
        bw, _ = bandwidth_batch(e, 0.025)
        new_X = mean_shift_iterations(e, bw, 10)
        st = nms_batch(new_X, e, bw)
    gt = [len(np.unique(step.labels[b])) for b in range(4)]
    print("steps %4d  embed loss %.4f  clusters %s  occupied centres %s  gt %s  bw %s" % (
        done, float(l.mean()), st["ncl"].tolist(), st["nocc"].tolist(), gt, [round(float(x), 3) for x in bw]),
        flush=True)
