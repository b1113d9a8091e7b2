"""Locality orders for the block-sparse mean-shift plans on the benchmark's embedding: share of tile pairs
kept with the committed order (128 cells along a greedy chain) and with two-level variants (fine cells
inside the coarse cells).  python tools/order_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARSENET_MS_STATS"] = "1"
import torch
from parsenet_codebase_amd import workloads, kernels as K
from parsenet_codebase_amd import mean_shift as MSM
from parsenet_codebase_amd.fitting_batch import bandwidth_batch

dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.model.eval()
base_order = MSM.locality_order


def two_level(P2, lloyd2=2):
    def order(x, lloyd=2):
        B, N, D = x.shape
        P = 128

        def assign(pts, cen):
            return torch.bmm(pts, cen.transpose(1, 2)).argmax(2)

        def centres(pts, lab, K_, old):
            hot = torch.nn.functional.one_hot(lab, K_).to(pts.dtype)
            acc = torch.bmm(hot.transpose(1, 2), pts)
            nrm = acc.norm(dim=2, keepdim=True)
            return torch.where(nrm > 1e-6, acc / nrm.clamp_min(1e-6), old)
        cen = x[:, torch.linspace(0, N - 1, P, device=x.device).long()]
        lab = assign(x, cen)
        for _ in range(lloyd):
            cen = centres(x, lab, P, cen)
            lab = assign(x, cen)
        rank = K.meanshift_chain_order(torch.bmm(cen, cen.transpose(1, 2)))
        cen2 = x[:, torch.linspace(0, N - 1, P2, device=x.device).long()]
        lab2 = assign(x, cen2)
        for _ in range(lloyd2):
            cen2 = centres(x, lab2, P2, cen2)
            lab2 = assign(x, cen2)
        coarse_of_fine = assign(cen2, cen)                       # (B,P2): the coarse cell of every fine centre
        key = torch.gather(rank, 1, torch.gather(coarse_of_fine, 1, lab2)).long() * P2 + lab2
        return torch.argsort(key, dim=1, stable=True)
    return order


for start in (0, 8):
    step.select(start)
    with torch.no_grad():
        emb, _, _ = step.model(step.x, step.labels, True)
        e = torch.nn.functional.normalize(emb.permute(0, 2, 1), dim=2).contiguous()
        bw, _ = bandwidth_batch(e, 0.025)
    MSM.SPARSE = True
    for name, fn in (("128 cells (committed)", base_order), ("128 x fine 256", two_level(256)), ("128 x fine 384", two_level(384)),
                     ("128 x fine 512", two_level(512)), ("128 x fine 1024", two_level(1024))):
        MSM.locality_order = fn
        with torch.no_grad():
            MSM.mean_shift_iterations(e, bw, 10)
        st = MSM.LAST_PLAN_STATS
        print("batch %d  %-24s pairs kept %.4f (it0 %.4f it9 %.4f)" % (start, name, sum(t[0] for t in st) / len(st), st[0][0], st[-1][0]))
