"""What a MASS-based drop criterion would buy the block-sparse plans (round 4 experiment).
Today a cap pair (a of the iterate, x of the data) is dropped when N exp((U_ax - 1)/b^2) <= eps exp((L_a - 1)/b^2):
all N points are assumed to sit at the pair's upper bound.  Tighter and still rigorous: drop the x caps with the
smallest U_ax as long as the SUM of 32 exp((U_ax - L_a)/b^2) over the dropped caps stays <= eps.
Emulated on the host from the kernels' own cap data (iteration 0 of the benchmark's embedding)."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parsenet_codebase_amd import workloads, kernels as K
from parsenet_codebase_amd import mean_shift as MSM
from parsenet_codebase_amd.fitting_batch import bandwidth_batch

dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.model.eval()
N = 10000
for start in (0, 4, 8, 12):
    step.select(start)
    with torch.no_grad():
        emb, _, _ = step.model(step.x, step.labels, True)
        e = torch.nn.functional.normalize(emb.permute(0, 2, 1), dim=2).contiguous()
        bw, _ = bandwidth_batch(e, 0.025)
        perm = MSM.locality_order(e, 2)
        x = torch.gather(e, 1, perm.unsqueeze(2).expand(-1, -1, 128))
        cen, rho, _cnt = K.meanshift_x3_tileinfo(x)            # (B,T,2,128), (B,T,2)
        B, T = rho.shape[:2]
        c = cen.reshape(B, 2 * T, 128).double()
        r = rho.reshape(B, 2 * T).double()
        valid = r >= 0
        th = torch.acos((c @ c.transpose(1, 2)).clamp(-1, 1))
        rs = r.clamp_min(0)
        hi = th + rs.unsqueeze(2) + rs.unsqueeze(1)
        pm = torch.where(hi >= math.pi, torch.full_like(hi, -1.0), torch.cos(hi))
        pm = torch.where(valid.unsqueeze(1) & valid.unsqueeze(2), pm, torch.full_like(pm, -2.0))
        L = pm.max(2)[0]                                   # (B,2T)
        lo = th - rs.unsqueeze(2) - rs.unsqueeze(1)
        U = torch.where(lo <= 0, torch.ones_like(lo), torch.cos(lo))
        ok = valid.unsqueeze(1) & valid.unsqueeze(2)
        bsq = (bw.double() ** 2).reshape(B, 1, 1)
        for eps in (1e-6,):
            keep_now = ok & (U >= L.unsqueeze(2) - bsq * math.log(N / eps))
            # mass criterion: per q cap sort x caps by U ascending, drop the longest prefix whose mass <= eps
            mass = torch.where(ok, 32.0 * torch.exp((U - L.unsqueeze(2)) / bsq), torch.zeros_like(U))
            Us, order = torch.sort(torch.where(ok, U, torch.full_like(U, 2.0)), dim=2)
            csum = torch.cumsum(torch.gather(mass, 2, order), 2)
            drop_sorted = csum <= eps
            drop = torch.zeros_like(drop_sorted).scatter_(2, order, drop_sorted)
            keep_mass = ok & ~drop
            def tiles(kp):
                k4 = kp.reshape(B, T, 2, T, 2)
                return k4.any(4).any(2).float().mean().item()
            # row-sum bound: r_i >= exp((L-1)/b^2) * R with R = sum over x caps of n_x exp((Lo - L)/b^2), Lo = lower
            # bound of every dot product of the cap pair (pm); n_x = 1 (rigorous without counts) or 16 (typical)
            res = []
            for nx in (1.0, 16.0):
                Rrow = (torch.where(ok, nx * torch.exp(((pm - L.unsqueeze(2)) / bsq).clamp(max=0.0)), torch.zeros_like(pm))).sum(2)
                Rrow = Rrow.clamp_min(1.0)
                drop_sorted2 = csum <= eps * Rrow.unsqueeze(2)
                drop2 = torch.zeros_like(drop_sorted2).scatter_(2, order, drop_sorted2)
                res.append((tiles(ok & ~drop2), float(Rrow.median()), float(Rrow.mean())))
            print("batch %d eps %.0e: tile pairs kept: N-bound %.4f  mass-bound %.4f  + row-sum bound n=1: %.4f (R median %.1f mean %.1f)  n=16: %.4f (R median %.1f)"
                  % (start, eps, tiles(keep_now), tiles(keep_mass), res[0][0], res[0][1], res[0][2], res[1][0], res[1][1]))
