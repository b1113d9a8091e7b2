"""The SIGN of LAPACK's minor-axis eigenvector follows the last bit of the covariance (round 6): for the golden
standardisation case, the covariance of the selected centred points as round 5's fp32 expressions form it (rocBLAS
product) next to the same sums accumulated in float64 and rounded once, and geev's eigenvectors for both — on the
box's own LAPACK.  On the evidence boxes the two minor axes come out with OPPOSITE signs (the canonical frame of the
spline patch turns by 180 degrees), which is why fitting_batch.standardize_segments keeps the fp32 expressions for
mean / covariance / rotation and fuses only the selection and the extents (csrc/fused.hip: pn_standardize_*)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from parsenet_codebase_amd import kernels as K

np.set_printoptions(precision=9, linewidth=200)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "fitting.npz"))
dev = torch.device("cuda:0")
P = torch.from_numpy(g["std_P"]).float().reshape(1, -1, 3).to(dev)
w = torch.from_numpy(g["std_w"]).float().reshape(1, -1).to(dev)
n = P.shape[1]
EPS = float(np.finfo(np.float32).eps)
kf = n // 4 if n >= 7500 else n // 2
sel = K.standardize_select(w, kf).bool()
hi = w > 0.8
cnt = hi.sum(1, keepdim=True)
top = torch.topk(w, kf, dim=1)[1]
fb = torch.zeros_like(hi).scatter_(1, top, torch.ones_like(top, dtype=torch.bool))
sel0 = torch.where(cnt < 400, fb, hi)
print("confident", int(cnt), "selection of the kernel equals topk's:", bool((sel == sel0).all()), int(sel.sum()))
s_ = sel0.float()
wsel = w * s_
mean0 = (P * wsel.unsqueeze(2)).sum(1) / (wsel.sum(1, keepdim=True) + EPS)
Pc = P - mean0.unsqueeze(1)
cov32 = torch.bmm((Pc * s_.unsqueeze(2)).transpose(1, 2), Pc)
Pd = (Pc * s_.unsqueeze(2)).double()
cov64 = torch.bmm(Pd.transpose(1, 2), Pc.double()).float()
print("cov (fp32 product)\n", cov32.cpu().numpy()[0], "\ncov (float64 sums, rounded once)\n", cov64.cpu().numpy()[0])
for name, c in (("fp32 product ", cov32), ("float64 sums ", cov64)):
    wv, v = torch.linalg.eig(c.cpu())
    k = int(torch.min(wv.real, 1)[1][0])
    print(name, "eigenvalues", wv.real.numpy()[0], "minor axis", v.real.numpy()[0][:, k])
print("fixture R (the reference's frame)\n", g["std_R"])
