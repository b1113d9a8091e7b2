"""The SIGN of LAPACK's minor-axis eigenvector follows the last bits of the covariance (round 6).

standardize_point_torch (src/fitting_utils.py:532-540) takes the eigenvector of the smallest eigenvalue of a 3 x 3
covariance from geev and rotates it to +x; the eigenvector's sign is whatever LAPACK returns.  This probe takes the
golden standardisation case, forms the covariance as round 5's fp32 expressions do (rocBLAS product), and asks the
box's own LAPACK for the minor axis of (i) that matrix, (ii) the same sums accumulated in float64 and rounded once,
(iii) the matrix a fused kernel of this round produced for the same data (fp64 sums around an fp64-summed mean; it
differs from (i) by 3 ... 7 ulp in the off-diagonal entries) and (iv) (i) with single entries moved by +-1 / +-2 ulp.
On the evidence box (iii) comes out with the OPPOSITE sign — the canonical frame of the spline patch turns by 180
degrees, the golden R and the end-to-end loss against the oracle move with it — while (ii) and all of (iv) keep it
(profiles/r06_std_sign_probe.txt): the sign is a discontinuous function of the matrix a few ulp away from this
input.  That is why fitting_batch.standardize_segments keeps the fp32 expressions (and their bits) for mean,
covariance and rotation and fuses only the selection and the extents (csrc/fused.hip: pn_standardize_*)."""
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
cov32 = torch.bmm((Pc * s_.unsqueeze(2)).transpose(1, 2), Pc).cpu()[0]
cov64 = torch.bmm((Pc * s_.unsqueeze(2)).double().transpose(1, 2), Pc.double()).float().cpu()[0]
fused = torch.tensor([[3.4391919e-01, 1.3438415e-03, 3.4097077e-03], [1.3438415e-03, 5.1762843e+00, 1.1162090e-01],
                      [3.4097077e-03, 1.1162090e-01, 1.3100487e+00]], dtype=torch.float32)


def minor_axis(c):
    wv, v = torch.linalg.eig(c.unsqueeze(0))
    k = int(torch.min(wv.real, 1)[1][0])
    return v.real.numpy()[0][:, k]


ref = minor_axis(cov32)
for name, c in (("(i)   fp32 product", cov32), ("(ii)  float64 sums", cov64), ("(iii) fused kernel of round 6", fused)):
    a = minor_axis(c)
    print("%-32s minor axis %s  %s" % (name, a, "SAME sign" if float(np.dot(a, ref)) > 0 else "OPPOSITE sign"))
flips = total = 0
for (i, j) in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)):
    for ulps in (-2, -1, 1, 2):
        c = cov32.clone()
        v = c[i, j].numpy().copy()
        for _ in range(abs(ulps)):
            v = np.nextafter(v, np.float32(np.inf if ulps > 0 else -np.inf), dtype=np.float32)
        c[i, j] = float(v)
        c[j, i] = float(v)
        total += 1
        flips += float(np.dot(minor_axis(c), ref)) < 0
print("(iv)  single entries of (i) moved by +-1 / +-2 ulp: %d of %d perturbations flip the sign" % (flips, total))
print("fixture R (the reference's frame)\n", g["std_R"])
