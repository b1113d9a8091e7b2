"""Paired row pass (csrc/meanshift_rows2.h) against the round-3 row pass: bit identity of one
backward iteration (planned and dense launches), then launch times of both on the benchmark's shape."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parsenet_codebase_amd import _lib
from parsenet_codebase_amd import mean_shift as MSM

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N = 4, 10000
g = torch.Generator().manual_seed(1)
proto = torch.nn.functional.normalize(torch.randn(9, 128, generator=g), dim=1)
lab = torch.randint(0, 9, (B, N), generator=g)
X = torch.nn.functional.normalize(proto[lab] + 0.45 * torch.randn(B, N, 128, generator=g) / 128 ** 0.5, dim=2).to(dev)
G = torch.randn(B, N, 128, generator=g).to(dev)
bw = torch.full((B,), 0.21, device=dev)
for sparse in (True, False):
    MSM.SPARSE = sparse
    outs = {}
    for mode in ("0", "1"):
        os.environ["PN_MS_ROWS2"] = mode
        x = X.clone().requires_grad_(True)
        y = MSM.mean_shift_iterations(x, bw, 10)
        (y * G).sum().backward()
        outs[mode] = (y.detach().clone(), x.grad.clone())
    same = torch.equal(outs["0"][1], outs["1"][1]) and torch.equal(outs["0"][0], outs["1"][0])
    gmax = float(outs["0"][1].abs().max())
    print("sparse=%s: gradients bit-identical: %s (max |diff| %.3e of max |grad| %.3e = %.2e rel)" % (
        sparse, same, float((outs["0"][1] - outs["1"][1]).abs().max()), gmax,
        float((outs["0"][1] - outs["1"][1]).abs().max()) / gmax))
    for mode in ("0", "1", "0", "1"):
        os.environ["PN_MS_ROWS2"] = mode
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(3):
            x = X.clone().requires_grad_(True)
            (MSM.mean_shift_iterations(x, bw, 10) * G).sum().backward()
        torch.cuda.synchronize()
        r = _lib.prof_results(); _lib.prof_enable(False)
        print("  PN_MS_ROWS2=%s  rows %.4f ms  cols %.4f ms  fwd %.4f ms per launch" % (
            mode, r["meanshift_bwd_rows"][0] / r["meanshift_bwd_rows"][1], r["meanshift_bwd_cols"][0] / r["meanshift_bwd_cols"][1],
            r["meanshift_fwd"][0] / r["meanshift_fwd"][1]))
