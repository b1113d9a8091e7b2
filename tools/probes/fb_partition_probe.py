"""How robust is the clustering of test_batched_stage_equals_shape_by_shape[10000-False] to the noise of its two
paths?  For several embedding noises / shape ids: cluster counts and partition agreement, sequential vs batched."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_fitting_batch_gpu import _structured_batch, _evaluation  # noqa: E402


def main():
    gpu = torch.device("cuda:0")
    torch.cuda.set_device(gpu)
    N, B = 10000, 3
    for seeds in ((3, 8, 21), (4, 9, 22), (5, 11, 30)):
        for noise in (0.04, 0.035, 0.03):
            P, Nn, lab, prim, emb, logp = _structured_batch(gpu, B, N, seeds, noise=noise)
            ev = _evaluation(gpu)
            out = {}
            for mode in ("sequential", "batched"):
                ev.batched = mode == "batched"
                e = emb.clone().requires_grad_(True)
                np.random.seed(5)
                if ev.batched:
                    res = ev.fitting_losses(e, P, Nn, lab, prim, logp, quantile=0.025, iterations=10, lamb=0.1)
                else:
                    res = [ev.fitting_loss(e[b:b + 1], P[b:b + 1], Nn[b:b + 1], lab[b:b + 1], prim[b:b + 1],
                                           logp[b:b + 1], quantile=0.025, iterations=10, lamb=0.1) for b in range(B)]
                out[mode] = res
            line = []
            for b in range(B):
                la, lc = out["sequential"][b][1][1], out["batched"][b][1][1]
                t = np.zeros((la.max() + 1, lc.max() + 1), np.int64)
                np.add.at(t, (la, lc), 1)
                agree = max(t.max(1).sum(), t.max(0).sum()) / la.size
                line.append("%d/%d %.4f (%.4f %.4f)" % (la.max() + 1, lc.max() + 1, agree,
                                                        float(out["sequential"][b][0][0]), float(out["batched"][b][0][0])))
            print(seeds, noise, " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
