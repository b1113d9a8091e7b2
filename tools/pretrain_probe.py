"""How much seg-only pre-training on DISJOINT shapes does the cfg5 stand-in need before the
embedding of HELD-OUT shapes has cluster structure?  Trains in chunks on the pre-training pool
(ids workloads.PRETRAIN_FIRST_SHAPE..) and after every chunk runs the clustering + fitting stage on
the timed pool (ids 0..15) and on 8 shapes of the training pool: clusters and fitted segments per
shape, block-sparse tile-pair fraction, losses.   python tools/pretrain_probe.py [lr] [pool]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from parsenet_codebase_amd import mean_shift as MSM, workloads

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
ppool = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pool=16)
step.opt = torch.optim.Adam(step.model.parameters(), lr=lr)
os.environ["PARSENET_MS_STATS"] = "1"


def evaluate(first, pool, tag):
    step.load_pool(first, pool)
    step.model.eval()
    ev = step.evaluation
    ev.stats.update(shapes=0, clusters=0, fitted=0)
    pairs, losses = [], []
    with torch.no_grad():
        for _ in range(pool // step.batch):
            step.next_batch()
            emb, logp, el = step.model(step.x, step.labels, True)
            MSM.LAST_PLAN_STATS = None
            res = ev.fitting_losses(emb.permute(0, 2, 1), step.points, step.normals, step.labels, step.prim_np, logp,
                                    quantile=0.025, iterations=10, lamb=0.1)
            losses += [float(r[0][0]) for r in res]
            if MSM.LAST_PLAN_STATS:
                st = MSM.LAST_PLAN_STATS
                pairs.append(sum(t[0] for t in st) / len(st))
    n = max(ev.stats["shapes"], 1)
    print("  %-9s clusters/shape %5.2f  fitted/shape %5.2f  tile pairs %s  residual loss %.4f  triplet %.3f"
          % (tag, ev.stats["clusters"] / n, ev.stats["fitted"] / n,
             " ".join("%.3f" % p for p in pairs), float(np.mean(losses)), float(el.mean())), flush=True)


done = 0
np.random.seed(4321)
for chunk in (150, 150, 300, 600, 800):
    step.load_pool(workloads.PRETRAIN_FIRST_SHAPE, ppool)
    step.model.train()
    t0 = time.time()
    for _ in range(chunk):
        loss = step.seg_step()
    torch.cuda.synchronize()
    done += chunk
    print("after %d steps (lr %g, %d training shapes): train loss %.4f, %.1f ms/step"
          % (done, lr, ppool, float(loss), 1e3 * (time.time() - t0) / chunk), flush=True)
    state = np.random.get_state()
    evaluate(0, 16, "held-out")
    evaluate(workloads.PRETRAIN_FIRST_SHAPE, 8, "training")
    np.random.set_state(state)
