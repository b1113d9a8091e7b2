"""Block-sparse mean-shift on the embedding the cfg5 bench really clusters (300 deterministic
pre-training steps): per-iteration plan statistics, per-kernel times (PN_PROF timers) and the
list-length distribution of the resident blocks.  python tools/ms_probe.py [dump path]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PARSENET_MS_STATS"] = "1"
import numpy as np
import torch
from parsenet_codebase_amd import workloads, _lib, kernels as K
from parsenet_codebase_amd import mean_shift as MSM
from parsenet_codebase_amd.fitting_batch import bandwidth_batch

dev = torch.device("cuda:0")
cache = os.environ.get("MS_PROBE_EMB")
if cache and os.path.exists(cache):
    st = torch.load(cache)
    e, bw = st["e"].to(dev).float(), st["bw"].to(dev)
else:
    step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
    step.model.eval()
    with torch.no_grad():
        emb, _, _ = step.model(step.x, step.labels, True)
        e = torch.nn.functional.normalize(emb.permute(0, 2, 1), dim=2).contiguous()
        bw, _ = bandwidth_batch(e, 0.025)
    if len(sys.argv) > 1:
        torch.save({"e": e.cpu(), "bw": bw.cpu()}, sys.argv[1])
print("bandwidths", [round(float(x), 4) for x in bw])
B, N, D = e.shape


def fwd_bwd():
    x = e.clone().requires_grad_(True)
    MSM.mean_shift_iterations(x, bw, 10).sum().backward()


for sparse in (1, 0):
    MSM.SPARSE = bool(sparse)
    fwd_bwd()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        fwd_bwd()
    t1.record()
    torch.cuda.synchronize()
    print("sparse=%d: 10 it fwd+bwd on B=%d: %.2f ms" % (sparse, B, t0.elapsed_time(t1) / 3))
    _lib.prof_enable(True)
    _lib.prof_reset()
    fwd_bwd()
    torch.cuda.synchronize()
    for name, (ms_, calls) in sorted(_lib.prof_results().items()):
        print("  %-24s %.4f ms per launch (%d launches)  %.3f ms total" % (name, ms_ / calls, calls, ms_))
    _lib.prof_enable(False)
    if sparse:
        for it, s in enumerate(MSM.LAST_PLAN_STATS):
            print("  it %d: pairs %.3f  lists fwd(256) %.3f rows(128) %.3f cols(256) %.3f" % ((it,) + tuple(s)))

# list lengths of the first and the last iteration's plan (row pass: 128-row blocks)
with torch.no_grad():
    perm = MSM.locality_order(e, 2)
    x = torch.gather(e, 1, perm.unsqueeze(2).expand(-1, -1, D))
    bsq = (bw ** 2).contiguous()
    xi = K.meanshift_x3_tileinfo(x)
    plan = K.meanshift_x3_plan(xi, xi, bsq, N)
    T = (N + 63) // 64 * 2
    nb0, nb1, nb2 = -(-N // 256), -(-N // 128), -(-N // 256)
    oc = (B * T * T + 255) // 256 * 256
    counts = plan[oc:oc + B * (nb0 + nb1 + nb2) * 4].view(torch.int32).reshape(B, -1).cpu().numpy()
    for name, sl in (("fwd/cols 256", slice(0, nb0)), ("rows 128", slice(nb0, nb0 + nb1))):
        c = np.sort(counts[:, sl].reshape(-1))
        print("%s-row blocks: %d lists of %d tiles: min %d  p10 %d  median %d  p90 %d  max %d  mean %.1f" % (
            name, c.size, T, c[0], c[c.size // 10], c[c.size // 2], c[9 * c.size // 10], c[-1], c.mean()))
    pairs = plan[:B * T * T].reshape(B, T, T)
    per_tile = pairs.float().sum(2).cpu().numpy().reshape(-1)
    print("per 32-row tile: active streamed tiles mean %.1f (of %d)" % (per_tile.mean(), T))
    rho = xi[1].cpu().numpy()
    print("tile angular radii: median %.3f  p90 %.3f  max %.3f ; bandwidth %.3f" % (
        np.median(rho), np.quantile(rho, 0.9), rho.max(), float(bw.mean())))
