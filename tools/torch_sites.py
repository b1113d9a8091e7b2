"""GPU time of the torch-side kernels (everything that is not a pn_* kernel or a GEMM) of one cfg5
step by the Python line of this package that issued them: python tools/torch_sites.py"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from parsenet_codebase_amd import dp, workloads

dp.limit_host_threads()       # as bench.py and the trainer run: one intra-op CPU thread

dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.warm_paths()
for _ in range(3):
    step.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step.step()
    torch.cuda.synchronize()
sites = collections.Counter()
counts = collections.Counter()
ops = collections.Counter()          # (site, operator) -> us
opn = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if not ev.kernels:
        continue
    t = sum(k.duration for k in ev.kernels if not (k.name.startswith("pn_") or k.name.startswith("void pn_") or "Cijk" in k.name))
    n = sum(1 for k in ev.kernels if not (k.name.startswith("pn_") or k.name.startswith("void pn_") or "Cijk" in k.name))
    if n == 0:
        continue
    site = "(autograd / no python frame)"
    node = ev
    while node is not None and site.startswith("("):    # the innermost enclosing fit:* / module range otherwise
        for fr in node.stack or []:
            if "parsenet_codebase_amd" in fr and "site-packages" not in fr:
                site = fr.replace(root + "/", "")
                break
        node = node.cpu_parent
    if site.startswith("("):
        node = ev.cpu_parent
        names = []
        while node is not None:
            names.append(node.name)
            node = node.cpu_parent
        site = "(no frame) " + " < ".join(names[:3])
    sites[site] += t
    counts[site] += n
    ops[(site, ev.name)] += t
    opn[(site, ev.name)] += n
tot = sum(sites.values())
print("torch-side kernels of one step: %.2f ms in %d launches" % (tot / 1e3, sum(counts.values())))
for s, t in sites.most_common(45):
    print("%8.3f ms %5d  %s" % (t / 1e3, counts[s], s[:150]))
print("\nby (site, operator):")
for (site, name), t in ops.most_common(60):
    print("%8.3f ms %5d  %-34s %s" % (t / 1e3, opn[(site, name)], name[:34], site[:110]))
