"""Which tensor-library copies / fills / device-to-device memcpys of ONE cfg5 step are large, and where they are
issued: every profiled operator whose device activity is a copy, a memcpy, a memset or a fill, with its input
shapes and the first package frames of its stack — by device time.  python tools/probes/copy_sites.py [min us]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from parsenet_codebase_amd import dp, workloads

dp.limit_host_threads()
min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.warm_paths()
for _ in range(3):
    step.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step.step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = []
census = collections.Counter()
census_t = collections.Counter()
for ev in prof.events():
    if not ev.kernels:
        continue
    for k in ev.kernels:
        nm = k.name
        kind = ("memcpy" if "Memcpy" in nm else "memset" if "Memset" in nm else "copy kernel" if "copy" in nm.lower()
                else "fill" if "FillFunctor" in nm else None)
        if kind is None:
            continue
        census[kind] += 1
        census_t[kind] += k.duration
        if k.duration < min_us:
            continue
        frames = [fr.replace(root + "/", "") for fr in (ev.stack or []) if "parsenet_codebase_amd" in fr or "bench.py" in fr][:4]
        node, names = ev.cpu_parent, []
        while node is not None and len(names) < 4:
            names.append(node.name)
            node = node.cpu_parent
        rows.append((k.duration, kind, ev.name, str(ev.input_shapes)[:90], " < ".join(frames) or "(no package frame) " + " < ".join(names)))
print("device copies / fills of one step:", {k: (census[k], round(census_t[k] / 1e3, 3)) for k in census}, "(count, ms)")
for d, kind, name, shapes, where in sorted(rows, reverse=True)[:60]:
    print("%8.1f us  %-11s %-18s %-90s %s" % (d, kind, name[:18], shapes, where[:200]))
