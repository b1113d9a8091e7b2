"""Every device copy / memset / fill of ONE cfg5 step (no duration threshold), counted by the package frames
that issued it, plus the raw count of device-side copy activities the profiler saw (attached to an operator
or not): python tools/probes/copy_census.py"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from parsenet_codebase_amd import dp, workloads

dp.limit_host_threads()
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
raw = collections.Counter()
for e in prof.profiler.kineto_results.events():
    if str(e.device_type()).endswith("CUDA"):
        nm = e.name()
        if "emcpy" in nm or "emset" in nm or "copyBuffer" in nm or "fillBuffer" in nm:
            raw[nm[:60]] += 1
print("raw device-side copy activities of one step:", dict(raw))
sites = collections.Counter()
times = collections.Counter()
for ev in prof.events():
    for k in ev.kernels or []:
        nm = k.name
        kind = ("memcpy" if "emcpy" in nm else "memset" if "emset" in nm else "copy kernel" if "copy" in nm.lower()
                else "fill" if "FillFunctor" in nm else None)
        if kind is None:
            continue
        frames = [fr.replace(root + "/", "").replace("parsenet_codebase_amd/", "") for fr in (ev.stack or [])
                  if "parsenet_codebase_amd" in fr or "bench.py" in fr][:3]
        node, names = ev.cpu_parent, []
        while node is not None and len(names) < 3:
            names.append(node.name)
            node = node.cpu_parent
        key = (kind, ev.name[:16], str(ev.input_shapes)[:60], " < ".join(frames) or "(no frame) " + " < ".join(names))
        sites[key] += 1
        times[key] += k.duration
print("attributed: %d activities, %.3f ms" % (sum(sites.values()), sum(times.values()) / 1e3))
for key, n in sorted(sites.items(), key=lambda kv: -times[kv[0]]):
    print("%4d x %7.1f us  %-11s %-16s %-60s %s" % (n, times[key] / n, key[0], key[1], key[2], key[3][:170]))
