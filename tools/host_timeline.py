"""Host side of one cfg5 step: CPU time of the record_function ranges (fit:*), of the synchronising
calls and of the busiest operators — where the GPU idles it is waiting for this thread.
python tools/host_timeline.py"""
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
for _ in range(4):
    step.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step.step()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.cpu_time_total)
print("%-58s %8s %10s %10s" % ("range / op", "calls", "cpu ms", "self ms"))
for e in rows:
    if e.key.startswith("fit:") or "ynchronize" in e.key or "item" in e.key or "_local_scalar" in e.key or "copy_" in e.key or "to" == e.key:
        print("%-58s %8d %10.3f %10.3f" % (e.key[:58], e.count / 3, e.cpu_time_total / 3e3, e.self_cpu_time_total / 3e3))
print("--- top self CPU")
for e in sorted(ev, key=lambda e: -e.self_cpu_time_total)[:30]:
    print("%-58s %8d %10.3f %10.3f" % (e.key[:58], e.count / 3, e.cpu_time_total / 3e3, e.self_cpu_time_total / 3e3))
