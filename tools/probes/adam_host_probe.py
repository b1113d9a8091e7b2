"""Host time of the optimizer step inside a cfg5 step, un-profiled (the kernel trace shows a 0.6 ms gap between the
fused Adam's two launches under rocprofv3): python tools/probes/adam_host_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parsenet_codebase_amd import dp, workloads
dp.limit_host_threads()
dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=50, pool=16, pretrain_pool=64)
step.warm_paths()
for _ in range(3): step.step()
torch.cuda.synchronize()
opt = step.opt
orig = opt.step
acc = []
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); acc.append(time.perf_counter() - t0); return r
opt.step = timed
for _ in range(10): step.step()
torch.cuda.synchronize()
print("opt.step host time per call: %.3f ms (min %.3f), params %d" % (1e3*sum(acc)/len(acc), 1e3*min(acc), len(list(step.model.parameters()))))
