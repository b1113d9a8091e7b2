"""Which rocBLAS products of a cfg5 step take more than 0.25 ms, and with what shapes?  (rocBLAS picks 40-50 TFLOP/s
kernels for some segment counts.)  python tools/probes/slow_gemm_probe.py"""
import collections
import os
import sys
import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parsenet_codebase_amd import dp, workloads  # noqa: E402

dp.limit_host_threads()
dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.warm_paths()
for _ in range(3):
    step.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(4):
        step.step()
    torch.cuda.synchronize()
seen = collections.Counter()
tot = collections.Counter()
for ev in prof.events():
    for k in ev.kernels:
        if "Cijk" in k.name:
            key = (ev.name, str(ev.input_shapes), k.name[:60])
            seen[key] += 1
            tot[key] += k.duration
print("rocBLAS products of four steps by (operator, shapes, kernel): launches, total ms, average us")
for key, t in tot.most_common(40):
    print("%3d %8.3f ms %8.1f us  %s %s  %s" % (seen[key], t / 1e3, t / seen[key], key[0], key[1], key[2]))
