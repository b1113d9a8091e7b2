"""Can a SplineNet training step (cfg2 / cfg3) be captured in a HIP graph, are the replays equal to eager steps bit
for bit, and what does a replayed step cost?  python tools/probes/spline_graph_probe.py [closed]"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parsenet_codebase_amd import dp, workloads  # noqa: E402

dp.limit_host_threads()
dev = torch.device("cuda:0")
closed = len(sys.argv) > 1 and sys.argv[1] == "closed"


def make():
    s = workloads.SplineNetStep(dev, closed=closed)
    s.opt = torch.optim.Adam(list(s.model.parameters()), lr=1e-3, fused=True, capturable=True)
    return s


def eager(s, n):
    out = []
    for _ in range(n):
        out.append(s.step().detach().clone())
    return out


a, b = make(), make()
ref = eager(a, 8)
torch.cuda.synchronize()
t0 = time.perf_counter()
eager(a, 40)
torch.cuda.synchronize()
print("eager: %.3f ms per step" % ((time.perf_counter() - t0) / 40 * 1e3))

# b: three eager steps on a side stream (warm-up), then capture one step and replay
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    got = eager(b, 3)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = b.step()
for _ in range(5):
    g.replay()
    got.append(loss.detach().clone())
torch.cuda.synchronize()
# (the capture itself does not run the step: replays are steps 4 .. 8)
for i, (x, y) in enumerate(zip(ref, got)):
    print(i, float(x), float(y), "equal" if torch.equal(x, y) else "DIFFERENT")
t0 = time.perf_counter()
for _ in range(40):
    g.replay()
torch.cuda.synchronize()
print("graph replay: %.3f ms per step" % ((time.perf_counter() - t0) / 40 * 1e3))
