"""One kNN shape a few times (the process rocprofv3 is pointed at): python3 tools/probes/knn_one.py B C N k [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from parsenet_codebase_amd import kernels

B, C, N, k = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, C, N, device="cuda", generator=g) * (0.5 + torch.rand(B, C, 1, device="cuda", generator=g))
for _ in range(iters):
    idx = kernels.knn(x, k, "feature")
torch.cuda.synchronize()
print(int(idx.sum()))
