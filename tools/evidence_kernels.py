"""The three kernels BASELINE.json's north_star names (edge-feature gather, fused edge-conv
gather-reduce, Chamfer nearest neighbour) at the benchmark sizes, a few launches each — the
workload rocprofv3 is pointed at for profiles/r02_named_kernels_*.csv:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/evidence_kernels.py
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... (own pass)
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (own passes)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parsenet_codebase_amd import kernels

dev = torch.device("cuda:0")
torch.manual_seed(0)
REP = 5
B, N, k, C = 4, 10000, 80, 64
x = torch.randn(B, C, N, device=dev)
idx = kernels.knn(x, k, "feature")                    # a real kNN graph of the features
xt = x.transpose(1, 2).contiguous()
for _ in range(REP):
    feat = kernels.edge_feature_fwd(xt, idx)
del feat
PQ = torch.randn(B, N, 2 * C, device=dev)
gamma = torch.ones(C, device=dev)
for _ in range(REP):
    kernels.edgeconv_reduce_fwd(PQ, idx, gamma, 2, True)
a, b = torch.rand(1, 10000, 3, device=dev), torch.rand(1, 10000, 3, device=dev)
for _ in range(REP):
    kernels.chamfer_nn(a, b)
a, b = torch.rand(32, 1600, 3, device=dev), torch.rand(32, 700, 3, device=dev)
for _ in range(REP):
    kernels.chamfer_nn(a, b)
torch.cuda.synchronize()
