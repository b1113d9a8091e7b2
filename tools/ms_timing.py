import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parsenet_codebase_amd import _lib
from parsenet_codebase_amd.mean_shift import MeanShift
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 10000
X = torch.nn.functional.normalize(torch.randn(N, 128, device=dev), dim=1)
ms = MeanShift()
b = torch.tensor(0.3, device=dev)
x = X.clone().requires_grad_(True)
y, _ = ms.mean_shift_(x, b, 3)
y.sum().backward()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else _lib.load()
buf = (ctypes.c_ulonglong * 24)()
lib.pn_ms_debug_read.argtypes = [ctypes.c_void_p]
print("rc", lib.pn_ms_debug_read(buf))
for p in range(3):
    g1, g2, bar, tot, nt, ew, dma, bar1 = [buf[p * 8 + i] for i in range(8)]
    nt = max(nt, 1)
    print("PASS %d: tiles %d  per tile: G1 %.0f  EW %.0f  G2 %.0f  barrier %.0f  | loop total/tile %.0f (shader clocks)" %
          (p, nt, g1 / nt, ew / nt, g2 / nt, bar / nt, tot / nt) +
          ("  [dma issue %.0f  of the barrier: wait for the DMA %.0f]" % (dma / nt, bar1 / nt) if dma else ""))
