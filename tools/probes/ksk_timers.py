"""Phase timers of the small-k kNN kernel (library built with PN_EXTRA_HIPCC_FLAGS=-DKSK_TIMERS):
shader cycles of wave 0 of every workgroup, by phase.  python tools/probes/ksk_timers.py B C N [k]"""
import ctypes
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parsenet_codebase_amd import kernels as K, _lib  # noqa: E402

NAMES = ["total", "prologue", "stage issue", "tile (mfma + rows)", "barrier", "flush", "epilogue", "workgroups"]


def main():
    B, C, N = (int(a) for a in sys.argv[1:4])
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libparsenet_hip.so"))
    buf = (ctypes.c_ulonglong * 8)()
    x = torch.randn(B, C, N, device="cuda")
    for _ in range(3):
        K.knn(x, k, "feature", int32=True)
    torch.cuda.synchronize()
    assert lib.pn_knn_smallk_timers(buf, 1) == 0
    reps = 5
    for _ in range(reps):
        K.knn(x, k, "feature", int32=True)
    torch.cuda.synchronize()
    assert lib.pn_knn_smallk_timers(buf, 1) == 0
    t = np.array(list(buf), dtype=np.float64)
    wg = t[7] / reps
    print(f"B={B} C={C} N={N} k={k}: {wg:.0f} workgroups per launch; cycles per workgroup (wave 0):")
    for i in range(7):
        print(f"  {NAMES[i]:22s} {t[i] / t[7]:12.0f}  {100 * t[i] / t[0]:5.1f} %")


if __name__ == "__main__":
    main()
