"""Randomised parity sweep of the index-producing kernels against the C oracle and of the two
mean-shift arithmetics against each other: python tools/fuzz.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import cbind
from parsenet_codebase_amd import kernels as K
import parsenet_codebase_amd.mean_shift as MS

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(time.time()) % 100000)
t0 = time.time()
n = {"knn": 0, "pn": 0, "argmax": 0, "kth": 0, "chamfer": 0, "ms": 0}
while time.time() - t0 < budget:
    # kNN, feature metric
    B, C, N = rng.randint(1, 3), int(rng.choice([3, 6, 17, 64, 128, 200])), int(rng.randint(40, 1500))
    k = int(rng.randint(1, min(N, 90)))
    x = (rng.randn(B, C, N) * rng.choice([0.1, 1.0])).astype(np.float32)
    if rng.rand() < 0.2:
        x[:, :, rng.randint(0, N, N // 3)] = x[:, :, :1]       # masses of duplicates
    got = K.knn(torch.from_numpy(x).to(dev), k, "feature").cpu().numpy()
    assert np.array_equal(got, cbind.knn(x, k, 0)), ("knn", B, C, N, k)
    n["knn"] += 1
    # the same on shapes that take the bf16 passes (>= 2048 candidates, <= 128 channels), at a
    # random PN_KNN_X3 level, with clumped rows (neighbours closer than the error of an
    # approximate distance) and duplicates
    os.environ["PN_KNN_X3"] = str(int(rng.randint(0, 3)))
    B, C, N = int(rng.randint(1, 3)), int(rng.choice([17, 40, 64, 100, 128])), int(rng.randint(2048, 5200))
    k = int(rng.choice([1, 10, 20, 80, 128]))
    x = (rng.randn(B, C, N) * rng.choice([0.1, 1.0, 5.0])).astype(np.float32)
    if rng.rand() < 0.5:
        cen = (rng.randn(B, C, 9) * 3).astype(np.float32)
        lab = rng.randint(0, 9, (B, N))
        x = np.take_along_axis(cen, lab[:, None, :].repeat(C, 1), 2) + x * np.float32(rng.choice([1e-5, 1e-3, 0.1]))
    if rng.rand() < 0.3:
        x[:, :, rng.randint(0, N, N // 5)] = x[:, :, :1]
    x = np.ascontiguousarray(x.astype(np.float32))
    got = K.knn(torch.from_numpy(x).to(dev), k, "feature").cpu().numpy()
    assert np.array_equal(got, cbind.knn(x, k, 0)), ("knn_x3", os.environ["PN_KNN_X3"], B, C, N, k)
    os.environ.pop("PN_KNN_X3")
    n["knn_x3"] = n.get("knn_x3", 0) + 1
    # small k (k <= 16): the one-pass kernel of csrc/knn_smallk.h on every instance (2 / 4 / 33 / 65 / 129 k-steps,
    # four and eight waves, sliced and unsliced candidate ranges), clumped rows, duplicates, lattices of ties
    C = int(rng.choice([1, 2, 3, 5, 7, 8, 40, 64, 65, 128, 129, 200, 256]))
    N = int(rng.randint(32, 3200 if C > 128 else 6000))
    B, k = int(rng.randint(1, 4)), int(rng.randint(1, 17))
    x = (rng.randn(B, C, N) * rng.choice([0.1, 1.0, 5.0])).astype(np.float32)
    mode = rng.rand()
    if mode < 0.3:
        cen = (rng.randn(B, C, 7) * 3).astype(np.float32)
        lab = rng.randint(0, 7, (B, N))
        x = np.take_along_axis(cen, lab[:, None, :].repeat(C, 1), 2) + x * np.float32(rng.choice([0.0, 1e-5, 1e-3, 0.1]))
    elif mode < 0.5:
        x = (rng.randint(-4, 5, (B, C, N)) / 8.0).astype(np.float32)
    if rng.rand() < 0.3:
        x[:, :, rng.randint(0, N, N // 4)] = x[:, :, :1]
    x = np.ascontiguousarray(x.astype(np.float32))
    got = K.knn(torch.from_numpy(x).to(dev), k, "feature", int32=bool(rng.rand() < 0.5)).cpu().numpy()
    assert np.array_equal(got, cbind.knn(x, k, 0)), ("knn_smallk", B, C, N, k)
    n["knn_smallk"] = n.get("knn_smallk", 0) + 1
    # points + normals metric
    N = int(rng.randint(60, 1200))
    k = int(rng.randint(1, min(N, 81)))
    p = rng.uniform(-0.5, 0.5, (1, 3, N)).astype(np.float32)
    nr = rng.randn(1, 3, N).astype(np.float32)
    nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    xx = np.concatenate([p, nr], 1)
    got = K.knn(torch.from_numpy(xx).to(dev), k, "points_normals").cpu().numpy()
    assert np.array_equal(got, cbind.knn(xx, k, 1)), ("knn_pn", N, k)
    n["pn"] += 1
    # dot-product arg-max and k-th value
    Nq, Nc, Cd = int(rng.randint(10, 1500)), int(rng.randint(10, 1500)), int(rng.choice([8, 64, 128]))
    q = rng.randn(1, Nq, Cd).astype(np.float32)
    c = rng.randn(1, Nc, Cd).astype(np.float32)
    if rng.rand() < 0.3:
        c[:, rng.randint(0, Nc, Nc // 2)] = c[:, :1]
    res = K.dot_select(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev), 1, want_value=False)
    if res is not None:
        idx, flags = res
        ok = flags[0].cpu().numpy() == 0
        want = cbind.dot_argmax(c[0], q[0])
        assert np.array_equal(idx[0, :, 0].cpu().numpy()[ok], want[ok]), ("argmax", Nq, Nc, Cd)
        n["argmax"] += 1
    kk = int(rng.randint(1, min(Nq, 300)))
    res = K.dot_select(torch.from_numpy(q).to(dev), torch.from_numpy(q).to(dev), kk, want_value=True)
    if res is not None:
        val, flags = res
        ok = flags[0].cpu().numpy() == 0
        want = cbind.kth_largest_dot(q[0], kk)
        assert np.array_equal(val[0].cpu().numpy()[ok], want[ok]), ("kth", Nq, Cd, kk)
        n["kth"] += 1
    # Chamfer arg-mins
    Na, Nb = int(rng.randint(1, 3000)), int(rng.randint(1, 3000))
    a = rng.uniform(-1, 1, (1, Na, 3)).astype(np.float32)
    bb = rng.uniform(-1, 1, (1, Nb, 3)).astype(np.float32)
    ga = K.chamfer_nn(torch.from_numpy(a).to(dev), torch.from_numpy(bb).to(dev))
    wa = cbind.chamfer_nn(a, bb)
    for g_, w_ in zip(ga, wa):
        assert np.array_equal(g_.cpu().numpy(), w_), ("chamfer", Na, Nb)
    n["chamfer"] += 1
    # mean-shift: the three arithmetics agree (random sizes, bandwidths, gradient scales)
    Nm = int(rng.randint(5, 1200))
    X = torch.nn.functional.normalize(torch.randn(Nm, 128), dim=1)
    bw = float(rng.uniform(0.1, 0.9))
    w = torch.randn(Nm, 128) * float(10.0 ** rng.uniform(-6, 6))
    its = int(rng.randint(1, 5))
    out = {}
    for mode in ("f32", "bf16x3", "fp16x2"):
        MS.ARITH = mode
        xg = X.to(dev).requires_grad_(True)
        y = MS.mean_shift_iterations(xg, bw, its)
        (y * w.to(dev)).sum().backward()
        out[mode] = (y.detach(), xg.grad.detach())
    for mode in ("bf16x3", "fp16x2"):
        dy = float((out["f32"][0] - out[mode][0]).abs().max())
        dg = float((out["f32"][1] - out[mode][1]).abs().max() / (out["f32"][1].abs().max() + 1e-30))
        assert dy < 2e-6 and dg < 1e-4, ("meanshift", mode, Nm, bw, its, dy, dg)
    # bandwidth statistic on the fp16 cores against the exact engine
    if Nm >= 700:
        MS.ARITH = "fp16x2"
        np.random.seed(1)
        b1 = float(MS.MeanShift().compute_bandwidth(X.to(dev), 10000, 0.002))
        MS.ARITH = "f32"
        np.random.seed(1)
        b0 = float(MS.MeanShift().compute_bandwidth(X.to(dev), 10000, 0.002))
        assert abs(b1 - b0) <= 1e-6 * abs(b0), ("bandwidth", Nm, b0, b1)
    n["ms"] += 1
print("fuzz ok", n, "in %.0f s" % (time.time() - t0))
