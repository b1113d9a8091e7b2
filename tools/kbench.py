"""Micro-benchmarks of individual HIP kernels (HIP-event timing on torch's current stream).
Usage: python tools/kbench.py [knn] [chamfer] [edge] [meanshift] ..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parsenet_codebase_amd import kernels


def timeit(fn, warmup=2, iters=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_knn():
    dev = torch.device("cuda:0")
    for (B, C, N, k, metric) in [(4, 64, 10000, 80, "feature"), (4, 6, 10000, 80, "points_normals"),
                                 (4, 3, 10000, 80, "feature"), (32, 64, 700, 10, "feature"),
                                 (32, 3, 700, 10, "feature"), (32, 128, 700, 10, "feature")]:
        x = torch.randn(B, C, N, device=dev)
        if metric == "points_normals":
            x[:, 3:] = torch.nn.functional.normalize(x[:, 3:], dim=1)
        ms = timeit(lambda: kernels.knn(x, k, metric))
        flops = B * N * N * (2 * C + 3)
        print("knn B=%d C=%d N=%d k=%d %s: %.3f ms  %.1f TFLOP/s" % (B, C, N, k, metric, ms, flops / ms / 1e9))


def bench_gemm():
    """Per-point layers: y = W x on the bf16 matrix cores (csrc/gemm_x3.hip: split image of x + GEMM) against the
    rocBLAS fp32 product it replaces (torch.bmm with batch stride 0 on the weight), forward shapes of cfg3 / cfg5."""
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    for (B, M, Kd, N, what) in [(32, 1024, 1152, 700, "cfg3 conv5"), (32, 1152, 1024, 700, "cfg3 conv5, gradient w.r.t. x"),
                                (4, 1024, 256, 10000, "cfg5 mlp1"), (4, 512, 256, 10000, "cfg5 conv1 (local part)"),
                                (4, 256, 512, 10000, "cfg5 conv2"), (4, 256, 256, 10000, "cfg5 seg_prob1"),
                                (4, 128, 256, 10000, "cfg5 seg_prob2"), (10, 1024, 512, 2500, "cfg5 open SplineNet conv5"),
                                (10, 1024, 1152, 2500, "cfg5 closed SplineNet conv5"), (32, 1024, 256, 700, "cfg3 layer 4 half"),
                                (6, 1024, 1152, 5000, "cfg5 SplineNet conv5, 6 segments"),
                                (3, 1024, 1152, 5000, "cfg5 SplineNet conv5, 3 segments"),
                                (4, 1024, 1152, 5000, "cfg5 SplineNet conv5, 4 segments"),
                                (1, 1024, 1152, 5000, "cfg5 SplineNet conv5, 1 segment"),
                                (5, 1024, 512, 5000, "cfg5 SplineNet layer, 5 segments")]:
        w = torch.randn(M, Kd, device=dev)
        x = torch.randn(B, Kd, N, device=dev)
        img = kernels.gemm_x3_weight_image(w)
        t_x3 = timeit(lambda: kernels.gemm_x3(img, M, x))
        we = w.unsqueeze(0).expand(B, -1, -1)
        t_bl = timeit(lambda: torch.bmm(we, x))
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(5):
            kernels.gemm_x3(img, M, x)
        torch.cuda.synchronize()
        pr = {kn: t / calls for kn, (t, calls) in _lib.prof_results().items()}
        _lib.prof_enable(False)
        gf = 2.0 * M * Kd * B * N / 1e9
        print("%-34s M=%d K=%d points=%d: bf16x3 %.3f ms (image %.3f + GEMM %.3f = %.0f TFLOP/s fp32-equivalent, %.2f of the "
              "bf16 peak), rocBLAS fp32 %.3f ms (%.0f TFLOP/s)" % (what, M, Kd, B * N, t_x3, pr.get("gemm_x3_image", 0),
              pr.get("gemm_x3", 0), gf / pr.get("gemm_x3", 1), 6 * gf / pr.get("gemm_x3", 1) / 2500.0, t_bl, gf / t_bl))


def bench_wgrad():
    """Weight gradients of the trained per-point layers: pn_gemm_x3_wgrad_f32 (two rows images + split-K GEMM +
    fixed-order reduction + bias gradient) against torch.bmm(gy, x^T).sum(0) + gy.sum((0, 2)) on rocBLAS."""
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    for (B, M, Kd, N, what) in [(4, 1024, 256, 10000, "cfg5 mlp1"), (4, 512, 256, 10000, "cfg5 conv1 (local part)"),
                                (4, 256, 512, 10000, "cfg5 conv2"), (4, 256, 256, 10000, "cfg5 seg_prob1"),
                                (4, 128, 256, 10000, "cfg5 seg_prob2"), (32, 1024, 512, 700, "cfg2 conv5"),
                                (32, 1024, 1152, 700, "cfg3 conv5")]:
        gy = torch.randn(B, M, N, device=dev)
        x = torch.randn(B, Kd, N, device=dev)
        t_x3 = timeit(lambda: kernels.gemm_x3_wgrad(gy, x, want_bias=True))
        t_bl = timeit(lambda: (torch.bmm(gy, x.transpose(1, 2)).sum(0), gy.sum((0, 2))))
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(5):
            kernels.gemm_x3_wgrad(gy, x, want_bias=True)
        torch.cuda.synchronize()
        pr = {kn: t / calls for kn, (t, calls) in _lib.prof_results().items()}
        _lib.prof_enable(False)
        gf = 2.0 * M * Kd * B * N / 1e9
        print("%-26s M=%d K=%d points=%d: bf16x3 %.3f ms (images %.3f + GEMM, reduction, bias %.3f = %.0f TFLOP/s "
              "fp32-equivalent), rocBLAS %.3f ms (%.0f TFLOP/s)" % (what, M, Kd, B * N, t_x3, pr.get("gemm_x3_image", 0),
              pr.get("gemm_x3_wgrad", 0), gf / max(pr.get("gemm_x3_wgrad", 1), 1e-9), t_bl, gf / t_bl))


def bench_knnwide():
    """The wide layers of the SplineNets inside a cfg5 step (segments of 2 500 sub-sampled points) and of a
    cfg3 step, bf16 x 3 passes on (PN_KNN_X3 = 2, default) and off (0: the fp32 matrix-core engine)."""
    import os
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    for (B, C, N, k) in [(6, 256, 2500, 10), (12, 256, 2500, 10), (6, 128, 2500, 10), (12, 128, 2500, 10), (32, 256, 700, 10)]:
        x = torch.randn(B, C, N, device=dev) * (0.5 + torch.rand(B, C, 1, device=dev))
        out = {}
        for lvl in ("2", "0"):
            os.environ["PN_KNN_X3"] = lvl
            ms = timeit(lambda: kernels.knn(x, k, "feature"))
            out[lvl] = (ms, kernels.knn(x, k, "feature"))
            _lib.prof_enable(True)
            _lib.prof_reset()
            for _ in range(5):
                kernels.knn(x, k, "feature")
            torch.cuda.synchronize()
            print("    PN_KNN_X3=%s: " % lvl + "  ".join("%s %.3f" % (kn, t / calls) for kn, (t, calls) in sorted(_lib.prof_results().items())))
            _lib.prof_enable(False)
        os.environ.pop("PN_KNN_X3")
        same = bool((out["2"][1] == out["0"][1]).all())
        print("knn B=%d C=%d N=%d k=%d: bf16x3 passes %.3f ms, fp32 engine %.3f ms, same graph: %s"
              % (B, C, N, k, out["2"][0], out["0"][0], same))


def bench_smallk():
    """The SplineNets' graphs (k = 10) inside a cfg5 step (segments of 5 000 sub-sampled points) and a cfg2 / cfg3
    step (32 x 700): the one-pass small-k kernel (PN_KNN_SMALLK=1, default) against the two-pass engine (0)."""
    import os
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    for (B, C, N, k) in [(6, 3, 5000, 10), (6, 64, 5000, 10), (6, 128, 5000, 10), (6, 256, 5000, 10), (3, 256, 5000, 10),
                         (12, 64, 5000, 10), (32, 3, 700, 10), (32, 64, 700, 10), (32, 128, 700, 10), (32, 256, 700, 10)]:
        x = torch.randn(B, C, N, device=dev) * (0.5 + torch.rand(B, C, 1, device=dev))
        out = {}
        for lvl in ("f", "1", "0"):         # f: the one-pass bf16 x 3 form (PN_KNN_FUSED=1) where its plan applies
            os.environ["PN_KNN_SMALLK"] = "0" if lvl == "0" else "1"
            os.environ["PN_KNN_FUSED"] = "1" if lvl == "f" else "0"
            ms = timeit(lambda: kernels.knn(x, k, "feature"))
            out[lvl] = (ms, kernels.knn(x, k, "feature"))
            _lib.prof_enable(True)
            _lib.prof_reset()
            for _ in range(5):
                kernels.knn(x, k, "feature")
            torch.cuda.synchronize()
            print("    %s: " % {"f": "PN_KNN_FUSED=1", "1": "PN_KNN_SMALLK=1", "0": "PN_KNN_SMALLK=0"}[lvl] + "  ".join("%s %.3f" % (kn, t / calls) for kn, (t, calls) in sorted(_lib.prof_results().items())))
            _lib.prof_enable(False)
        os.environ.pop("PN_KNN_SMALLK")
        os.environ.pop("PN_KNN_FUSED")
        same = bool((out["1"][1] == out["0"][1]).all()) and bool((out["1"][1] == out["f"][1]).all())
        gf = B * N * N * (2 * C + 3) / 1e9
        print("knn B=%d C=%d N=%d k=%d: one pass %.3f ms (%.1f TFLOP/s), two-pass engine %.3f ms, one-pass bf16 x 3 form "
              "%.3f ms, same graph: %s" % (B, C, N, k, out["1"][0], gf / out["1"][0], out["0"][0], out["f"][0], same))


def bench_knn64():
    """cfg4's 64-channel layer and the bandwidth selection of cfg5, with the per-kernel timers."""
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    x = torch.randn(4, 64, 10000, device=dev)
    e = torch.nn.functional.normalize(torch.randn(4, 10000, 128, device=dev), dim=2)
    for name, fn in (("knn B=4 C=64 N=10000 k=80", lambda: kernels.knn(x, 80, "feature")),
                     ("dot_select value B=4 C=128 N=10000 k=250", lambda: kernels.dot_select(e, e, 250, True))):
        ms = timeit(fn)
        print("%s: %.3f ms" % (name, ms))
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        for kn, (t, calls) in sorted(_lib.prof_results().items()):
            print("    %-22s %.4f ms" % (kn, t / calls))
        _lib.prof_enable(False)


def bench_chamfer():
    """Both nearest-neighbour kernels of csrc/chamfer.hip (PN_CHAMFER_MFMA=0: scalar, =1: matrix-core pre-filter with
    the exact decision) at the sizes the configs name; results compared bit for bit."""
    import os
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    for (B, Na, Nb) in [(1, 10000, 10000), (32, 1600, 700), (1, 900, 2000), (8, 900, 1200), (16, 930, 1500)]:
        a = torch.rand(B, Na, 3, device=dev)
        b = torch.rand(B, Nb, 3, device=dev)
        res = {}
        for mode in ("0", "1"):
            os.environ["PN_CHAMFER_MFMA"] = mode
            ms = timeit(lambda: kernels.chamfer_nn(a, b))
            res[mode] = kernels.chamfer_nn(a, b)
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(5):
                kernels.chamfer_nn(a, b)
            torch.cuda.synchronize()
            pr = _lib.prof_results()
            _lib.prof_enable(False)
            kms = sum(pr[k][0] for k in ("chamfer_nn", "chamfer_image") if k in pr) / 5.0     # per call, both directions
            # roofs: fp32 VALU issue (8 arithmetic instructions per pair, unfused like the reference's elementwise
            # path; 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 Tinstr/s) for the pairs of the problem; the
            # pre-filter executes one 32 x 32 x 16 bf16 MFMA (32 768 FLOP) per 1 024 pairs: 2.5 PFLOP/s dense
            pairs = 2.0 * B * Na * Nb
            extra = "" if mode == "0" else "; %.2f of the bf16 MFMA peak on the pre-filter's products" % (
                pairs / 1024 * 32768 / (kms * 1e-3) / 2.5e15)
            print("chamfer B=%d %dx%d %s: whole call %.3f ms; kernels %.4f ms = %.2f Tpair/s = %.2f of the 78.6 "
                  "Tinstr/s VALU issue roof at 8 instructions per pair%s" %
                  (B, Na, Nb, "matrix-core pre-filter" if mode == "1" else "scalar kernel         ", ms, kms,
                   pairs / kms / 1e9, 8.0 * pairs / kms / 1e9 / 78.6, extra))
        os.environ.pop("PN_CHAMFER_MFMA", None)
        same = all(torch.equal(x, y) for x, y in zip(res["0"], res["1"]))
        print("    minima and arg-mins of the two kernels bit-identical: %s" % same)


def bench_edge():
    """The API form of get_graph_feature (src/PointNet.py:72-103): cat(x_j - x_i, x_i) materialised,
    and the fused gather-reduce the networks use instead.  Algorithmic bytes per SURVEY 8(d)."""
    from parsenet_codebase_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (B, N, k, C) in [(4, 10000, 80, 64), (4, 10000, 80, 6), (32, 700, 10, 64)]:
        xt = torch.randn(B, N, C, device=dev)
        idx = torch.randint(0, N, (B, N, k), device=dev, dtype=torch.int64)
        ms = timeit(lambda: kernels.edge_feature_fwd(xt, idx))
        nbytes = 4.0 * B * N * C + 8.0 * B * N * k + 4.0 * B * N * k * 2 * C
        print("edge_feature_fwd B=%d N=%d k=%d C=%d: %.3f ms  %.2f GB algorithmic -> %.0f GB/s (%.1f %% of 8 TB/s)"
              % (B, N, k, C, ms, nbytes / 1e9, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
        g = torch.randn(B, N, k, 2 * C, device=dev)
        ms = timeit(lambda: kernels.edge_feature_bwd(g, idx))
        nb = 4.0 * B * N * C + 8.0 * B * N * k + 4.0 * B * N * k * 2 * C
        print("edge_feature_bwd B=%d N=%d k=%d C=%d: %.3f ms  -> %.0f GB/s" % (B, N, k, C, ms, nb / ms / 1e6))
    # real kNN graph (neighbours of a random cloud in feature space) for the fused kernel: the int64 graph of the
    # API and the library's own int32 graph (what the encoders pass between their layers), forward and backward
    for (B, N, k, Cin, Cout) in [(4, 10000, 80, 64, 64), (4, 10000, 80, 64, 128), (6, 5000, 10, 64, 64)]:
        x = torch.randn(B, Cin, N, device=dev)
        PQ = torch.randn(B, N, 2 * Cout, device=dev)
        gamma = torch.ones(Cout, device=dev)
        for int32 in (False, True):
            idx = kernels.knn(x, k, "feature", int32=int32)
            ib = 4.0 if int32 else 8.0
            ms = timeit(lambda: kernels.edgeconv_reduce_fwd(PQ, idx, gamma, 2, True))
            gathered = 4.0 * B * N * k * Cout
            hbm = 4.0 * B * N * 2 * Cout + ib * B * N * k + 9.0 * B * N * Cout
            print("edgeconv_reduce_fwd B=%d N=%d k=%d Cout=%d %s graph: %.3f ms  gathered rows %.0f GB/s (L2), "
                  "HBM-algorithmic %.0f GB/s" % (B, N, k, Cout, "int32" if int32 else "int64", ms, gathered / ms / 1e6,
                                                 hbm / ms / 1e6))
            yext, argk, s1, stats = kernels.edgeconv_reduce_fwd(PQ, idx, gamma, 2, True)
            mean, rstd = kernels.moments(stats, (Cout // 2) * N * k, 1e-5)
            t = torch.randn(B, N, Cout, device=dev)
            c1c2 = torch.randn(B, 2, 2, device=dev) * 1e-3
            _lib.prof_enable(True)
            _lib.prof_reset()
            for _ in range(5):
                kernels.edgeconv_bwd(PQ, idx, t, s1, argk, mean, rstd, c1c2, 2, True, True)
            torch.cuda.synchronize()
            pr = {kn: tt / calls for kn, (tt, calls) in _lib.prof_results().items()}
            _lib.prof_enable(False)
            print("edgeconv_bwd        B=%d N=%d k=%d Cout=%d %s graph: transposed graph %.3f ms, gather %.3f ms"
                  % (B, N, k, Cout, "int32" if int32 else "int64", pr.get("edgeconv_bwd_csr", 0), pr.get("edgeconv_bwd", 0)))


def bench_eval():
    """Evaluation-mode fitting: Evaluation.fitting_loss(eval=True) of 4 shapes x 10 000 points —
    the stage-wise path over all shapes and segments (fitting_eval.py) against the per-segment functions, on
    embeddings with cluster structure (a noisy code of the ground-truth segments, like workloads.warm_paths) and
    the ground-truth types as predicted types; frozen random-init SplineNets; with and without the LS refit."""
    import time
    import numpy as np
    from parsenet_codebase_amd import synthetic
    from parsenet_codebase_amd.encoders import DGCNNControlPoints
    from parsenet_codebase_amd.fitting import Evaluation
    from parsenet_codebase_amd import dp
    dp.limit_host_threads()         # like bench.py / the trainers: the host computes nothing, one intra-op thread
    dev = torch.device("cuda:0")
    B, N = 4, 10000
    pts, nrm, lab, prim = synthetic.make_batch(2000, B, N)
    torch.manual_seed(0)
    ev = Evaluation(closed_path=DGCNNControlPoints(20, num_points=10, mode=1).eval().to(dev),
                    open_path=DGCNNControlPoints(20, num_points=10, mode=0).eval().to(dev))
    g = torch.Generator().manual_seed(5)
    code = torch.nn.functional.normalize(torch.randn(64, 128, generator=g), dim=1)
    emb = torch.nn.functional.normalize(code[torch.from_numpy(lab)] + 0.02 * torch.randn(B, N, 128, generator=g), dim=2).to(dev)
    logp = torch.log_softmax(8.0 * torch.nn.functional.one_hot(torch.from_numpy(prim), 10).float().permute(0, 2, 1), 1).to(dev)
    P, Nr = torch.from_numpy(pts).to(dev), torch.from_numpy(nrm).to(dev)
    kw = dict(quantile=0.025, iterations=10, lamb=0.1)

    def wall(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    for opt in (False, True):
        np.random.seed(1)
        t_b = wall(lambda: ev.fitting_losses_eval(emb, P, Nr, lab, prim, logp, if_optimize=opt, **kw), 3)
        out = ev.fitting_losses_eval(emb, P, Nr, lab, prim, logp, if_optimize=opt, **kw)
        nseg = sum(sum(1 for v in o[1][0].values() if v is not None) for o in out)
        nspl = sum(sum(1 for v in o[1][0].values() if v is not None and "spline" in v[0]) for o in out)
        line = "eval-mode fitting B=4 x 10000 points, %d fitted segments (%d splines), if_optimize=%s: stage-wise %.1f ms " \
               "per batch = %.1f shapes/s" % (nseg, nspl, opt, 1e3 * t_b, B / t_b)
        if not opt:
            ev.batched = False
            np.random.seed(1)
            t_s = wall(lambda: [ev.fitting_loss(emb[b:b + 1], P[b:b + 1], Nr[b:b + 1], lab[b:b + 1], prim[b:b + 1],
                                                logp[b:b + 1], eval=True, **kw) for b in range(B)], 2)
            ev.batched = True
            line += "; per-segment path %.1f ms = %.1f shapes/s" % (1e3 * t_s, B / t_s)
        print(line)


def bench_meanshift():
    import numpy as np
    from parsenet_codebase_amd.mean_shift import MeanShift
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N = 10000
    X = torch.nn.functional.normalize(torch.randn(N, 128, device=dev), dim=1)
    ms = MeanShift()
    b = torch.tensor(0.3, device=dev)

    def fwd_bwd():
        x = X.clone().requires_grad_(True)
        y, _ = ms.mean_shift_(x, b, 10)
        y.sum().backward()
    t = timeit(fwd_bwd, warmup=1, iters=3)
    print("meanshift N=%d 10 it fwd+bwd: %.2f ms  (%.1f TFLOP/s on 9 GEMM units/it)" %
          (N, t, 10 * 9 * 2.0 * N * N * 128 / t / 1e9))
    from parsenet_codebase_amd import _lib
    _lib.prof_enable(True)
    _lib.prof_reset()
    fwd_bwd()
    torch.cuda.synchronize()
    for name, (ms_, calls) in sorted(_lib.prof_results().items()):
        if name.startswith("meanshift"):
            print("  %-20s %.3f ms per launch (%d launches)" % (name, ms_ / calls, calls))
    _lib.prof_enable(False)


def bench_meanshift_batch():
    """Ten iterations forward + backward on a batch of 4 shapes in one launch per pass (how the
    stage-wise fitting path clusters a step) — the workload of the PMC traffic runs."""
    from parsenet_codebase_amd.mean_shift import mean_shift_iterations
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, N = 4, 10000
    X = torch.nn.functional.normalize(torch.randn(B, N, 128, device=dev), dim=2)
    b = torch.full((B,), 0.3, device=dev)

    def fwd_bwd():
        x = X.clone().requires_grad_(True)
        mean_shift_iterations(x, b, 10).sum().backward()
    t = timeit(fwd_bwd, warmup=1, iters=3)
    print("meanshift B=%d N=%d 10 it fwd+bwd: %.2f ms" % (B, N, t))


def bench_meanshift_fwd():
    import parsenet_codebase_amd.mean_shift as MS
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N = 10000
    X = torch.nn.functional.normalize(torch.randn(N, 128, device=dev), dim=1)
    b = torch.tensor(0.3, device=dev)
    for mode in ("f32", "bf16x3", "fp16x2"):
        MS.ARITH = mode
        with torch.no_grad():
            t = timeit(lambda: MS.MeanShift().mean_shift_(X, b, 10), warmup=1, iters=3)
        print("meanshift fwd only, %s: %.3f ms per iteration (%.1f TFLOP/s algorithmic)" %
              (mode, t / 10, 2 * 2.0 * N * N * 128 / (t / 10) / 1e9))


def bench_fitting():
    """Per-shape cost of the clustering + fitting stage when the embedding HAS cluster structure
    (what a trained network yields): embedding = noisy one-hot code of the ground-truth segment."""
    import numpy as np
    from parsenet_codebase_amd import synthetic
    from parsenet_codebase_amd.encoders import DGCNNControlPoints
    from parsenet_codebase_amd.fitting import Evaluation
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N = 10000
    pts, nrm, lab, prim = synthetic.make_batch(0, 2, N)
    ev = Evaluation(closed_path=DGCNNControlPoints(20, num_points=10, mode=1),
                    open_path=DGCNNControlPoints(20, num_points=10, mode=0))
    code = torch.nn.functional.normalize(torch.randn(32, 128), dim=1)
    for b in range(2):
        emb = code[torch.from_numpy(lab[b]).long()] + 0.01 * torch.randn(N, 128)
        emb = torch.nn.functional.normalize(emb, dim=1).to(dev).unsqueeze(0).requires_grad_(True)
        P = torch.from_numpy(pts[b:b + 1]).to(dev)
        Nn = torch.from_numpy(nrm[b:b + 1]).to(dev)
        logp = torch.log_softmax(torch.randn(1, 10, N, device=dev), 1)

        def run():
            res, extra = ev.fitting_loss(emb, P, Nn, lab[b:b + 1], prim[b:b + 1], logp, quantile=0.025,
                                         iterations=10, lamb=0.1)
            res[0].backward()
            return res, extra
        res, extra = run()
        t = timeit(run, warmup=1, iters=3)
        print("fitting shape %d: %d gt segments, %d predicted clusters, s-iou %.3f: %.1f ms fwd+bwd" %
              (b, len(np.unique(lab[b])), len(np.unique(extra[1])), float(res[3]), t))


def bench_fitting_batch():
    """The whole clustering + fitting stage of a step (B = 4 shapes, ground-truth-structured
    embedding): stage-wise path (fitting_batch.py) against the reference-ordered shape-by-shape
    path, with kernel launches and host synchronisations counted by torch's profiler."""
    import numpy as np
    from parsenet_codebase_amd import synthetic
    from parsenet_codebase_amd.encoders import DGCNNControlPoints
    from parsenet_codebase_amd.fitting import Evaluation
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, N = 4, 10000
    pts, nrm, lab, prim = synthetic.make_batch(0, B, N)
    ev = Evaluation(closed_path=DGCNNControlPoints(20, num_points=10, mode=1),
                    open_path=DGCNNControlPoints(20, num_points=10, mode=0))
    code = torch.nn.functional.normalize(torch.randn(32, 128), dim=1)
    emb = torch.stack([torch.nn.functional.normalize(code[torch.from_numpy(lab[b]).long()] + 0.01 * torch.randn(N, 128),
                                                     dim=1) for b in range(B)]).to(dev).requires_grad_(True)
    P = torch.from_numpy(pts).to(dev)
    Nn = torch.from_numpy(nrm).to(dev)
    logp = torch.log_softmax(torch.randn(B, 10, N, device=dev), 1)

    def run(batched):
        ev.batched = batched
        if batched:
            res = ev.fitting_losses(emb, P, Nn, lab, prim, logp, quantile=0.025, iterations=10, lamb=0.1)
        else:
            res = [ev.fitting_loss(emb[b:b + 1], P[b:b + 1], Nn[b:b + 1], lab[b:b + 1], prim[b:b + 1], logp[b:b + 1],
                                   quantile=0.025, iterations=10, lamb=0.1) for b in range(B)]
        sum(r[0][0].sum() for r in res).backward()
        return res
    for batched in (False, True):
        run(batched)
        t = timeit(lambda: run(batched), warmup=1, iters=5)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            res = run(batched)
            torch.cuda.synchronize()
        ev_list = prof.events()
        launches = sum(1 for e in ev_list if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name
                       and "Memset" not in e.name)
        syncs = sum(1 for e in ev_list if e.name in ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize",
                                                     "cudaStreamSynchronize", "cudaDeviceSynchronize",
                                                     "cudaEventSynchronize"))
        d2h = sum(1 for e in ev_list if "Memcpy DtoH" in e.name or "Memcpy DtoH" in str(e.name))
        if batched:     # launches and GPU time per stage of the forward (record_function ranges of fitting_batch.py)
            try:
                import collections
                kev = prof.profiler.kineto_results.events()
                ranges = [(e.name(), e.start_ns(), e.start_ns() + e.duration_ns()) for e in kev
                          if e.name().startswith("fit:")]
                launch_t = {e.correlation_id(): e.start_ns() for e in kev
                            if "LaunchKernel" in e.name() and e.correlation_id() != 0}
                per = collections.OrderedDict((n, [0, 0.0]) for n, _, _ in ranges)
                other = [0, 0.0]
                for e in kev:
                    if str(e.device_type()).endswith("CUDA") and e.correlation_id() in launch_t and "Memcpy" not in e.name() \
                            and "Memset" not in e.name():
                        tl = launch_t[e.correlation_id()]
                        hit = [(b_ - a_, n) for n, a_, b_ in ranges if a_ <= tl <= b_]
                        if hit:
                            n = min(hit)[1]          # innermost enclosing range
                            per[n][0] += 1
                            per[n][1] += e.duration_ns()
                        else:
                            other[0] += 1
                            other[1] += e.duration_ns()
                for n, (c, ns) in list(per.items()) + [("(outside: glue, backward)", other)]:
                    print("      %-28s %4d launches  %7.2f ms GPU" % (n, c, ns / 1e6))
            except Exception as ex:      # profiler internals differ between torch builds
                print("      (per-stage attribution unavailable: %r)" % (ex,))
        nseg = sum(sum(1 for v in r[1][0].values() if v is not None) for r in res)
        print("%-14s %6.1f ms per step of %d shapes (%d fitted segments): %d kernel launches (%.0f per shape), "
              "%d device->host copies, %d host synchronisations (the profiled pass includes the final one)"
              % ("stage-wise" if batched else "shape-by-shape", t, B, nseg, launches, launches / B, d2h, syncs))


if __name__ == "__main__":
    which = sys.argv[1:] or ["knn", "chamfer"]
    for w in which:
        globals()["bench_" + w]()
