"""Mean-shift on the HIP kernels against the torch-CPU oracle: bandwidth, iterates, gradients
through the iterations, and the NMS labels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _clustered(N, C, seed, spread=0.25, d=128):
    g = torch.Generator().manual_seed(seed)
    centers = torch.nn.functional.normalize(torch.randn(C, d, generator=g), dim=1)
    lab = torch.randint(0, C, (N,), generator=g)
    x = centers[lab] + spread * torch.randn(N, d, generator=g) / np.sqrt(d)
    return torch.nn.functional.normalize(x, dim=1), lab


@pytest.mark.parametrize("N,q", [(8000, 0.025), (10000, 0.025), (9000, 0.015)])
def test_bandwidth(gpu, N, q):
    from oracle import ref_torch as R
    from parsenet_codebase_amd.mean_shift import MeanShift
    X, _ = _clustered(N, 7, N)
    np.random.seed(3)
    want = R.MeanShift().compute_bandwidth(X, 10000, q)
    st = np.random.get_state()[1][:4].copy()
    np.random.seed(3)
    got = MeanShift().compute_bandwidth(X.to(gpu), 10000, q)
    assert abs(got.item() - want.item()) / want.item() < 1e-5
    # numpy's RNG must be left in the same state as the reference leaves it
    assert np.array_equal(np.random.get_state()[1][:4], st)


@pytest.mark.parametrize("N,iters", [(500, 3), (2000, 5), (3001, 10)])
def test_iterations_forward_backward(gpu, N, iters):
    from oracle import ref_torch as R
    from parsenet_codebase_amd.mean_shift import MeanShift
    X, _ = _clustered(N, 5, 11 + N)
    b = torch.tensor(0.35)
    w = torch.randn(N, 128)
    xr = X.clone().requires_grad_(True)
    yr, _ = R.MeanShift().mean_shift_(xr, b, iters)
    (yr * w).sum().backward()
    xg = X.to(gpu).requires_grad_(True)
    yg, _ = MeanShift().mean_shift_(xg, b.to(gpu), iters)
    (yg * w.to(gpu)).sum().backward()
    assert _rel(yg, yr) < 1e-5
    assert _rel(xg.grad, xr.grad) < 5e-5


SPLIT_ARITH = ("bf16x3", "fp16x2")


def test_split_products_are_fp32_grade(gpu):
    """The 16-bit matrix-core paths (bf16 x 3, fp16 x 2) against the fp64 evaluation of the
    reference formulas, next to the exact-fp32 path: their error must be of the same size (fp32
    dot products in another summation order, not reduced-precision ones)."""
    from oracle import ref_torch as R
    import parsenet_codebase_amd.mean_shift as MS
    X, _ = _clustered(3001, 5, 7)
    b = 0.35
    y64, _ = R.MeanShift().mean_shift_(X.double(), torch.tensor(b, dtype=torch.float64), 5)
    errs = {}
    saved = MS.ARITH
    try:
        for mode in ("f32",) + SPLIT_ARITH:
            MS.ARITH = mode
            y, _ = MS.MeanShift().mean_shift_(X.to(gpu), torch.tensor(b, device=gpu), 5)
            errs[mode] = float((y.double().cpu() - y64).abs().max())
    finally:
        MS.ARITH = saved
    print(errs)
    assert errs["f32"] < 2e-6, errs
    for mode in SPLIT_ARITH:
        assert errs[mode] < 2e-6 and errs[mode] <= 2.0 * errs["f32"] + 1e-7, errs


@pytest.mark.parametrize("arith", ["fp16x2", "bf16x3", "f32"])
def test_batched_and_tiny_inputs(gpu, arith):
    """(B,N,D) batches with per-item bandwidths equal the per-item runs; N below one tile works."""
    import parsenet_codebase_amd.mean_shift as MS
    saved = MS.ARITH
    MS.ARITH = arith
    try:
        Xa, _ = _clustered(333, 4, 3)
        Xb, _ = _clustered(333, 3, 4)
        Xs = torch.stack([Xa, Xb]).to(gpu).requires_grad_(True)
        bw = torch.tensor([0.3, 0.45], device=gpu)
        w = torch.randn(2, 333, 128, device=gpu)
        y = MS.mean_shift_iterations(Xs, bw, 4)
        (y * w).sum().backward()
        for i, X in enumerate((Xa, Xb)):
            xi = X.to(gpu).requires_grad_(True)
            yi = MS.mean_shift_iterations(xi, bw[i], 4)
            (yi * w[i]).sum().backward()
            assert _rel(y[i], yi) < 1e-6
            assert _rel(Xs.grad[i], xi.grad) < 1e-5
        Xt, _ = _clustered(20, 2, 5)
        from oracle import ref_torch as R
        yr, _ = R.MeanShift().mean_shift_(Xt, torch.tensor(0.4), 3)
        yt = MS.mean_shift_iterations(Xt.to(gpu), 0.4, 3)
        assert _rel(yt, yr) < 1e-5
    finally:
        MS.ARITH = saved


def test_split_backward_is_fp32_grade(gpu):
    """Same for the gradient through 5 iterations (row and column passes)."""
    from oracle import ref_torch as R
    import parsenet_codebase_amd.mean_shift as MS
    X, _ = _clustered(2000, 5, 9)
    b = 0.35
    w = torch.randn(2000, 128, generator=torch.Generator().manual_seed(1))
    x64 = X.double().requires_grad_(True)
    y64, _ = R.MeanShift().mean_shift_(x64, torch.tensor(b, dtype=torch.float64), 5)
    (y64 * w.double()).sum().backward()
    scale = float(x64.grad.abs().max())
    errs = {}
    saved = MS.ARITH
    try:
        for mode in ("f32",) + SPLIT_ARITH:
            MS.ARITH = mode
            xg = X.to(gpu).requires_grad_(True)
            y, _ = MS.MeanShift().mean_shift_(xg, torch.tensor(b, device=gpu), 5)
            (y * w.to(gpu)).sum().backward()
            errs[mode] = float((xg.grad.double().cpu() - x64.grad).abs().max()) / scale
    finally:
        MS.ARITH = saved
    print(errs)
    assert errs["f32"] < 2e-5, errs
    for mode in SPLIT_ARITH:
        assert errs[mode] < 2e-5 and errs[mode] <= 2.0 * errs["f32"] + 1e-6, errs


@pytest.mark.parametrize("gscale", [1e-12, 1.0, 1e9])
@pytest.mark.parametrize("b", [0.05, 1.1])
def test_fp16x2_scaling_rules(gpu, gscale, b):
    """The fp16 x 2 path brings every operand into the fp16 range by powers of two: upstream
    gradients of any magnitude, rows whose gradients differ by many orders of magnitude (one
    global power of two for the column pass), narrow and wide kernels — against fp64."""
    from oracle import ref_torch as R
    import parsenet_codebase_amd.mean_shift as MS
    N = 1500
    X, _ = _clustered(N, 4, 13)
    g = torch.Generator().manual_seed(5)
    w = torch.randn(N, 128, generator=g) * gscale
    w[: N // 2] *= torch.logspace(0, -8, N // 2).unsqueeze(1)    # rows of widely different weight
    w[7] = 0.0                                                    # and an all-zero one
    x64 = X.double().requires_grad_(True)
    y64, _ = R.MeanShift().mean_shift_(x64, torch.tensor(b, dtype=torch.float64), 3)
    (y64 * w.double()).sum().backward()
    saved = MS.ARITH
    try:
        MS.ARITH = "fp16x2"
        xg = X.to(gpu).requires_grad_(True)
        y, _ = MS.MeanShift().mean_shift_(xg, torch.tensor(b, device=gpu), 3)
        (y * w.to(gpu)).sum().backward()
    finally:
        MS.ARITH = saved
    assert float((y.detach().double().cpu() - y64.detach()).abs().max()) < (2e-6 if b > 0.1 else 2e-5)
    scale = float(x64.grad.abs().max())
    err = float((xg.grad.double().cpu() - x64.grad).abs().max()) / scale
    assert torch.isfinite(xg.grad).all()
    assert err < (2e-5 if b > 0.1 else 2e-4), err


def test_full_size_consistency_of_the_arithmetics(gpu):
    """N = 10 000 (the BASELINE size; the CPU oracle would need minutes and 3 x 400 MB per
    iteration): the split-operand kernels and the exact-fp32 kernels are independent
    implementations of the same iteration — they must agree on the iterates and on the gradient,
    and rows must stay unit vectors."""
    import parsenet_codebase_amd.mean_shift as MS
    X, _ = _clustered(10000, 9, 21)
    w = torch.randn(10000, 128, generator=torch.Generator().manual_seed(2))
    res = {}
    saved = MS.ARITH
    try:
        for mode in ("f32",) + SPLIT_ARITH:
            MS.ARITH = mode
            xg = X.to(gpu).requires_grad_(True)
            y = MS.mean_shift_iterations(xg, 0.3, 10)
            (y * w.to(gpu)).sum().backward()
            res[mode] = (y.detach(), xg.grad.detach())
    finally:
        MS.ARITH = saved
    y0, g0 = res["f32"]
    for mode in SPLIT_ARITH:
        y1, g1 = res[mode]
        assert _rel(y1, y0) < 1e-5, mode
        assert _rel(g1, g0) < 1e-4, mode
        assert float((y1.norm(dim=1) - 1).abs().max()) < 1e-5


def _canonical(labels):
    """Relabel by order of first occurrence: equal iff the partitions are equal."""
    labels = np.asarray(labels)
    _, first = np.unique(labels, return_index=True)
    order = labels[np.sort(first)]
    remap = {int(l): i for i, l in enumerate(order)}
    return np.array([remap[int(l)] for l in labels])


def test_dot_select_bit_exact(gpu):
    """The selection engine in dot mode against the C oracle: arg-max indices and K-th values
    are bit-exact (same fma chains, same tie rule)."""
    from oracle import cbind
    from parsenet_codebase_amd import kernels
    X, _ = _clustered(3000, 6, 1)
    C, _ = _clustered(700, 6, 2)
    idx, flags = kernels.dot_select(X.to(gpu).unsqueeze(0), C.to(gpu).unsqueeze(0), 1, False)
    assert int(flags.sum()) == 0
    assert np.array_equal(idx[0, :, 0].cpu().numpy(), cbind.dot_argmax(C.numpy(), X.numpy()))
    Xs = X[:2600]
    val, flags = kernels.dot_select(Xs.to(gpu).unsqueeze(0), Xs.to(gpu).unsqueeze(0), 65, True)
    assert int(flags.sum()) == 0
    assert np.array_equal(val[0].cpu().numpy(), cbind.kth_largest_dot(Xs.numpy(), 65))


def test_nms_labels_bit_exact_on_identical_inputs(gpu):
    """NMS on IDENTICAL shifted points: centre ids and label integers equal the oracle's
    (whose membership step is hooked to the kernels' documented arithmetic)."""
    from oracle import cbind, ref_torch as R
    from parsenet_codebase_amd.mean_shift import MeanShift
    X, _ = _clustered(4000, 9, 5, spread=0.2)
    b = torch.tensor(0.3)
    new_X, _ = R.MeanShift().mean_shift_(X, b, 10)
    R.MEMBERSHIP_IMPL = lambda c, x: torch.from_numpy(cbind.dot_argmax(c.numpy(), x.numpy()))
    try:
        _, ids_r, lab_r = R.MeanShift().nms(new_X, X, b)
    finally:
        R.MEMBERSHIP_IMPL = None
    _, ids_g, lab_g = MeanShift().nms(new_X.to(gpu), X.to(gpu), b.to(gpu))
    assert np.array_equal(ids_g.cpu().numpy(), ids_r.numpy())
    assert np.array_equal(lab_g.cpu().numpy(), lab_r.numpy())


def test_full_mean_shift_partition(gpu):
    """Whole pipeline (bandwidth -> 10 iterations -> NMS) on well separated clusters.  The label
    INTEGERS of mean-shift NMS are decided by arg-min ties between near-coincident shifted
    points (which representative of a mode is kept), so across implementations only the
    partition is well defined: canonically relabelled, the segmentations are identical."""
    from oracle import ref_torch as R
    from parsenet_codebase_amd.mean_shift import MeanShift
    N = 4000
    X, lab = _clustered(N, 9, 5, spread=0.2)
    np.random.seed(0)
    newr, cr, bwr, lr = R.MeanShift().mean_shift(X, 10000, 0.025, 10)
    np.random.seed(0)
    newg, cg, bwg, lg = MeanShift().mean_shift(X.to(gpu), 10000, 0.025, 10)
    assert abs(bwg.item() - bwr.item()) / bwr.item() < 1e-5
    assert _rel(newg, newr) < 1e-4
    assert cg.shape == cr.shape
    assert np.array_equal(_canonical(lg.cpu().numpy()), _canonical(lr.numpy()))
    assert np.array_equal(_canonical(lr.numpy()), _canonical(lab.numpy()))


@pytest.mark.parametrize("N,k", [(3000, 75), (10000, 250), (777, 19)])
def test_kth_dot_on_the_fp16_cores(gpu, N, k):
    """The bandwidth pass on the fp16 matrix cores: the K-th largest dot product of every row within
    6e-7 of the fp64 value (the C oracle's fp32 fma chains are themselves ~5e-7 away from it; the
    two agree to 1.5e-6), no row flagged on tie-free data."""
    from oracle import cbind
    from parsenet_codebase_amd import kernels
    X, _ = _clustered(N, 6, 31 + N)
    Xg = X.to(gpu).unsqueeze(0)
    res = kernels.dot_kth_unit(Xg, kernels.meanshift_h2_split(Xg), N, k)
    assert res is not None
    val, flags = res
    assert int(flags.sum()) == 0
    got = val[0].cpu().numpy()
    assert np.abs(got - cbind.kth_largest_dot(X.numpy(), k)).max() < 1.5e-6
    if N <= 3000:
        x64 = X.numpy().astype(np.float64)
        truth = -np.partition(-(x64 @ x64.T), k - 1, axis=1)[:, k - 1]
        assert np.abs(got - truth).max() < 6e-7


@pytest.mark.parametrize("N,clusters,noise", [(10000, 9, 0.25), (4100, 5, 0.2), (3000, 1, 0.0), (20000, 12, 0.2),
                                              (2049, 3, 0.15)])
def test_block_sparse_iterations_equal_the_dense_ones(gpu, N, clusters, noise):
    """The block-sparse plan (locality order + rigorous tile bounds, csrc/meanshift_x3.h) skips only
    what stays below 1e-9 of the smallest row sum: iterates and gradients equal the dense launches
    to fp32 noise — on a clustered embedding (where most tile pairs are skipped) and on an
    unstructured one (where nothing can be)."""
    import os
    import parsenet_codebase_amd.mean_shift as MS
    torch.cuda.set_device(gpu)
    g = torch.Generator().manual_seed(N)
    if clusters > 1:
        proto = torch.nn.functional.normalize(torch.randn(clusters, 128, generator=g), dim=1)
        lab = torch.randint(0, clusters, (2, N), generator=g)
        X = proto[lab] + noise * torch.randn(2, N, 128, generator=g) / np.sqrt(128)
    else:
        X = torch.randn(2, N, 128, generator=g)
    X = torch.nn.functional.normalize(X, dim=2).to(gpu)
    b = torch.tensor([0.07, 0.11], device=gpu)
    w = torch.randn(2, N, 128, generator=g).to(gpu)
    saved = (MS.ARITH, MS.SPARSE)
    os.environ["PARSENET_MS_STATS"] = "1"
    out = {}
    try:
        MS.ARITH = "bf16x3"
        for sparse in (False, True):
            MS.SPARSE = sparse
            MS.LAST_PLAN_STATS = None
            x = X.clone().requires_grad_(True)
            y = MS.mean_shift_iterations(x, b, 6)
            (y * w).sum().backward()
            out[sparse] = (y.detach(), x.grad.clone(), MS.LAST_PLAN_STATS)
    finally:
        MS.ARITH, MS.SPARSE = saved
        os.environ.pop("PARSENET_MS_STATS", None)
    (yd, gd, _), (ys, gs, stats) = out[False], out[True]
    # the two paths add the same fp32 terms in another order (other slice counts); a rounding of
    # 1e-7 in a dot product is a relative 1e-7 / b^2 = 2e-5 in its kernel value at b = 0.07, so
    # six iterations agree to ~1e-6 — the bar is the parity bar of the iterates, 1e-5
    assert float((yd - ys).abs().max()) < 1e-5
    assert float((gd - gs).abs().max()) < 2e-4 * float(gd.abs().max())
    assert stats is not None and len(stats) == 6
    pair_frac = stats[-1][0]
    if clusters >= 5:
        assert pair_frac < 0.5, stats            # most tile pairs are provably irrelevant
    if clusters == 1:
        assert pair_frac > 0.99, stats           # nothing can be skipped on an unstructured cloud


_SCHEDULE_CHILD = r"""
import sys, numpy as np, torch
import parsenet_codebase_amd.mean_shift as MS
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
g = torch.Generator().manual_seed(7)
N = 4100
proto = torch.nn.functional.normalize(torch.randn(5, 128, generator=g), dim=1)
lab = torch.randint(0, 5, (2, N), generator=g)
X = torch.nn.functional.normalize(proto[lab] + 0.2 * torch.randn(2, N, 128, generator=g) / np.sqrt(128), dim=2).to(dev)
b = torch.tensor([0.07, 0.11], device=dev)
w = torch.randn(2, N, 128, generator=g).to(dev)
MS.ARITH = "bf16x3"
out = {}
for sparse in (False, True):
    MS.SPARSE = sparse
    x = X.clone().requires_grad_(True)
    y = MS.mean_shift_iterations(x, b, 4)
    (y * w).sum().backward()
    out["y%d" % sparse] = y.detach().cpu().numpy()
    out["g%d" % sparse] = x.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_pingpong_schedule_is_bit_identical(gpu, tmp_path):
    """PN_MS_PINGPONG (csrc/meanshift_x3.h) changes WHEN the two waves of a SIMD run their halves
    of a tile, not what a wave computes or in which order it adds: iterates and gradients of the
    dense and the planned launches are equal bit for bit under the old schedule (0), the default
    (1: column pass) and 2 (column and forward pass).  The switch is read once per process: one
    child process per value (children of this process, never an exec of it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("0", "1", "2"):
        f = str(tmp_path / ("pp%s.npz" % mode))
        env = dict(os.environ, PN_MS_PINGPONG=mode, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", _SCHEDULE_CHILD, f], env=env, cwd=root, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = dict(np.load(f))
    for mode in ("1", "2"):
        for k, v in res["0"].items():
            assert np.array_equal(v, res[mode][k]), (mode, k, float(np.abs(v - res[mode][k]).max()))
    assert np.isfinite(res["0"]["g1"]).all() and float(np.abs(res["0"]["g1"]).max()) > 0


@pytest.mark.parametrize("rel_eps", [1e-9, 1e-6])
def test_tile_caps_contain_their_rows_and_the_plan_keeps_every_heavy_pair(gpu, rel_eps):
    """Rigour of the block-sparse plan — at the bound of the plans a dense backward reuses (1e-9,
    mean_shift.PLAN_REL_EPS_DENSE_BWD) and at the training path's forward-only bound (1e-6, PLAN_REL_EPS) —, checked by brute force on a clustered cloud whose tiles
    straddle clusters (N not a multiple of 32, natural order = no locality at all for half of it):
    (1) every row of a tile lies inside one of the tile's two bounding caps;
    (2) for EVERY ROW the kernel values of all the tile pairs the plan drops for its tile add up to
        less than rel_eps of the row's sum — the bound the kernels rely on (round 4: the plan drops by
        the mass of the dropped caps, no longer by "N points at the nearest dropped bound")."""
    from parsenet_codebase_amd import kernels as K
    torch.cuda.set_device(gpu)
    N, B = 4100, 2
    g = torch.Generator().manual_seed(11)
    proto = torch.nn.functional.normalize(torch.randn(6, 128, generator=g), dim=1)
    lab = torch.randint(0, 6, (B, N), generator=g)
    lab[:, :N // 2] = torch.sort(lab[:, :N // 2], dim=1)[0]          # first half grouped, second half mixed
    X = torch.nn.functional.normalize(proto[lab] + 0.15 * torch.randn(B, N, 128, generator=g) / np.sqrt(128), dim=2)
    Xg = X.to(gpu)
    cen, rho, cnt = K.meanshift_x3_tileinfo(Xg)
    T = cen.shape[1]
    assert cnt.shape == rho.shape and float(cnt.sum()) == B * N
    assert cen.shape == (B, T, 2, 128) and rho.shape == (B, T, 2) and T == (N + 63) // 64 * 2
    cen_c, rho_c = cen.cpu().double(), rho.cpu().double()
    Xd = X.double()
    for b in range(B):
        for t in range(T):
            rows = Xd[b, 32 * t:min(N, 32 * t + 32)]
            if rows.shape[0] == 0:
                assert rho_c[b, t, 0] >= 3.14 and rho_c[b, t, 1] < 0           # padding tile: meets everything
                continue
            ang = torch.acos((rows @ cen_c[b, t].T).clamp(-1, 1))               # (rows, 2)
            inside = (ang <= rho_c[b, t][None, :]) & (rho_c[b, t][None, :] >= 0)
            assert bool(inside.any(1).all()), (b, t)
    bsq = torch.tensor([0.08 ** 2, 0.12 ** 2], device=gpu)
    from parsenet_codebase_amd import mean_shift as MSM
    assert {MSM.PLAN_REL_EPS, MSM.PLAN_REL_EPS_DENSE_BWD} == {1e-6, 1e-9}      # the two values this test covers
    plan = K.meanshift_x3_plan((cen, rho), (cen, rho, cnt), bsq, N, rel_eps)
    pairs = plan[:B * T * T].reshape(B, T, T).cpu().bool()
    assert 0.05 < pairs.float().mean() < 0.9
    # a looser bound keeps a subset of the pairs of a tighter one
    tight = K.meanshift_x3_plan((cen, rho), (cen, rho, cnt), bsq, N, 1e-9)[:B * T * T].reshape(B, T, T).cpu().bool()
    assert bool((tight | ~pairs).all())
    # without the counts of the data caps the bounds are more conservative: a superset of the pairs
    loose = K.meanshift_x3_plan((cen, rho), (cen, rho), bsq, N, rel_eps)[:B * T * T].reshape(B, T, T).cpu().bool()
    assert bool((loose | ~pairs).all()) and loose.float().mean() >= pairs.float().mean()
    S = Xd @ Xd.transpose(1, 2)                                                  # (B,N,N) exact enough in fp64
    for b in range(B):
        Kmat = torch.exp((S[b] - 1.0) / float(bsq[b]))
        rsum = Kmat.sum(1)
        pad = T * 32 - N
        Kp = torch.nn.functional.pad(Kmat, (0, pad, 0, pad))
        dropped = ~pairs[b]
        dropped[T - 1:, :] = False                                               # the all-padding q tile has no rows
        assert int(dropped.sum()) > 0
        # mass of the dropped tiles per row: K (T*32, T*32) masked by the (T,T) tile predicate of the row's tile
        col_mass = Kp.reshape(T * 32, T, 32).sum(2)                              # (rows, T): mass per streamed tile
        row_tile = torch.arange(T * 32) // 32
        dropped_mass = (col_mass * dropped[row_tile].double()).sum(1)[:N]
        share = dropped_mass / rsum
        assert float(share.max()) <= rel_eps, float(share.max())


def test_chain_order_is_the_greedy_nearest_neighbour_chain(gpu):
    from parsenet_codebase_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    cen = torch.nn.functional.normalize(torch.randn(3, 128, 128, generator=g), dim=2)
    cen[1, 7] = cen[1, 3]                                                         # a tie: smaller index first
    sim = torch.bmm(cen, cen.transpose(1, 2)).to(gpu)
    rank = K.meanshift_chain_order(sim).cpu()
    simc = sim.cpu()
    for b in range(3):
        used = torch.zeros(128, dtype=torch.bool)
        cur, order = 0, [0]
        used[0] = True
        for _ in range(127):
            s = simc[b, cur].clone()
            s[used] = -float("inf")
            cur = int(torch.nonzero(s == s.max())[0])
            used[cur] = True
            order.append(cur)
        want = torch.empty(128, dtype=torch.long)
        want[torch.tensor(order)] = torch.arange(128)
        assert torch.equal(rank[b], want)


@pytest.mark.parametrize("N,B,dups", [(4100, 2, False), (10000, 2, False), (3000, 1, True)])
def test_pruned_nearest_equals_the_selection_engine(gpu, N, B, dups, monkeypatch):
    """kernels.meanshift_x3_nearest (round 4): the nearest shifted point of every point — the arg-max
    the NMS starts from (src/mean_shift.py:146-149) — evaluated only on the tile pairs whose caps allow a
    maximum.  The pruning is exact and the chains are the engine's (fp32 fma over the channels in order,
    ties to the smaller ORIGINAL index): the indices must equal dot_select's on the unpermuted tensors,
    on a clustered embedding after ten iterations, with duplicated points and duplicated shifted points."""
    import parsenet_codebase_amd.mean_shift as MS
    from parsenet_codebase_amd import kernels as K
    torch.cuda.set_device(gpu)
    g = torch.Generator().manual_seed(N + 1)
    proto = torch.nn.functional.normalize(torch.randn(8, 128, generator=g), dim=1)
    lab = torch.randint(0, 8, (B, N), generator=g)
    X = torch.nn.functional.normalize(proto[lab] + 0.25 * torch.randn(B, N, 128, generator=g) / np.sqrt(128), dim=2)
    if dups:
        X[:, 100:400] = X[:, 1000:1300]            # equal points: equal shifted points, ties everywhere
    X = X.to(gpu)
    bw = torch.full((B,), 0.15, device=gpu)
    monkeypatch.setattr(MS, "SPARSE", True)
    monkeypatch.setattr(MS, "WANT_NEAREST", True)
    with torch.no_grad():
        new_X = MS.mean_shift_iterations(X, bw, 10)
    got = MS.LAST_NEAREST
    assert got is not None and got.shape == (B, N)
    want, flags = K.dot_select(X, new_X, 1, want_value=False)
    if int((flags != 0).sum()) == 0:               # (the engine flags rows with massive ties instead of deciding them)
        assert torch.equal(got, want[:, :, 0])
    else:
        ok = flags == 0
        assert torch.equal(got[ok], want[:, :, 0][ok])
    # and the brute-force definition in fp64 wherever the maximum is unique by a margin
    d = torch.bmm(X.double(), new_X.double().transpose(1, 2))
    top2 = d.topk(2, dim=2)[0]
    clear = (top2[:, :, 0] - top2[:, :, 1]) > 1e-5
    assert torch.equal(got[clear], d.argmax(2)[clear])


@pytest.mark.parametrize("N,B,sparse", [(10000, 2, True), (4100, 3, True), (3001, 2, False), (700, 4, False)])
def test_centre_rows_backward_equals_the_dense_backward(gpu, N, B, sparse, monkeypatch):
    """The training path reads the final iterate at the cluster centres only (src/mean_shift.py:36-43) and a
    step maps every row on its own (src/mean_shift.py:45-79): the backward restricted to those rows
    (mean_shift.centre_rows, csrc/meanshift_rows.hip) returns the gradient of the dense passes — rows picked
    at random, repeats among them (the padded centre lists repeat), planned and dense launches — and is
    bit-reproducible."""
    import parsenet_codebase_amd.mean_shift as MS
    torch.cuda.set_device(gpu)
    monkeypatch.setattr(MS, "ARITH", "bf16x3")
    monkeypatch.setattr(MS, "SPARSE", sparse)
    # (the forward-only state path plans with PLAN_REL_EPS, the autograd path — whose dense backward reuses the
    # plans — with the tighter PLAN_REL_EPS_DENSE_BWD; the bit-equality of the two forward passes asserted below
    # is a statement about the same plans)
    monkeypatch.setattr(MS, "PLAN_REL_EPS_DENSE_BWD", MS.PLAN_REL_EPS)
    g = torch.Generator().manual_seed(N + B)
    proto = torch.nn.functional.normalize(torch.randn(7, 128, generator=g), dim=1)
    lab = torch.randint(0, 7, (B, N), generator=g)
    X = torch.nn.functional.normalize(proto[lab] + 0.25 * torch.randn(B, N, 128, generator=g) / np.sqrt(128), dim=2).to(gpu)
    b = (0.15 + 0.1 * torch.rand(B, generator=g)).to(gpu)
    R = 64
    ids = torch.randint(0, N, (B, R), generator=g)
    ids[:, 40:] = ids[:, :1]                       # the padded tail repeats a row ...
    w = torch.randn(B, R, 128, generator=g)
    w[:, 40:] = 0.0                                # ... with zero gradient, like the masked centre rows
    ids[:, 5] = ids[:, 4]                          # and a genuine repeat with gradient in both
    ids, w = ids.to(gpu), w.to(gpu)
    x0 = X.clone().requires_grad_(True)
    y = MS.mean_shift_iterations(x0, b, 10)
    (torch.gather(y, 1, ids.unsqueeze(2).expand(-1, -1, 128)) * w).sum().backward()
    grads = []
    for _ in range(2):
        x1 = X.clone().requires_grad_(True)
        new_X, state = MS.mean_shift_iterations_state(x1, b, 10)
        assert not new_X.requires_grad
        assert torch.equal(new_X, y.detach())
        c = MS.centre_rows(x1, state, ids)
        assert torch.equal(c, torch.gather(new_X, 1, ids.unsqueeze(2).expand(-1, -1, 128)))
        (c * w).sum().backward()
        grads.append(x1.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert _rel(grads[0], x0.grad) < 2e-5
    # rows without gradient anywhere: exactly zero unless the data row itself was picked or is reached
    assert torch.isfinite(grads[0]).all()


def test_centre_rows_backward_against_the_fp64_reference(gpu, monkeypatch):
    """... and against the reference's own formula evaluated in float64 with autograd (src/mean_shift.py:45-79)."""
    import parsenet_codebase_amd.mean_shift as MS
    torch.cuda.set_device(gpu)
    monkeypatch.setattr(MS, "ARITH", "bf16x3")
    B, N, T = 2, 1500, 10
    g = torch.Generator().manual_seed(5)
    proto = torch.nn.functional.normalize(torch.randn(5, 128, generator=g), dim=1)
    lab = torch.randint(0, 5, (B, N), generator=g)
    X = torch.nn.functional.normalize(proto[lab] + 0.3 * torch.randn(B, N, 128, generator=g) / np.sqrt(128), dim=2)
    b = torch.tensor([0.2, 0.3])
    ids = torch.randint(0, N, (B, 12), generator=g)
    w = torch.randn(B, 12, 128, generator=g)
    xr = X.double().requires_grad_(True)
    outs = []
    for i in range(B):
        new = xr[i].clone()
        for _ in range(T):
            dist = 2.0 - 2.0 * new @ xr[i].t()
            Kmat = torch.exp(torch.clamp(-dist / (b[i].double() ** 2) / 2, max=75, min=-75))
            new = new + ((Kmat @ xr[i]) / Kmat.sum(1, keepdim=True) - new)
            new = new / torch.norm(new, dim=1, p=2, keepdim=True)
        outs.append(new[ids[i]])
    (torch.stack(outs) * w.double()).sum().backward()
    xg = X.to(gpu).requires_grad_(True)
    new_X, state = MS.mean_shift_iterations_state(xg, b.to(gpu), T)
    c = MS.centre_rows(xg, state, ids.to(gpu))
    (c * w.to(gpu)).sum().backward()
    assert _rel(c, torch.stack(outs)) < 1e-5
    assert _rel(xg.grad, xr.grad) < 5e-5


@pytest.mark.parametrize("N", [4100, 10000])
def test_caps_of_the_new_iterate_come_out_of_the_combining_launch(gpu, N):
    """pn_meanshift_x3_iter_fwd_info_f32: iterate, row sums and norms equal the plain planned launch bit for bit,
    and the caps it returns equal pn_meanshift_x3_tileinfo_f32 of that iterate bit for bit (N = 4 100: the last
    tile holds 4 rows and a padding tile follows)."""
    from parsenet_codebase_amd import kernels as K
    torch.cuda.set_device(gpu)
    B = 2
    X = torch.stack([_clustered(N, 7, 3 + b)[0] for b in range(B)]).to(gpu)
    bsq = torch.tensor([0.09 ** 2, 0.13 ** 2], device=gpu)
    x3 = K.meanshift_x3_split(X)
    ws = K.MeanShiftWorkspace(B, N, 128, X.device)
    info = K.meanshift_x3_tileinfo(X)
    plan = K.meanshift_x3_plan(info, info, bsq, N, 1e-6)
    y0, r0, n0 = K.meanshift_x3_iter_fwd(X, x3, bsq, ws, plan)
    y1, r1, n1, got = K.meanshift_x3_iter_fwd(X, x3, bsq, ws, plan, want_info=True)
    assert torch.equal(y0, y1) and torch.equal(r0, r1) and torch.equal(n0, n1)
    want = K.meanshift_x3_tileinfo(y0)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # dense launch (no plan): the caps come from the tile-info kernel behind it
    y2, _, _, got2 = K.meanshift_x3_iter_fwd(X, x3, bsq, ws, None, want_info=True)
    for a, b in zip(got2, K.meanshift_x3_tileinfo(y2)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,K_", [(10000, 128), (4100, 384), (384, 128)])
def test_kmeans_kernels_of_the_locality_order(gpu, N, K_):
    """csrc/kmeans.hip against the tensor expressions they replace: the assignment is the arg-max of the dot
    products (compared where the two best centres of a point are not within rounding of each other, ties to
    the smaller index on exactly duplicated centres), the centres are the normalised sums of their cells
    (an empty cell keeps its old centre); two calls give the same bits."""
    from parsenet_codebase_amd import kernels as K
    torch.cuda.set_device(gpu)
    B = 2
    x = torch.stack([_clustered(N, 9, 20 + b)[0] for b in range(B)]).to(gpu)
    cen = x[:, torch.linspace(0, N - 1, K_).long()].contiguous()
    cen[:, 7] = cen[:, 3]                                        # a duplicated centre: never chosen over its twin
    lab = K.kmeans_assign(x, cen)
    assert lab.dtype == torch.int32 and torch.equal(lab, K.kmeans_assign(x, cen))
    dots = torch.bmm(x.double(), cen.double().transpose(1, 2))
    top2 = dots.topk(2, dim=2)
    clear = (top2.values[..., 0] - top2.values[..., 1]) > 1e-5
    assert float(clear.float().mean()) > 0.9
    assert torch.equal(lab.long()[clear], top2.indices[..., 0][clear])
    assert int((lab == 7).sum()) == 0
    picked = torch.gather(dots, 2, lab.long().unsqueeze(2)).squeeze(2)
    assert float((top2.values[..., 0] - picked).max()) < 1e-5    # elsewhere: a centre within rounding of the best
    new = K.kmeans_centres(x, lab, cen)
    assert torch.equal(new, K.kmeans_centres(x, lab, cen))
    hot = torch.nn.functional.one_hot(lab.long(), K_).double()
    acc = torch.bmm(hot.transpose(1, 2), x.double())
    nrm = acc.norm(dim=2, keepdim=True)
    want = torch.where(nrm > 1e-6, acc / nrm.clamp_min(1e-6), cen.double())
    assert float((new.double() - want).abs().max()) < 2e-6
    empty = (hot.sum(1) == 0)
    assert bool(empty.any()) and torch.equal(new[empty], cen[empty])      # (the duplicated centre's cell, at least)


@pytest.mark.parametrize("N,P,F", [(10000, 128, 384), (10007, 128, 768), (700, 16, 5), (64, 4, 64), (3, 2, 2)])
def test_cell_order_is_the_stable_argsort_of_the_cell_keys(gpu, N, P, F):
    """csrc/kmeans.hip pn_cell_order_i32 = argsort(rank[home[fine]] * F + fine, stable=True), bit for bit; with empty
    cells, a cloud shorter than the 16 waves' ranges, and element counts that are no multiple of 64."""
    from parsenet_codebase_amd import kernels as K
    torch.cuda.set_device(gpu)
    g = torch.Generator().manual_seed(N + F)
    B = 3
    rank = torch.stack([torch.randperm(P, generator=g) for _ in range(B)]).int().to(gpu)
    home = torch.randint(0, P, (B, F), generator=g).int().to(gpu)
    fine = torch.randint(0, F, (B, N), generator=g).int()
    if F > 4:
        fine[fine == 3] = 4                                      # an empty cell
        fine[1, : N // 2] = 1                                    # one crowded cell: every lane of a group in it
    fine = fine.to(gpu)
    got = K.cell_order(rank, home, fine)
    key = torch.gather(rank.long(), 1, torch.gather(home.long(), 1, fine.long())) * F + fine.long()
    assert got.dtype == torch.int64 and torch.equal(got, torch.argsort(key, dim=1, stable=True))
    with pytest.raises(Exception):
        K.cell_order(rank, torch.zeros((B, 769), dtype=torch.int32, device=gpu), fine)


def test_locality_order_with_and_without_the_order_kernel(gpu):
    """mean_shift.locality_order: the one-launch counting sort gives the permutation the tensor-library path gives."""
    from parsenet_codebase_amd import mean_shift as M
    torch.cuda.set_device(gpu)
    x = torch.stack([_clustered(10000, 9, 30 + b)[0] for b in range(2)]).to(gpu)
    old = M.ORDER_KERNEL
    try:
        M.ORDER_KERNEL = True
        a = M.locality_order(x)
        M.ORDER_KERNEL = False
        b = M.locality_order(x)
    finally:
        M.ORDER_KERNEL = old
    assert a.dtype == b.dtype and torch.equal(a, b)
    assert torch.equal(a.sort(1).values, torch.arange(10000, device=gpu).expand(2, -1))


@pytest.mark.parametrize("d,kernel_type", [(64, "gaussian"), (64, "epa"), (128, "epa")])
def test_other_embedding_widths_and_the_epanechnikov_kernel_against_the_oracle(gpu, d, kernel_type):
    """src/mean_shift.py:45-79 with an embedding that is not 128 wide and / or the Epanechnikov kernel
    (:64-68) — no config uses them; the product runs them as materialised N x N tensor expressions
    (mean_shift.py: the one place it builds such a matrix).  Iterates and the gradient through the
    iterations against the oracle; whole clustering (bandwidth, partition) at width 64."""
    from oracle import ref_torch as R
    from parsenet_codebase_amd.mean_shift import MeanShift
    N = 1500
    X, lab = _clustered(N, 6, 21 + d, spread=0.2, d=d)
    b = torch.tensor(0.5)
    w = torch.randn(N, d, generator=torch.Generator().manual_seed(4))
    xr = X.clone().requires_grad_(True)
    yr, _ = R.MeanShift().mean_shift_(xr, b, 5, kernel_type=kernel_type)
    (yr * w).sum().backward()
    xg = X.to(gpu).requires_grad_(True)
    yg, _ = MeanShift().mean_shift_(xg, b.to(gpu), 5, kernel_type=kernel_type)
    (yg * w.to(gpu)).sum().backward()
    assert _rel(yg, yr) < 1e-5
    assert _rel(xg.grad, xr.grad) < 5e-5
    if kernel_type == "gaussian":
        np.random.seed(0)
        # (K = int(0.015 * 10000) = 150 stays inside a cluster of ~250 points)
        newr, cr, bwr, lr = R.MeanShift().mean_shift(X, 10000, 0.015, 10)
        np.random.seed(0)
        newg, cg, bwg, lg = MeanShift().mean_shift(X.to(gpu), 10000, 0.015, 10)
        assert abs(bwg.item() - bwr.item()) / bwr.item() < 1e-5
        assert _rel(newg, newr) < 1e-4 and cg.shape == cr.shape
        assert np.array_equal(_canonical(lg.cpu().numpy()), _canonical(lr.numpy()))
        assert np.array_equal(_canonical(lr.numpy()), _canonical(lab.numpy()))
