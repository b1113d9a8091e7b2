"""Differentiable mean-shift clustering on the unit hypersphere (src/mean_shift.py) on the HIP
kernels: bandwidth quantile and nearest-centre assignment through the matrix-core selection
engine, iterations through the fused flash-style kernels with a recompute backward.  Same class,
method names and return values as the reference."""
import os

import numpy as np
import torch

from . import kernels as K
from ._lib import h2d, require_cuda


# Arithmetic of the two GEMMs per tile.  Default "bf16x3": every fp32 operand is split WITHOUT
# error into three bf16 pieces (3 x 8 = 24 significand bits) and the six piece products above
# 2^-25 relative size are accumulated in fp32 on the bf16 matrix cores — an fp32 dot product in
# another summation order, i.e. the reference's precision.  "f32": v_mfma_f32_32x32x2_f32, exact
# fma chains.  "fp16x2" (opt-in): scaled 2-way fp16 split, three piece products, 22 significand
# bits per operand — NARROWER than fp32; faster, never the default, and bench.py labels it.
# PARSENET_MS_ARITH overrides.
ARITH = os.environ.get("PARSENET_MS_ARITH", "bf16x3")
# Block-sparse iterations (bf16x3 path, N >= 2048): points are put in a locality order, and tile
# pairs whose kernel values are rigorously below PLAN_REL_EPS of the smallest row sum are skipped
# (csrc/meanshift_x3.h, "block-sparse plan").  How much that skips is DATA DEPENDENT: an embedding
# early in training (small bandwidth, everything in one region of the sphere) keeps ~25 % of the
# pairs, a triplet embedding after 2000 pre-training steps on held-out shapes 70-90 %, and above a
# bandwidth of ~0.26 NOTHING can be skipped (b^2 ln(N / 1e-9) exceeds the range of a dot product).
# The duration of a planned launch follows the share of tile pairs it executes (measured: 0.73 of
# the dense launch at 0.71 of the pairs, 0.83 at 0.85; matrix cores at 0.59-0.61 of peak on the
# executed pairs, like the dense kernel) plus ~1.2 ms of plan kernels per call of ten iterations:
# it pays below ~0.93 of the pairs.  PARSENET_MS_SPARSE: "1" always plan, "0" always dense, "auto"
# (default): plan AUTO_SAMPLES calls, look at the share of pairs their plans kept (the number rides
# in the fitting stage's cluster-id download: no extra synchronisation), and launch dense for the
# next AUTO_DENSE_STEPS calls of that problem size when their mean was above AUTO_DENSE_ABOVE.
_env_sparse = os.environ.get("PARSENET_MS_SPARSE", "auto")
SPARSE = True if _env_sparse == "1" else False if _env_sparse == "0" else "auto"
AUTO_DENSE_ABOVE = 0.93
AUTO_DENSE_STEPS = 48
AUTO_SAMPLES = 4
_AUTO = {}                  # (B, N) -> {calls left before the next planned (probing) calls, shares of the last ones}
CALLS = {"planned": 0, "dense": 0}   # calls of the bf16x3 iterations by launch kind (bench.py reports them)
AUTO_STAT = None            # device scalar of the most recent planned call in auto mode (see fitting_batch)
# what a plan may drop: tile pairs whose N terms together stay below this share of the SMALLEST row
# sum of the q tile (csrc/meanshift_x3.h).  The bound is rigorous; its size is a choice.  Round 4:
# 1e-6 instead of 1e-9.  A row sum is an fp32 sum of N = 10 000 terms: its own rounding error is
# ~sqrt(N) 2^-24 = 6e-6 relative whatever the order (the reference's GEMM included), so a dropped
# share of 1e-6 is below the arithmetic's noise — measured on the benchmark's held-out embedding
# (tools/plan_eps_probe.py, profiles/r04_plan_eps_probe.txt): the ten-iteration result differs
# from the dense launches' by 7e-6 ... 7e-5 (summation order: the locality permutation) and does
# not move between rel_eps = 1e-9 and 1e-4, while the plans keep 0.81 / 0.70 / 0.66 of the tile
# pairs at 1e-9 / 1e-6 / 1e-5 (with the round-2/3 criterion; with the mass / row-sum criterion of
# round 4, csrc/meanshift_x3.h: 0.51 at 1e-6).  PARSENET_MS_REL_EPS overrides (1e-9: rounds 2-3).
PLAN_REL_EPS = float(os.environ.get("PARSENET_MS_REL_EPS", "1e-6"))
# Plans that a DENSE backward reuses (the autograd path of mean_shift_iterations: the public API,
# guard_mean_shift's fallback, PARSENET_MS_ROWS_BWD=0): the backward drops the same pairs, and the share
# of a gradient row it drops is bounded by rel_eps / b^2, not rel_eps (csrc/meanshift_x3.h) — 1e-4 at
# b = 0.1 with the forward's 1e-6.  Those plans keep the round-2/3 bound of 1e-9 (1e-7 at b = 0.1); the
# training path's forward-only state (mean_shift_iterations_state) has an exact, dense row backward
# (csrc/meanshift_rows.hip) and plans with PLAN_REL_EPS.
PLAN_REL_EPS_DENSE_BWD = float(os.environ.get("PARSENET_MS_REL_EPS_BWD", str(min(PLAN_REL_EPS, 1e-9))))
SPARSE_MIN_N = 2048
SPARSE_MAX_N = 32768        # the plan's threshold search holds one row of <= 2048 cap bounds in registers
LAST_PLAN_STATS = None      # diagnostics of the most recent call (only filled when PARSENET_MS_STATS=1)
# The nearest shifted point of every point (the first step of MeanShift.nms) as a by-product of a planned
# call: set WANT_NEAREST before mean_shift_iterations, read LAST_NEAREST ((B,N) int64 or None) after it.
# The iterations hold points and final iterate in the locality order the exact pruning needs.
WANT_NEAREST = False
LAST_NEAREST = None


def use_sparse(B, N):
    """Whether the next call of this problem size plans its launches (see SPARSE)."""
    if SPARSE != "auto":
        return bool(SPARSE)
    st = _AUTO.get((B, N))
    if st is not None and st["left"] > 0:
        st["left"] -= 1
        return False
    return True


def auto_report(B, N, share):
    """The share of tile pairs the plans of the last planned call kept.  The shapes of a batch differ
    (0.70 ... 0.95 inside one pool of 16): the decision to launch dense is taken on the mean of
    AUTO_SAMPLES consecutive planned calls, not on one."""
    if SPARSE != "auto":
        return
    st = _AUTO.setdefault((B, N), {"left": 0, "hist": []})
    st["hist"] = (st["hist"] + [float(share)])[-AUTO_SAMPLES:]
    if len(st["hist"]) >= AUTO_SAMPLES and sum(st["hist"]) / len(st["hist"]) > AUTO_DENSE_ABOVE:
        st["left"], st["hist"] = AUTO_DENSE_STEPS, []


KMEANS_KERNELS = os.environ.get("PARSENET_MS_KMEANS_KERNELS", "1") != "0"   # (0: the tensor-library form, A/B)
FINE_CELLS = int(os.environ.get("PARSENET_MS_FINE", "384"))    # second-level cells of the locality order (0: off)


ORDER_KERNEL = os.environ.get("PARSENET_MS_ORDER_KERNEL", "1") != "0"       # (0: the tensor library's argsort, A/B)
_SEEDS = {}


def _seed_rows(N, K_, device):
    """Evenly spaced row indices, the seeds of a k-means (cached: two launches per call otherwise)."""
    hit = _SEEDS.get((N, K_, device))
    if hit is None:
        hit = _SEEDS[(N, K_, device)] = torch.linspace(0, N - 1, K_, device=device).long()
    return hit


def locality_order(x, lloyd=2):
    """A permutation (B,N) that puts points of the same region of the sphere next to each other:
    128 cells (spherical k-means: evenly spaced rows of x as seeds, ``lloyd`` refinement steps)
    laid out along a greedy nearest-neighbour chain of their centres; inside a cell the points are
    grouped by a second, finer k-means (FINE_CELLS cells of ~26 points, each filed under the coarse cell
    of its centre) instead of staying in their original order, so that a 32-point tile holds one or two
    fine cells rather than a random third of a coarse one (round 4: the plans keep 0.465 instead of
    0.504 of the tile pairs on the benchmark's embedding, tools/order_probe.py).  Any permutation is
    valid — mean-shift is permutation-equivariant; a good one makes the 32-point tiles tight and the
    tiles of a resident block alike, which is what lets the plan skip tile pairs."""
    B, N, D = x.shape
    P = 128

    fused = D == 128 and KMEANS_KERNELS      # csrc/kmeans.hip: two launches per Lloyd step instead of a dozen

    def assign(pts, cen):
        if fused:
            return K.kmeans_assign(pts, cen)                     # int32 (converted once, below)
        return torch.bmm(pts, cen.transpose(1, 2)).argmax(2)

    def centres(pts, lab, K_, old):
        if fused:
            return K.kmeans_centres(pts, lab, old)
        # one-hot GEMM instead of index_add_: atomics would make the order — and with it the
        # summation order of every later launch — vary from run to run
        hot = torch.nn.functional.one_hot(lab, K_).to(pts.dtype)             # (B,n,K)
        acc = torch.bmm(hot.transpose(1, 2), pts)
        nrm = acc.norm(dim=2, keepdim=True)
        return torch.where(nrm > 1e-6, acc / nrm.clamp_min(1e-6), old)      # empty cell: keep its seed

    def kmeans(K_):
        cen = x[:, _seed_rows(N, K_, x.device)].contiguous()
        lab = assign(x, cen)
        for _ in range(lloyd):
            cen = centres(x, lab, K_, cen)
            lab = assign(x, cen)
        return cen, lab
    cen, coarse = kmeans(P)
    two_level = not (FINE_CELLS <= P or N < 8 * FINE_CELLS)
    one_launch = fused and two_level and ORDER_KERNEL and FINE_CELLS <= K.CELL_ORDER_MAX_CELLS
    rank = K.meanshift_chain_order(torch.bmm(cen, cen.transpose(1, 2)), as_long=not one_launch)       # (B,P)
    if not two_level:
        return torch.argsort(torch.gather(rank, 1, coarse.long()), dim=1, stable=True)
    cen2, fine = kmeans(FINE_CELLS)
    home = assign(cen2, cen)                                                   # (B,FINE): coarse cell of a fine centre
    if one_launch:
        # every fine cell has ONE key: the stable argsort is a counting sort over the cells (csrc/kmeans.hip), one
        # launch where conversions, gathers, the key arithmetic and the tensor library's sort took 23
        return K.cell_order(rank, home, fine)
    fine, home = fine.long(), home.long()
    key = torch.gather(rank, 1, torch.gather(home, 1, fine)).long() * FINE_CELLS + fine
    return torch.argsort(key, dim=1, stable=True)


_SPLIT = {"fp16x2": (K.meanshift_h2_split, K.meanshift_h2_iter_fwd, K.meanshift_h2_iter_bwd),
          "bf16x3": (K.meanshift_x3_split, K.meanshift_x3_iter_fwd, K.meanshift_x3_iter_bwd)}


def _run_iterations(X, bsq, iterations, stacked=False, rel_eps=None):
    """The forward pass of the iterations (no autograd; ``rel_eps``: the bound of the plans, default
    PLAN_REL_EPS).  Returns a dict: the operands in the order the
    kernels ran on (``x``: the data, locality-ordered when the launches are planned; ``perm`` / ``inv``),
    the iterates / row sums / norms of every step and the plans.  ``stacked``: the iterates, row sums and
    norms of all steps live in ONE buffer each (``iterates_all`` (T+1,B,N,D), ``rsums_all`` / ``norms_all``
    (T,B,N)) — what the row-restricted backward gathers its rows from in three launches."""
    x = X.contiguous()
    B, N, D = x.shape
    if ARITH not in _SPLIT and ARITH != "f32":
        raise ValueError("PARSENET_MS_ARITH must be fp16x2, bf16x3 or f32, not %r" % ARITH)
    kern = _SPLIT.get(ARITH) if iterations > 0 else None
    sparse = kern is not None and ARITH == "bf16x3" and SPARSE_MIN_N <= N <= SPARSE_MAX_N and use_sparse(B, N)
    if kern is not None and ARITH == "bf16x3":
        CALLS["planned" if sparse else "dense"] += 1
    perm = inv = None
    if sparse:   # everything below runs on the locality-ordered points; undone on the way out
        perm = locality_order(x, int(os.environ.get("PARSENET_MS_LLOYD", "2")))
        inv = torch.empty_like(perm).scatter_(1, perm, torch.arange(N, device=x.device).expand(B, N))
        x = torch.gather(x, 1, perm.unsqueeze(2).expand(-1, -1, D))
    x3 = kern[0](x) if kern is not None else None
    # streamed copy of X: pre-split tile images (16-bit pieces) or channel-first fp32 (exact path)
    xt = x3 if x3 is not None else K.meanshift_pack(x)
    ws = K.MeanShiftWorkspace(B, N, D, x.device)
    direct = stacked and ARITH == "bf16x3" and x3 is not None       # the bf16 x 3 kernels write into the buffers
    it_all = rs_all = nr_all = None
    if direct:
        it_all = torch.empty((iterations + 1, B, N, D), dtype=torch.float32, device=x.device)
        rs_all = torch.empty((iterations, B, N), dtype=torch.float32, device=x.device)
        nr_all = torch.empty((iterations, B, N), dtype=torch.float32, device=x.device)
        it_all[0].copy_(x)
        x = it_all[0]
    iterates, rsums, norms, plans = [x], [], [], []
    x_info = K.meanshift_x3_tileinfo(x) if sparse else None
    # (the plans of all iterations in one buffer: the auto mode's statistic is then one reduction, not one per plan)
    plans_buf, plan_slots, plan_core = K.meanshift_x3_plan_buffer(B, N, iterations, x.device) \
        if sparse and iterations > 0 else (None, None, 0)
    q = x
    q_info = x_info          # (the first iterate IS the data: its caps are x_info)
    for it in range(iterations):
        out = (it_all[it + 1], rs_all[it], nr_all[it]) if direct else None
        if sparse:
            plan = K.meanshift_x3_plan(q_info, x_info, bsq, N, PLAN_REL_EPS if rel_eps is None else rel_eps,
                                       out=plan_slots[it])
            plans.append(plan)
            # (the caps of the new iterate — the next plan's, and the nearest-point search's at the end — come
            # out of the launch that combines the partial results)
            q, r, n, q_info = K.meanshift_x3_iter_fwd(q, x3, bsq, ws, plan, out=out, want_info=True)
        elif x3 is not None and ARITH == "bf16x3":
            q, r, n = K.meanshift_x3_iter_fwd(q, x3, bsq, ws, None, out=out)
        elif x3 is not None:
            q, r, n = kern[1](q, x3, bsq, ws)
        else:
            q, r, n = K.meanshift_iter_fwd(q, x, xt, bsq, ws)
        iterates.append(q)
        rsums.append(r)
        norms.append(n)
    global LAST_PLAN_STATS, AUTO_STAT, LAST_NEAREST
    LAST_NEAREST = None
    if sparse and WANT_NEAREST and iterations > 0:
        LAST_NEAREST = K.meanshift_x3_nearest(x, q, x_info, q_info, perm)
    if sparse and os.environ.get("PARSENET_MS_STATS") == "1":
        LAST_PLAN_STATS = [K.meanshift_x3_plan_stats(p, B, N) for p in plans]
    if sparse and SPARSE == "auto" and plans:
        AUTO_STAT = K.meanshift_x3_plan_visited((plans_buf, len(plans), plan_core), B, N)
    if stacked and not direct:
        it_all = torch.stack(iterates)
        rs_all = torch.stack(rsums) if rsums else torch.empty((0, B, N), device=x.device)
        nr_all = torch.stack(norms) if norms else torch.empty((0, B, N), device=x.device)
    return {"x": x, "xt": xt, "x3": x3, "kern": kern, "sparse": sparse, "perm": perm, "inv": inv, "q": q,
            "iterates": iterates, "rsums": rsums, "norms": norms, "plans": plans,
            "iterates_all": it_all, "rsums_all": rs_all, "norms_all": nr_all}


class _MeanShiftIterations(torch.autograd.Function):
    """X (B,N,D) unit rows, bsq (B) squared bandwidths -> iterate after ``iterations`` steps.
    Saves only the iterates, row sums and norms (O(T N D)); the backward recomputes the kernel."""

    @staticmethod
    def forward(ctx, X, bsq, iterations):
        D = X.shape[2]
        st = _run_iterations(X, bsq, iterations, rel_eps=PLAN_REL_EPS_DENSE_BWD)   # (its backward reuses the plans)
        sparse, q = st["sparse"], st["q"]
        ctx.iterations = iterations
        ctx.x3 = st["x3"]
        ctx.kern = st["kern"]
        ctx.sparse = sparse
        ctx.save_for_backward(st["xt"], bsq, *st["iterates"], *st["rsums"], *st["norms"], *st["plans"],
                              *([st["perm"], st["inv"]] if sparse else []))
        if iterations == 0:
            return st["x"].clone()
        return torch.gather(q, 1, st["inv"].unsqueeze(2).expand(-1, -1, D)) if sparse else q

    @staticmethod
    def backward(ctx, gy):
        T = ctx.iterations
        saved = ctx.saved_tensors
        xt, bsq = saved[0], saved[1]
        iterates = saved[2:3 + T]
        rsums = saved[3 + T:3 + 2 * T]
        norms = saved[3 + 2 * T:3 + 3 * T]
        plans = saved[3 + 3 * T:3 + 4 * T] if ctx.sparse else None
        x = iterates[0]
        B, N, D = x.shape
        ws = K.MeanShiftWorkspace(B, N, D, x.device, backward=True, exact_f32=ctx.x3 is None)
        gX = torch.zeros_like(x)
        g = gy.contiguous()
        if ctx.sparse:
            perm, inv = saved[-2], saved[-1]
            g = torch.gather(g, 1, perm.unsqueeze(2).expand(-1, -1, D))
        for it in reversed(range(T)):
            if ctx.sparse:
                g = K.meanshift_x3_iter_bwd(g, iterates[it + 1], iterates[it], x, ctx.x3, rsums[it], norms[it], bsq,
                                            ws, gX, plans[it])
            elif ctx.x3 is not None:
                g = ctx.kern[2](g, iterates[it + 1], iterates[it], x, ctx.x3, rsums[it], norms[it], bsq, ws, gX)
            else:
                g = K.meanshift_iter_bwd(g, iterates[it + 1], iterates[it], x, xt, rsums[it], norms[it], bsq,
                                         ws, gX)
        gX += g  # the first iterate is X itself
        if ctx.sparse:
            gX = torch.gather(gX, 1, inv.unsqueeze(2).expand(-1, -1, D))
        return gX, None, None


def mean_shift_iterations(X, b, iterations):
    """X (N,D) or (B,N,D); b scalar / 0-dim tensor / (B,) tensor of bandwidths."""
    require_cuda(X)
    squeeze = X.dim() == 2
    Xb = X.unsqueeze(0) if squeeze else X
    B = Xb.shape[0]
    bt = torch.as_tensor(b, dtype=torch.float32, device=X.device).reshape(-1)
    if bt.numel() == 1:
        bt = bt.expand(B)
    bsq = (bt.detach() ** 2).contiguous()
    out = _MeanShiftIterations.apply(Xb, bsq, int(iterations))
    return out[0] if squeeze else out


class MeanShiftState:
    """What a forward pass of the iterations leaves behind for the row-restricted backward
    (``mean_shift_iterations_state`` / ``centre_rows``)."""
    __slots__ = ("x", "bsq", "iterates", "rsums", "norms", "inv", "iterations", "new_X")


def mean_shift_iterations_state(X, b, iterations):
    """The iterations WITHOUT an autograd graph: returns (new_X (B,N,D) detached, state).  The training
    path reads the final iterate at the cluster centres only; ``centre_rows(X, state, ids)`` returns those
    rows WITH the gradient path back to X."""
    require_cuda(X)
    if X.dim() != 3:
        raise ValueError("mean_shift_iterations_state expects (B,N,D)")
    B, N, D = X.shape
    bt = torch.as_tensor(b, dtype=torch.float32, device=X.device).reshape(-1)
    if bt.numel() == 1:
        bt = bt.expand(B)
    bsq = (bt.detach() ** 2).contiguous()
    with torch.no_grad():
        st = _run_iterations(X.detach(), bsq, int(iterations), stacked=True)
        q = st["q"]
        new_X = torch.gather(q, 1, st["inv"].unsqueeze(2).expand(-1, -1, D)) if st["sparse"] else q
        if iterations == 0:
            new_X = new_X.clone()
    state = MeanShiftState()
    state.x, state.bsq, state.inv, state.iterations, state.new_X = st["x"], bsq, st["inv"], int(iterations), new_X
    state.iterates, state.rsums, state.norms = st["iterates_all"], st["rsums_all"], st["norms_all"]
    return new_X, state


class _CentreRows(torch.autograd.Function):
    """rows ``ids`` (B,R <= 64) of the final iterate, differentiable w.r.t. the data X.

    A step of src/mean_shift.py:45-79 maps row i of the iterate to a function of that row and of X alone,
    so a gradient that enters the final iterate at R rows (centres = new_X[indices],
    src/mean_shift.py:36-43) is zero in every other row of every earlier iterate: the backward runs the R
    rows against the N data points, step by step (csrc/meanshift_rows.hip), instead of N x N."""

    @staticmethod
    def forward(ctx, X, state, ids):
        B, N, D = state.x.shape
        T = state.iterations
        ids = ids.long()
        rows = torch.gather(state.inv, 1, ids) if state.inv is not None else ids      # positions in the kernels' order
        R = rows.shape[1]
        Qc = torch.gather(state.iterates, 2, rows.view(1, B, R, 1).expand(T + 1, B, R, D))          # (T+1,B,R,D)
        rc = torch.gather(state.rsums, 2, rows.view(1, B, R).expand(T, B, R))
        nc = torch.gather(state.norms, 2, rows.view(1, B, R).expand(T, B, R))
        ctx.T = T
        ctx.has_inv = state.inv is not None
        ctx.save_for_backward(Qc, rc, nc, rows, state.x, state.bsq, *([state.inv] if state.inv is not None else []))
        return Qc[T].clone()

    @staticmethod
    def backward(ctx, g):
        Qc, rc, nc, rows, x, bsq = ctx.saved_tensors[:6]
        B, N, D = x.shape
        gX = torch.zeros_like(x)
        g = g.contiguous()
        ws = K.meanshift_rows_workspace(B, N, x.device)
        for t in reversed(range(ctx.T)):
            g = K.meanshift_rows_bwd(g, Qc[t + 1], Qc[t], rc[t], nc[t], x, bsq, gX, ws)
        K.meanshift_rows_scatter_add(gX, rows, g)            # the first iterate is X itself
        if ctx.has_inv:
            inv = ctx.saved_tensors[6]
            gX = torch.gather(gX, 1, inv.unsqueeze(2).expand(-1, -1, D))
        return gX, None, None


def centre_rows(X, state, ids):
    """Rows ``ids`` (B,R) of the final iterate of ``mean_shift_iterations_state(X, ...)`` with the gradient
    path to X (R <= 64; ids may repeat)."""
    if ids.shape[1] > 64:
        raise ValueError("centre_rows: at most 64 rows per batch item, got %d" % ids.shape[1])
    return _CentreRows.apply(X, state, ids)


def _first_argmax(vals, dim):
    """argmax with the smallest index among ties (torch's GPU reduction leaves it unspecified)."""
    m = vals.max(dim, keepdim=True)[0]
    n = vals.shape[dim]
    shape = [1] * vals.dim()
    shape[dim] = n
    ar = torch.arange(n, device=vals.device).view(shape)
    return torch.where(vals == m, ar, torch.full_like(ar, n)).min(dim)[0]


class MeanShift:
    def __init__(self):
        pass

    # -- src/mean_shift.py:19-43 ---------------------------------------------------------
    def mean_shift(self, X, num_samples, quantile, iterations, kernel_type="gaussian", bw=None, nms=True):
        """X (N,d) unit rows.  Returns (new_X, center, bw, labels), or (new_X, bw) if not nms."""
        if bw is None:
            with torch.no_grad():
                bw = self.compute_bandwidth(X, num_samples, quantile)
                bw = torch.clamp(bw, min=0.003)   # avoid numerical issues
        new_X, _ = self.mean_shift_(X, b=bw, iterations=iterations, kernel_type=kernel_type)
        if not nms:
            return new_X, bw
        with torch.no_grad():
            _, indices, new_labels = self.nms(new_X, X, b=bw)
        center = new_X[indices]
        return new_X, center, bw, new_labels

    # -- the same in two stages, for callers that overlap shapes on two streams -------------
    def shift_async(self, X, num_samples, quantile, iterations):
        """Stage 1 of ``mean_shift`` (bandwidth + iterations) WITHOUT any host synchronisation, so
        that it can be queued on a side stream while the host drives another shape's fitting
        stage.  Returns (new_X, bw, flag); flag (0-dim tensor) > 0 means the bandwidth selection met
        massive ties and the caller must use ``mean_shift`` instead (checked in ``finish``)."""
        if num_samples < X.shape[0]:
            raise ValueError("shift_async needs num_samples >= N: the bandwidth then does not depend on the "
                             "shuffle, which is what lets ``finish`` draw it at the reference's place in "
                             "numpy's RNG stream")
        with torch.no_grad():
            bw, flag = self.compute_bandwidth(X, num_samples, quantile, defer_flags=True, defer_shuffle=True)
            bw = torch.clamp(bw, min=0.003)
        new_X, _ = self.mean_shift_(X, b=bw, iterations=iterations)
        return new_X, bw, flag

    def finish(self, X, new_X, bw, flag=None):
        """Stage 2: non-maximum suppression.  Returns (new_X, center, bw, labels) like
        ``mean_shift``, or None if stage 1 has to be redone on the synchronous path.  The shuffle
        that ``mean_shift`` owes numpy's RNG (src/mean_shift.py:121-122) is drawn HERE, i.e. when the
        caller turns to this shape — shape by shape like the reference, however early stage 1 was
        queued; a redo on the synchronous path draws its own."""
        if flag is not None and int(flag) > 0:
            return None
        np.random.shuffle(np.arange(X.shape[0]))
        with torch.no_grad():
            _, indices, new_labels = self.nms(new_X, X, b=bw)
        return new_X, new_X[indices], bw, new_labels

    # -- src/mean_shift.py:45-79 ---------------------------------------------------------
    def mean_shift_(self, X, b, iterations=10, kernel_type="gaussian"):
        if kernel_type == "gaussian" and X.shape[-1] == 128:
            return mean_shift_iterations(X, b, iterations), X
        # Epanechnikov kernel / other embedding sizes: never used by the training path; plain
        # tensor expressions on the GPU (materialises N x N like the reference)
        new_X = X.clone()
        for _ in range(iterations):
            Kmat = self.kernel_between(new_X, X, kernel_type, b)
            D = 1 / torch.sum(Kmat, 1, keepdim=True)
            new_X = new_X + ((Kmat @ X) * D - new_X)
            new_X = new_X / torch.norm(new_X, dim=1, p=2, keepdim=True)
        return new_X, X

    # -- src/mean_shift.py:81-96 ---------------------------------------------------------
    def guard_mean_shift(self, embedding, quantile, iterations, kernel_type="gaussian"):
        while True:
            _, center, bandwidth, cluster_ids = self.mean_shift(embedding, 5000, quantile, iterations,
                                                                kernel_type=kernel_type)
            if center.shape[0] > 49 and torch.unique(cluster_ids).shape[0] > 49:
                quantile *= 2
            else:
                break
        return center, bandwidth, cluster_ids

    # -- src/mean_shift.py:98-113 --------------------------------------------------------
    def kernel_between(self, A, X, kernel_type, bw):
        dist = 2.0 - 2.0 * A @ torch.transpose(X, 1, 0)
        if kernel_type == "gaussian":
            return torch.exp(torch.clamp(-dist / (bw ** 2) / 2, max=75, min=-75))
        return torch.nn.functional.relu(3 / 4 * (1 - dist / (bw ** 2)))

    def kernel(self, X, kernel_type, bw):
        """N x N kernel matrix (diagnostics only; nothing on the training path calls it)."""
        return self.kernel_between(X, X, "gaussian" if kernel_type == "gaussian" else "epa", bw)

    # -- src/mean_shift.py:115-137 -------------------------------------------------------
    def compute_bandwidth(self, X, num_samples, quantile, defer_flags=False, defer_shuffle=False):
        """Mean over rows of the K-th smallest distance sqrt(2 - 2 x_i.x_j), K = int(quantile *
        num_samples).  Consumes numpy's RNG exactly like the reference (one shuffle of N).
        ``defer_flags``: do not look at the selection kernel's tie flags on the host (that is a
        synchronisation); return (bw, number of flagged rows as a tensor) instead."""
        require_cuda(X)
        N = X.shape[0]
        L = np.arange(N)
        if not (defer_shuffle and num_samples >= N):   # see ``finish``
            np.random.shuffle(L)
        if num_samples < N:
            X = X[h2d(L[0:num_samples], X.device)]
        # with num_samples >= N every row is used: the statistic does not depend on the order
        Kq = int(quantile * num_samples)
        Xc = X.detach().contiguous().unsqueeze(0)
        res = None
        if 1 <= Kq <= Xc.shape[1]:
            if ARITH == "fp16x2" and Xc.shape[2] == 128:
                # the statistic is a mean of K-th distances compared at 1e-5: both distance passes
                # on the fp16 matrix cores (values to ~1e-7), same selection engine
                res = K.dot_kth_unit(Xc, K.meanshift_h2_split(Xc), Xc.shape[1], Kq)
            elif ARITH == "bf16x3":
                # ... or on the bf16 cores with the error-free 3-way split: fp32-grade values, the
                # arithmetic of the iterations this bandwidth parametrises
                res = K.dot_kth_x3(Xc, Xc, Kq)
            if res is None:
                res = K.dot_select(Xc, Xc, Kq, want_value=True)
        if res is not None:
            kth_dot, flags = res
            kth_dot = kth_dot[0]
            if defer_flags:
                kth = 2.0 - 2.0 * kth_dot
                return torch.mean(torch.sqrt(torch.clamp(kth, min=1e-6))), (flags[0] != 0).sum()
            bad = torch.nonzero(flags[0]).flatten()
            if bad.numel() > 0:   # massively tied rows: redo those rows densely
                d = X[bad] @ X.t()
                kth_dot[bad] = torch.topk(d, Kq, dim=1, largest=True)[0][:, -1]
            kth = 2.0 - 2.0 * kth_dot
        else:
            kth = torch.empty(X.shape[0], device=X.device)
            for s in range(0, X.shape[0], 2048):   # shapes outside the kernel's fast path
                d = 2 - 2 * X[s:s + 2048] @ X.t()
                kth[s:s + 2048] = torch.topk(d, Kq, dim=1, largest=False)[0][:, -1]
        bw = torch.mean(torch.sqrt(torch.clamp(kth, min=1e-6)))
        return (bw, torch.zeros((), dtype=torch.int64, device=X.device)) if defer_flags else bw

    # -- src/mean_shift.py:139-179 -------------------------------------------------------
    def nms(self, centers, X, b):
        """Non-maximum suppression of the shifted points.  Returns (pruned centres, their row
        indices, per-point labels)."""
        require_cuda(centers, X)
        N = X.shape[0]
        centers = centers.detach()
        X = X.detach()
        # nearest centre of every point = largest dot product (2 - 2 dot is exact and decreasing)
        membership = None
        res = K.dot_select(X.contiguous().unsqueeze(0), centers.contiguous().unsqueeze(0), 1, want_value=False)
        if res is not None:
            idx, flags = res
            membership = idx[0, :, 0]
            bad = torch.nonzero(flags[0]).flatten()
            if bad.numel() > 0:
                membership[bad] = _first_argmax(X[bad] @ centers.t(), 1)
        else:
            membership = torch.empty(N, dtype=torch.int64, device=X.device)
            for s in range(0, N, 2048):
                membership[s:s + 2048] = _first_argmax(X[s:s + 2048] @ centers.t(), 1)
        # members per centre and the occupied centres in ascending order (np.unique + return_counts of
        # the reference) without leaving the device
        num_mem_cluster = torch.bincount(membership, minlength=N).to(torch.float32)
        uq = torch.nonzero(num_mem_cluster).flatten()
        # neighbours (distance < b, not b^2, as in the reference) of the occupied centres only
        dist = 2.0 - 2.0 * centers[uq] @ centers.t()
        score = (dist < b).float() * num_mem_cluster.reshape(1, -1)
        cluster_center_ids = torch.unique(_first_argmax(score, 1))
        centers = centers[cluster_center_ids]
        labels = _first_argmax(centers @ X.t(), 0)
        return centers, cluster_center_ids, labels

    def pdist(self, x, y):
        return torch.sum((x.unsqueeze(1) - y.unsqueeze(0)) ** 2, 2)
