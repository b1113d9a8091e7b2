"""Headline benchmark: shapes/s for forward+backward(+all-reduce+Adam) of the ParSeNet hot path
on synthetic 10 000-point clouds (BASELINE.json).  One process per GPU; for N > 1 launch with
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.  See DESIGN.md (Measurement) for the definitions of ``roofline``
and ``cpu_baseline``.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / fp16 dense peak (v_mfma_f32_32x32x16_{bf16,f16})
HBM_PEAK_GBS = 8000.0


POOL = 16             # distinct shapes per rank the timed steps rotate through
PRETRAIN_POOL = 64    # disjoint shapes of the pre-training stand-in (cfg5)


def build_workload(name, device, rank, pretrain=2000):
    from parsenet_codebase_amd import workloads
    if name == "cfg4":
        B, N = 4, 10000
        return workloads.ParsenetSegStep(device, batch=B, num_points=N, first_shape=rank * POOL, pool=POOL), {
            "workload": "cfg4: ParSeNet seg-only points+normals, 10k pts, batch 4 per GPU "
                        "(PrimitivesEmbeddingDGCNGn mode 5, k=80, triplet+NLL, fwd+bwd+allreduce+Adam)",
            "batch_per_gpu": B, "points": N, "k": 80,
            "pool": "%d distinct shapes per rank (ids %d..%d on rank 0), a new batch of %d every step"
                    % (POOL, 0, POOL - 1, B)}
    if name == "cfg5":
        B, N = 4, 10000
        step = workloads.ParsenetE2EStep(device, batch=B, num_points=N, first_shape=rank * POOL,
                                         pretrain_steps=pretrain, pool=POOL, pretrain_pool=PRETRAIN_POOL)
        return step, {
            "workload": "cfg5: ParSeNet e2e (seg + mean-shift 10 it. + per-segment spline/primitive fit + "
                        "Chamfer/residual), 10k pts, batch 4 per GPU, fwd+bwd+allreduce+Adam",
            "batch_per_gpu": B, "points": N, "k": 80,
            "pool": "%d distinct shapes per rank (ids rank*%d ..), a new batch of %d every step; every timed "
                    "shape is HELD OUT from the pre-training" % (POOL, POOL, B),
            "init": "segmentation network after %d deterministic seg-only steps (Adam 1e-2) over %d shapes with ids "
                    "%d.. — disjoint from the timed pool — run by rank 0 and broadcast (stand-in for the reference's "
                    "pretrained parsenet_with_normals.pth, train_parsenet_e2e.py:82-84); frozen random-init SplineNets"
                    % (pretrain, PRETRAIN_POOL, workloads.PRETRAIN_FIRST_SHAPE),
            "pretrain_final_loss": step.pretrain_loss}
    if name in ("cfg2", "cfg3"):
        B, N = 32, 700
        closed = name == "cfg3"
        return workloads.SplineNetStep(device, closed=closed, batch=B, num_points=N, first_shape=rank * B), {
            "workload": "%s: %s SplineNet (DGCNNControlPoints mode %d, k=10), 700-pt patches, batch 32 per GPU, "
                        "one-sided Chamfer + permutation regression%s, fwd+bwd+allreduce+Adam"
                        % (name, "closed" if closed else "open", int(closed), "" if closed else " + Laplacian"),
            "batch_per_gpu": B, "points": N, "k": 10}
    if name == "stub":
        return StubStep(device, rank, pretrain), {"workload": "stub: control-flow self-test, no kernels"}
    raise SystemExit("unknown workload %r" % name)


class StubStep:
    """--workload stub: a 2-layer perceptron on the CPU with the step interface of the real
    workloads — pre-training on rank 0 + broadcast, flat gradient bucket, one all-reduce per step.
    It exists so that the CPU test suite can drive bench.py's WHOLE control flow (pretrain ->
    broadcast -> timed loop -> dense re-run -> profiled steps -> MAX-reduce -> rank-0 line) on
    two gloo ranks: a deadlock or a rank mismatch must not first appear on the 8-GPU node."""

    def __init__(self, device, rank, pretrain):
        from parsenet_codebase_amd import workloads
        from parsenet_codebase_amd.dp import FitStatusError, FlatGradBucket
        self._fit_status_error = FitStatusError
        torch.manual_seed(100 + rank)            # ranks start from DIFFERENT weights: the broadcast must fix that
        self.model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1)).to(device)
        self.bucket = FlatGradBucket(self.model.parameters())
        self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-2)
        self.batch, self.num_points, self.pool = 4, 8, 8
        g = torch.Generator().manual_seed(7 + rank)
        self.data = torch.randn(self.pool, 8, generator=g).to(device)
        self.cursor = 0
        self.pretrain_loss = None
        # PN_STUB_FAIL="rank:step": that rank's status check raises in its step-th call (counted from the
        # start of the timed region) — the CPU suite's stand-in for a degenerate segment on one rank
        fail = os.environ.get("PN_STUB_FAIL", "")
        self.fail_at = int(fail.split(":")[1]) if fail and int(fail.split(":")[0]) == rank else -1
        self.calls, self.skipped_steps = 0, 0

        def train():
            for _ in range(pretrain):
                self.step()
        workloads.train_on_rank0_then_broadcast(self.model, self.bucket, train)
        self.cursor = 0

    def shapes_per_step(self):
        return self.batch

    def step(self):
        x = self.data[self.cursor:self.cursor + self.batch]
        self.cursor = (self.cursor + self.batch) % self.pool
        self.bucket.zero()
        loss = (self.model(x) ** 2).mean()
        loss.backward()
        if not self.bucket.collective:             # rank 0's pre-training: nobody to agree with
            self.opt.step()
            return loss
        self.calls += 1

        def finish():
            if self.calls == self.fail_at:
                raise self._fit_status_error("injected: degenerate segment on this rank")
        _, _, took = self.bucket.finish_or_skip(finish, self.opt)
        self.skipped_steps += 0 if took else 1
        return loss


def pmc_traffic(kernel_name, arith, shapes_per_launch=1, planned=False):
    """HBM-side bytes per launch of a mean-shift kernel from the committed PMC run of the same
    launch configuration (profiles/: FETCH_SIZE and WRITE_SIZE in KB, separate passes; FETCH
    doubled for 16-byte-per-lane reads as the guide's gfx950 correction prescribes).  None if
    the file is not there — bench.py never profiles counters itself (rocprofv3 does, in passes of
    their own: tools/evidence_round5.sh).  The round-3 / round-4 files were collected inside
    ``bench.py --workload cfg5`` (batched launches of 4 shapes; round 4: ``--profile-only``, i.e. only
    the launches the roofline is quoted on): one for planned launches (256 workgroups), one for dense
    ones; the round-1 files are per shape."""
    import csv
    files = {"fp16x2": ("r01_meanshift_h2_pmc.csv", "pn_msh_kernel<%d>"),
             "bf16x3": ("r01_meanshift_x3_pmc.csv", "pn_ms3_kernel<%d>"),
             "f32": ("r01_meanshift_f32_pmc.csv", "pn_ms_kernel<%d>")}
    scale = float(shapes_per_launch)
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    if arith == "bf16x3" and shapes_per_launch == 4:
        # (round 5: the planned file of the DEFAULT process — forward-only plans at a bound of 1e-6, the launches the
        # line times — comes first; the older planned files hold the launches of a process with the dense backward
        # passes, whose plans at 1e-9 keep more pairs)
        fwd_only = os.environ.get("PARSENET_MS_ROWS_BWD", "1") != "0"
        cand = ((["r05_meanshift_x3_planned_fwd_only_cfg5_pmc.csv"] if fwd_only else []) +
                ["r05_meanshift_x3_planned_cfg5_pmc.csv", "r04_meanshift_x3_planned_cfg5_pmc.csv",
                 "r03_meanshift_x3_planned_cfg5_pmc.csv", "r02_meanshift_x3_sparse_cfg5_pmc.csv"] if planned else
                ["r05_meanshift_x3_dense_cfg5_pmc.csv", "r04_meanshift_x3_dense_cfg5_pmc.csv",
                 "r03_meanshift_x3_dense_cfg5_pmc.csv", "r02_meanshift_x3_batch4_pmc.csv"])
        cand = [c for c in cand if os.path.exists(os.path.join(prof, c)) and
                "FETCH_SIZE" in open(os.path.join(prof, c)).read()] or cand
        files["bf16x3"] = (next((c for c in cand if os.path.exists(os.path.join(prof, c))), cand[-1]),
                           "pn_ms3_kernel<%d>")
        scale = 1.0
    fn = os.path.join(prof, files[arith][0])
    idx = {"meanshift_fwd": 0, "meanshift_bwd_rows": 1, "meanshift_bwd_cols": 2}.get(kernel_name)
    if idx is None or not os.path.exists(fn):
        return None
    want = (files[arith][1] % idx)[:-1]       # "pn_ms3_kernel<2": also "<2, true>" (ping-pong schedule)
    rows = [r for r in csv.DictReader(open(fn)) if r["kernel"].startswith(want)]
    grids = sorted({int(r["grid_size"]) for r in rows if r.get("grid_size", "").isdigit()})
    if len(grids) > 1:      # dense launches: the largest grid; planned (flat schedule): the smallest
        pick = str(grids[0] if planned else grids[-1])
        rows = [r for r in rows if r["grid_size"] == pick]
    vals = {r["counter"]: float(r["avg_per_launch"]) for r in rows}
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 * scale


def _scale(x, k):
    return None if x is None else x * k


def kernel_roofline(step, nprof):
    """Average launch duration of every kernel family over ``nprof`` profiled steps (HIP events
    on the launch stream, recorded inside the C library), and the roofline entry of the
    dominant one."""
    from parsenet_codebase_amd import _lib
    from parsenet_codebase_amd import mean_shift as _ms
    _lib.prof_reset()
    _lib.prof_enable(True)
    _lib.meanshift_exec_tiles()                # clear the executed-work counters of the mean-shift kernels
    os.environ["PARSENET_MS_STATS"] = "1"      # active fractions of the block-sparse mean-shift plans
    plan_stats = []                            # one entry per profiled step: mean over its 10 iterations
    for _ in range(nprof):
        _ms.LAST_PLAN_STATS = None
        step.step()
        if _ms.LAST_PLAN_STATS:
            st = _ms.LAST_PLAN_STATS
            plan_stats.append([sum(t[c] for t in st) / len(st) for c in range(4)])
    torch.cuda.synchronize()
    os.environ.pop("PARSENET_MS_STATS", None)
    _lib.prof_enable(False)
    exec_pairs = dict(zip(("meanshift_fwd", "meanshift_bwd_rows", "meanshift_bwd_cols"), _lib.meanshift_exec_tiles()))
    res = _lib.prof_results()
    if not res:
        return None, {}
    table = {k: {"ms_total": v[0], "calls": v[1], "avg_ms": v[0] / max(v[1], 1)} for k, v in res.items()}
    dom = max(table, key=lambda k: table[k]["ms_total"])
    B, N, k = step.batch, step.num_points, 80
    if not hasattr(step, "labels"):   # SplineNet steps: k = 10
        k = 10
    avg_s = table[dom]["avg_ms"] * 1e-3
    # algorithmic work per launch (DESIGN.md / SURVEY.md §8d)
    if dom.startswith("knn_mfma_pass"):
        C = 64 if dom.endswith("c64") else (6 if dom.endswith("pn") else 3)
        # one kNN layer needs B*N^2*(2C+3) FLOP once; the exact two-pass selection launches the
        # distance kernel twice, so each launch is credited with half of the layer's work
        flops = 0.5 * B * N * N * (2 * C + 3)
        ach = flops / avg_s / 1e12
        roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                "avg_launch_ms": table[dom]["avg_ms"], "mfma": "v_mfma_f32_32x32x2_f32",
                # each of the two passes evaluates ALL N^2 distances: what the matrix cores execute
                "executed_tflops": 2.0 * ach, "executed_peak_tflops": MFMA_F32_PEAK_TFLOPS}
    elif dom.startswith("edgeconv"):
        # fused edge-conv: read x-products (B,N,2Cout) + idx, write (B,N,Cout) x3 small outputs
        Cout = 64
        nbytes = 4.0 * B * N * 2 * Cout + 8.0 * B * N * k + 4.0 * B * N * Cout * 2 + B * N * Cout
        ach = nbytes / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": table[dom]["avg_ms"]}
    elif dom.startswith("meanshift"):
        # one shape, one iteration: units of 2*N^2*d FLOP (d = 128): forward 2, row pass 3, column pass 4
        units = {"meanshift_fwd": 2, "meanshift_bwd_rows": 3, "meanshift_bwd_cols": 4}[dom]
        # the stage-wise fitting path clusters all shapes of the batch in one launch per pass
        shapes_per_launch = B if getattr(step, "batched", False) else 1
        flops = units * 2.0 * N * N * 128 * shapes_per_launch
        ach = flops / avg_s / 1e12
        sparse = None
        executed = 1.0
        UNITS = {"meanshift_fwd": 2, "meanshift_bwd_rows": 3, "meanshift_bwd_cols": 4}
        PAIR_FLOP = 2.0 * 32 * 32 * 128              # one (resident tile, streamed tile) pair, one GEMM unit

        def executed_tflops(fam):
            """Piece-product FLOPs the launches of ``fam`` EXECUTED (counted by the kernels themselves:
            csrc/meanshift_x3.h pn_ms3_exec) over the event time of the same launches."""
            if fam not in table or not exec_pairs.get(fam):
                return None
            return exec_pairs[fam] * PAIR_FLOP * UNITS[fam] * 6.0 / (table[fam]["ms_total"] * 1e-3) / 1e12
        if _ms.ARITH == "bf16x3" and plan_stats:
            # block-sparse launches: the waves run the GEMMs of the tile pairs the plan keeps; the
            # others are rigorously below 1e-9 of the smallest row sum (csrc/meanshift_x3.h)
            col = {"meanshift_fwd": 1, "meanshift_bwd_rows": 2, "meanshift_bwd_cols": 3}[dom]

            def mmm(c):
                v = [t[c] for t in plan_stats]
                return {"min": min(v), "mean": sum(v) / len(v), "max": max(v)}
            sparse = {"tile_pairs_executed": mmm(0), "block_lists_visited": mmm(col),
                      "steps_sampled": len(plan_stats),
                      "note": "fractions of the dense N^2 tile pairs, per profiled step (mean over its 10 "
                              "iterations); the profiled steps visit every batch of the timed pool once"}
        if _ms.ARITH == "bf16x3" and exec_pairs.get(dom):
            # share of the dense work the profiled launches executed: counted pairs / dense pairs
            nt = (N + 31) // 32
            executed = exec_pairs[dom] / float(table[dom]["calls"] * shapes_per_launch * nt * nt)
        if _ms.ARITH in ("bf16x3", "fp16x2"):
            # every fp32 product is formed from 6 bf16 (3 fp16) piece products on the 16-bit matrix
            # cores (fp32 accumulate).  `achieved` / `frac`: the piece-product FLOPs the launch EXECUTES
            # (tile pairs kept by the plan only) against the dense 16-bit MFMA peak — the figure
            # SQ_VALU_MFMA_BUSY_CYCLES / 32 x 32768 FLOP / duration reproduces.  The work of the
            # reference's dense fp32 iteration that this launch stands for is reported separately.
            pieces = 6.0 if _ms.ARITH == "bf16x3" else 3.0
            ach_exec = pieces * ach * executed
            if _ms.ARITH == "bf16x3" and executed_tflops(dom) is not None:
                ach_exec = executed_tflops(dom)          # counter-backed: executed work / the same launches' time
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach_exec, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach_exec / MFMA_BF16_PEAK_TFLOPS,
                    "traffic": pmc_traffic(dom, _ms.ARITH, shapes_per_launch, planned=bool(plan_stats)),
                    "avg_launch_ms": table[dom]["avg_ms"],
                    "mfma": ("v_mfma_f32_32x32x16_bf16, 6 piece products per fp32 product (bf16x3 split)"
                             if _ms.ARITH == "bf16x3" else
                             "v_mfma_f32_32x32x16_f16, 3 piece products per fp32 product (scaled fp16x2 split)"),
                    "algorithmic_dense_equiv": {
                        "tflops_fp32": ach, "peak_fp32_via_split": MFMA_BF16_PEAK_TFLOPS / pieces,
                        "ratio": ach * pieces / MFMA_BF16_PEAK_TFLOPS,
                        "note": "SURVEY 8d work of the reference's dense iteration (units x 2 N^2 d per shape) / "
                                "launch duration; exceeds 1 when the plan skips tile pairs — not a roofline fraction"}}
            if _ms.ARITH == "bf16x3":
                # all three passes, each from its own counter and its own launches' event time
                roof["passes"] = {fam: {"frac": executed_tflops(fam) / MFMA_BF16_PEAK_TFLOPS,
                                        "executed_tflops": executed_tflops(fam),
                                        "avg_launch_ms": table[fam]["avg_ms"], "launches": table[fam]["calls"],
                                        "tile_pairs_executed": exec_pairs[fam],
                                        "share_of_dense_pairs": exec_pairs[fam] / float(
                                            table[fam]["calls"] * shapes_per_launch * ((N + 31) // 32) ** 2)}
                                  for fam in UNITS if executed_tflops(fam) is not None}
                roof["executed_work"] = ("tile pairs counted inside the kernels (pn_meanshift_x3_exec_tiles) x 2*32*32*128 "
                                         "FLOP x GEMM units (2/3/4) x 6 piece products / HIP-event time of the same "
                                         "launches; = SQ_VALU_MFMA_BUSY_CYCLES / 32 x 32768 FLOP of a counter run")
            if sparse:
                roof["block_sparse"] = sparse
        else:
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
                    "traffic": pmc_traffic(dom, "f32", shapes_per_launch),
                    "avg_launch_ms": table[dom]["avg_ms"], "mfma": "v_mfma_f32_32x32x2_f32"}
    else:
        roof = {"bound": "hbm", "kernel": dom, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None, "traffic": None, "avg_launch_ms": table[dom]["avg_ms"]}
    return roof, table


_HOST_THREADS_AT_START = 0


def cpu_baseline(name, state=None, state_note="random-init weights"):
    """The torch-CPU oracle (restatement of the reference's algorithm) on ONE shape of the same
    workload (shape 0 of rank 0's timed pool), all host cores; ``state`` = the state_dict the GPU
    side starts its timed region from, so that both sides cluster and fit the same segments.
    Bounded: up to 3 passes, no new pass once 45 s are spent (median reported)."""
    import numpy as np
    from oracle import ref_fitting as RF, ref_torch as R
    from parsenet_codebase_amd import synthetic
    torch.manual_seed(0)
    np.random.seed(0)
    # the GPU legs run with one host thread (dp.limit_host_threads); the oracle gets the CPUs the process
    # may really use: the cgroup quota, not the visible cores (128 threads on a 16-CPU quota are throttled)
    from parsenet_codebase_amd import dp as _dp
    cores = min(_HOST_THREADS_AT_START or torch.get_num_threads(), _dp.usable_cpus())
    torch.set_num_threads(cores)
    if name in ("cfg2", "cfg3"):
        return cpu_baseline_splinenet(name, cores)
    model = R.PrimitivesEmbeddingDGCNGn(embedding=True, emb_size=128, primitives=True, num_primitives=10,
                                        loss_function=R.EmbeddingLoss(1.0).triplet_loss, mode=5,
                                        num_channels=6, nn_nb=80)
    if state is not None:
        model.load_state_dict({k: v.detach().cpu() for k, v in state.items()})
    pts, nrm, lab, prim = synthetic.make_batch(0, 1, 10000)
    x = torch.from_numpy(np.ascontiguousarray(np.concatenate([pts, nrm], 2).transpose(0, 2, 1)))
    primt = torch.from_numpy(prim)
    ev = None
    if name == "cfg5":
        model.eval()
        torch.manual_seed(1)        # frozen random-init SplineNets (any fixed weights: their cost is what counts)
        ev = RF.Evaluation(R.DGCNNControlPoints(20, 10, 1), R.DGCNNControlPoints(20, 10, 0))
    times, fitted, clusters = [], None, None
    t_begin = time.time()
    for it in range(3):
        if it and time.time() - t_begin > 45.0:
            break
        t0 = time.time()
        model.zero_grad()
        e, p, l = model(x, lab, True)
        loss = l.mean() + R.primitive_loss(p, primt)
        if ev is not None:
            res, extra = ev.fitting_loss(e.permute(0, 2, 1), torch.from_numpy(pts), torch.from_numpy(nrm), lab, prim,
                                         quantile=0.025, iterations=10, lamb=0.1)
            loss = loss + res[0]
            fitted = sum(1 for v in extra[0].values() if v is not None)
            clusters = int(np.unique(extra[1]).shape[0])
        loss.backward()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    what = ""
    if ev is not None:
        what = "; this pass found %d clusters and fitted %d segments" % (clusters, fitted)
    return {"value": 1.0 / med, "unit": "shapes/s", "cores": cores, "kind": "port",
            "sample": "torch-CPU oracle (reference algorithm restated), %s, %s, 1 shape (id 0 of the timed pool) x "
                      "10000 pts fwd+bwd, median of %d passes (%s s)%s"
                      % (name, state_note, len(times), ", ".join("%.1f" % t for t in times), what)}


def oracle_splinenet_step(model, closed, points, control_points, nu, nv, loss_weight=0.9):
    """The reference's SplineNet training step restated on the oracle (CPU): returns
    (loss, cd, reg, lap, output).  Also used by tests/test_workloads_gpu.py as the checker."""
    from oracle import ref_fitting as RF
    B = points.shape[0]
    out = model(points)
    cd, _ = RF.spline_reconstruction_loss_one_sided(nu, nv, out, points, B, 20)
    if closed:
        reg, _ = RF.control_points_permute_closed_reg_loss(out, control_points, 20, 20)
        return reg * loss_weight + cd * (1 - loss_weight), cd, reg, None, out
    reg, perm = RF.control_points_permute_reg_loss(out, control_points, 20)
    lap = RF.laplacian_loss(out.reshape(B, 20, 20, 3), perm)
    return reg * loss_weight + (cd + lap) * (1 - loss_weight), cd, reg, lap, out


def cpu_baseline_splinenet(name, cores):
    import numpy as np
    from oracle import ref_fitting as RF, ref_torch as R
    from parsenet_codebase_amd import synthetic
    closed = name == "cfg3"
    B = 32
    model = R.DGCNNControlPoints(20, 10, 1 if closed else 0)
    nu, nv = RF.uniform_knot_bspline(20, 20, 3, 3, 30 if closed else 40)
    nu, nv = torch.from_numpy(nu.astype(np.float32)), torch.from_numpy(nv.astype(np.float32))
    pts, ctrl = synthetic.make_spline_patches(0, B, 700, 20, closed)
    x = torch.from_numpy(np.ascontiguousarray(pts.transpose(0, 2, 1)))
    cp = torch.from_numpy(ctrl)
    times = []
    for it in range(5):
        t0 = time.time()
        model.zero_grad()
        oracle_splinenet_step(model, closed, x, cp, nu, nv)[0].backward()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": B / med, "unit": "shapes/s", "cores": cores, "kind": "port",
            "sample": "torch-CPU oracle (reference algorithm restated), %s, one batch of 32 x 700 pts fwd+bwd, "
                      "median of %d passes" % (name, len(times))}


def _dtype_label(workload):
    """Arithmetic type of the path.  The opt-in fp16x2 mean-shift keeps 22 significand bits per
    operand — narrower than the reference's fp32 — and is labelled as such."""
    from parsenet_codebase_amd import mean_shift as _ms
    if workload == "cfg5" and _ms.ARITH == "fp16x2":
        return "f32 except mean-shift products (fp16x2 split: 22-bit operands, NOT fp32-wide)"
    return "f32"


def _ms_arith():
    from parsenet_codebase_amd import mean_shift as _ms
    return {"fp16x2": "fp32 via scaled fp16x2 operand split on the fp16 matrix cores, fp32 accumulate",
            "bf16x3": "fp32 via error-free bf16x3 operand split on the bf16 matrix cores, fp32 accumulate",
            }.get(_ms.ARITH, "fp32 matrix cores (v_mfma_f32_32x32x2_f32)")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(gpus):
    """``python bench.py --gpus N`` without a launcher environment: start N ranks (one per GPU) as
    CHILDREN through torch.distributed.run and relay rank 0's JSON line and the exit status.  This
    process has not touched the GPU (no HIP call, no torch.cuda.is_available()) and never
    replaces itself: the children are ordinary subprocesses."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the CPUs this job may USE (cgroup quota: 16 of 256 visible on this pool), not the visible ones
    from parsenet_codebase_amd.dp import usable_cpus
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // gpus)))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print("bench.py: a rank failed (torch.distributed.run exit status %d)" % rc, file=sys.stderr)
    return rc


def launch_selftest():
    """--selftest-launch: every rank joins the process group (gloo when no GPU is visible), the
    ranks count themselves with one all-reduce, rank 0 prints the JSON line.  Lets the CPU test
    suite cover the ``--gpus N`` start-up path without an accelerator."""
    from parsenet_codebase_amd import dp
    rank, world, device = dp.init_from_env(backend="gloo")
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_selftest": True, "n_gpus": world, "world_size_observed": int(t.item())}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=-1,
                    help="extra steps with the in-library kernel timers on (default: one pass over the pool)")
    ap.add_argument("--no-dense", action="store_true", help="cfg5: skip the second timed run with dense mean-shift launches")
    ap.add_argument("--pretrain", type=int, default=2000,
                    help="cfg5: deterministic seg-only steps before the timed region (see workloads.ParsenetE2EStep)")
    ap.add_argument("--profile-only", action="store_true",
                    help="no timed runs: pre-train (or load PARSENET_PRETRAIN_CACHE), then ONLY the profiled pass over "
                         "the pool with mean-shift launches of one kind (PARSENET_MS_SPARSE=1 planned / 0 dense; auto "
                         "counts as planned) and print the line with `value` null — the process to put under "
                         "rocprofv3: its kernel trace then holds the launches the roofline is quoted on, not the "
                         "warm-up pass on a synthetic embedding or the other launch kind")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("PARSENET_HOST_THREADS", "1")),
                    help="torch intra-op CPU threads while the GPU legs run (default 1, see dp.limit_host_threads; 0: "
                         "torch's default of one per visible core); the CPU baseline gets the usable CPUs")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))     # before anything touches the GPU
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    if args.selftest_launch:
        return launch_selftest()

    from parsenet_codebase_amd import dp
    global _HOST_THREADS_AT_START
    os.environ["PARSENET_HOST_THREADS"] = str(args.host_threads)     # init_from_env applies it
    _HOST_THREADS_AT_START = torch.get_num_threads()
    stub = args.workload == "stub"
    rank, world, device = dp.init_from_env(backend="gloo" if stub else None)
    if device.type != "cuda" and not stub:
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product has no CPU path)")
    if stub:
        device = torch.device("cpu")
    multi = dp.multi_rank()     # several ranks, or PARSENET_FORCE_COLLECTIVE=1 (the RCCL path on one GPU)
    import copy
    import numpy as np

    step, cfg = build_workload(args.workload, device, rank, args.pretrain)

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if multi:
            dist.barrier()
        sync()

    def snapshot():
        return {"model": copy.deepcopy(step.model.state_dict()), "opt": copy.deepcopy(step.opt.state_dict()),
                "cursor": getattr(step, "cursor", 0)}

    def restore(snap):
        step.model.load_state_dict(snap["model"])
        step.opt.load_state_dict(copy.deepcopy(snap["opt"]))
        if hasattr(step, "cursor"):
            step.cursor = snap["cursor"]

    per_rank = {}

    def timed_run(tag="default"):
        """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks.  Every rank also
        notes when ITS OWN K steps were done (synchronize, before the closing barrier): the spread over the
        ranks — data-dependent segment counts, mean-shift retries — goes into the line as
        ``per_rank_ms`` (per step, min / max over the ranks)."""
        np.random.seed(1000 + rank)
        for _ in range(args.warmup):
            step.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.step()
        sync()
        own = time.perf_counter() - t0
        barrier()
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            lo = torch.tensor([own], dtype=torch.float64, device=device)
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            own_lo, own_hi = float(lo.item()), float(hi.item())
        else:
            own_lo = own_hi = own
        per_rank[tag] = {"min": round(1e3 * own_lo / max(args.steps, 1), 3),
                         "max": round(1e3 * own_hi / max(args.steps, 1), 3)}
        return el

    # every run (default launches, dense launches, profiled steps) starts from the pre-trained state.
    # Before it: one untimed pass over the pool from that state, then the state is restored — the
    # batches of the pool have different segment counts, i.e. different tensor shapes in the fitting
    # stage, and the first visit of a shape pays for allocator growth and library heuristics that
    # belong to no steady-state step (the W warm-up steps the caller asks for follow as usual).
    if hasattr(step, "warm_paths") and not args.profile_only:
        step.warm_paths()
    elif hasattr(step, "_warmed"):
        step._warmed = True            # --profile-only: no warm-up pass on the synthetic embedding in the trace
    # everything allocated so far (modules, pools, compiled wrappers) is permanent: take it out of the
    # cyclic collector's generations so that a full collection in the middle of a step stays short
    import gc
    gc.collect()
    gc.freeze()
    start = snapshot()
    # (numpy's generator feeds the triplet sampling: seeded here too, so that the pass over the pool — whose
    # steps are counted in clusters_per_shape — does not depend on whether the pre-training ran in this
    # process or came from PARSENET_PRETRAIN_CACHE)
    np.random.seed(999 + rank)
    if hasattr(step, "pool") and not stub and not args.profile_only:
        for _ in range(max(1, step.pool // step.batch)):
            step.step()
        sync()
        restore(start)
    elapsed = None if args.profile_only else timed_run()

    # cfg5: the same steps once more with every mean-shift launch dense (PARSENET_MS_SPARSE=0 at run
    # time): the block-sparse plans are data dependent, the dense value is their floor
    elapsed_dense = None
    ms_mode = None
    _ms = None
    if args.workload == "cfg5":
        from parsenet_codebase_amd import mean_shift as _ms
        if _ms.ARITH == "bf16x3":
            ms_mode = _ms.SPARSE
            calls_timed = dict(_ms.CALLS)
            if args.profile_only:          # no timed steps to take the kind from: the environment decides
                calls_timed = {"planned": 0 if ms_mode is False else 1, "dense": 1 if ms_mode is False else 0}
    if ((ms_mode not in (None, False) and not args.no_dense) or stub) and not args.profile_only:
        restore(start)
        if ms_mode is not None:
            _ms.SPARSE = False
        try:
            elapsed_dense = timed_run("dense")
        finally:
            if ms_mode is not None:
                _ms.SPARSE = ms_mode

    # The profiled steps contain the gradient all-reduce, a collective: EVERY rank runs them
    # (only rank 0 keeps the per-kernel table), otherwise rank 0 would pair its all-reduce with
    # the other ranks' barrier.  Default: one pass over the timed pool.  Their mean-shift launches
    # are of ONE kind — the kind most timed steps used — so that a kernel family's average launch
    # duration and its executed work belong together.
    restore(start)
    nprof = args.profile_steps
    if nprof < 0:
        nprof = max(1, getattr(step, "pool", step.batch) // step.batch)
    census = None
    if ms_mode is not None:
        tally = torch.tensor([calls_timed["planned"], calls_timed["dense"]], dtype=torch.float64, device=device)
        if multi:     # ONE decision for all ranks: the steps below contain collectives
            dist.all_reduce(tally)
        mostly_planned = bool(tally[0] > tally[1])
        _ms.SPARSE = mostly_planned
    try:
        roof, table = (None, {}) if stub else kernel_roofline(step, nprof)
        if ms_mode is not None and not mostly_planned and not args.profile_only:
            # the plans of the timed pool all the same (one planned pass, not timed, not profiled):
            # how much of the dense work a planned launch WOULD execute on this embedding
            restore(start)
            _ms.SPARSE = True
            os.environ["PARSENET_MS_STATS"] = "1"
            fr = []
            for _ in range(max(1, step.pool // step.batch)):
                _ms.LAST_PLAN_STATS = None
                step.step()
                if _ms.LAST_PLAN_STATS:
                    fr.append(sum(t[0] for t in _ms.LAST_PLAN_STATS) / len(_ms.LAST_PLAN_STATS))
            os.environ.pop("PARSENET_MS_STATS", None)
            torch.cuda.synchronize()
            if fr:
                census = {"tile_pairs_a_plan_would_keep": {"min": min(fr), "mean": sum(fr) / len(fr), "max": max(fr)},
                          "steps_sampled": len(fr)}
    finally:
        if ms_mode is not None:
            _ms.SPARSE = ms_mode
    if ms_mode is not None and roof is not None:
        roof["meanshift_launches"] = {
            "mode": {True: "planned (PARSENET_MS_SPARSE=1)", False: "dense (PARSENET_MS_SPARSE=0)"}.get(
                ms_mode, "auto: plan %d calls, go dense for %d calls when their plans keep more than %.2f of the "
                         "tile pairs on average" % (_ms.AUTO_SAMPLES, _ms.AUTO_DENSE_STEPS, _ms.AUTO_DENSE_ABOVE)),
            "timed_and_warmup_calls": calls_timed, "profiled_as": "planned" if mostly_planned else "dense"}
        if census:
            roof["meanshift_launches"].update(census)
    if stub:
        for _ in range(nprof):
            step.step()
    # The CPU baseline: rank 0, outside every timed region (with several ranks the others wait at the barrier
    # below: a bounded sample, <= 45 s).  The stub workload reports a token entry so that the multi-rank control
    # flow of this leg is covered on gloo ranks too.
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and not args.profile_only:
        if stub:
            cpu = {"value": None, "unit": "shapes/s", "cores": 1, "kind": "port", "sample": "stub workload: none"}
        else:
            cpu = cpu_baseline(args.workload, start["model"] if args.workload == "cfg5" else None,
                               "the GPU side's pre-trained state_dict" if args.workload == "cfg5"
                               else "random-init weights")
    if multi:
        dist.barrier()

    if rank == 0:
        if hasattr(step, "segments_per_shape"):
            sps = step.segments_per_shape()
            cfg = dict(cfg, segments_per_shape=round(sps["fitted"], 2), clusters_per_shape=round(sps["clusters"], 2))
        shapes = step.shapes_per_step() * world * args.steps
        out = {
            "metric": ("shapes/sec fwd+bwd on 10k-pt clouds" if args.workload in ("cfg4", "cfg5")
                       else "shapes/sec fwd+bwd on 700-pt spline patches"),
            "value": None if elapsed is None else shapes / elapsed,
            "unit": "shapes/s",
            "n_gpus": world,
            "world_size_observed": dist.get_world_size() if multi else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": None if elapsed is None else 1e3 * elapsed / args.steps,
            # each rank's own time for its K steps (synchronize, before the closing barrier), per step
            "per_rank_ms": per_rank.get("default"),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if stub else _dtype_label(args.workload),
            "data": "synthetic",
            # host side of the GPU legs: torch intra-op threads / CPUs the cgroup grants / cores visible
            "host": {"threads": args.host_threads or _HOST_THREADS_AT_START, "usable_cpus": dp.usable_cpus(),
                     "visible_cores": os.cpu_count()},
            "config": dict(cfg, parallelism="dp%d" % world, global_batch=step.shapes_per_step() * world,
                           **({"collective": "forced on one rank (PARSENET_FORCE_COLLECTIVE=1)"}
                              if dp.collective_forced() and world == 1 else {}),
                           **({"meanshift_products": _ms_arith()} if args.workload == "cfg5" else {})),
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": {k: round(v["avg_ms"], 4) for k, v in sorted(table.items())},
            # the same families as GPU milliseconds per step (avg launch x launches per step; launches in brackets)
            "kernel_ms_per_step": {k: [round(v["ms_total"] / max(nprof, 1), 3), round(v["calls"] / max(nprof, 1), 1)]
                                   for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms_total"])},
        }
        if getattr(step, "skipped_steps", 0):
            # a rank's fitting stage raised: the step was dropped on EVERY rank (no reduction, no
            # optimizer move; train_parsenet_e2e.py:243-257) — counted over all runs of this process
            out["skipped_steps"] = step.skipped_steps
        if elapsed_dense is not None:
            # same weights, same shapes, same RNG stream, every mean-shift launch dense
            out["value_dense"] = shapes / elapsed_dense
            out["ms_per_step_dense"] = 1e3 * elapsed_dense / args.steps
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
