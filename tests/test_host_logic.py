"""CPU: host-side logic that needs no GPU — synthetic generator, B-spline bases, sharding, the
flat gradient bucket under a world-size-2 gloo group, and the product's refusal of CPU tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_shapes_are_deterministic_and_normalised():
    from parsenet_codebase_amd import synthetic
    a = synthetic.make_shape(3, 2000)
    b = synthetic.make_shape(3, 2000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    pts, nrm, lab, prim = a
    assert pts.shape == (2000, 3) and pts.dtype == np.float32
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-5
    ext = pts.max(0) - pts.min(0)
    assert abs(ext.max() - 1.0) < 1e-5                       # divided by the largest extent
    assert ext.argmin() == 0                                  # minor PCA axis rotated to x
    assert set(np.unique(prim)) <= {1, 2, 3, 4, 5, 9}
    assert len(np.unique(lab)) >= 4
    P, CP = synthetic.make_spline_patches(0, 2, 700)
    assert P.shape == (2, 700, 3) and CP.shape == (2, 20, 20, 3)


def test_bspline_basis_known_answers():
    """partition of unity, 4 non-zeros per row, N_{8,3}(0.5) = 1/48."""
    from parsenet_codebase_amd.bspline import basis_function_one, uniform_knot_bspline, uniform_knots
    for grid in (30, 40):
        nu, nv = uniform_knot_bspline(20, 20, 3, 3, grid)
        assert nu.shape == (grid, 20) and np.abs(nu.sum(1) - 1).max() < 1e-12
        assert ((nu != 0).sum(1) <= 4).all() and np.array_equal(nu, nv)
    assert abs(basis_function_one(3, uniform_knots(20, 3), 8, 0.5) - 1 / 48) < 1e-12


def test_shard_range_covers_everything_once():
    from parsenet_codebase_amd.dp import shard_range
    for n, w in ((32, 8), (10, 4), (3, 8)):
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))


def test_product_has_no_cpu_path():
    from parsenet_codebase_amd import graph
    from parsenet_codebase_amd.encoders import DGCNNEncoderGn
    from src.utils import chamfer_distance_single_shape
    with pytest.raises(RuntimeError):
        graph.knn(torch.zeros(1, 3, 32), 4)
    with pytest.raises(RuntimeError):
        DGCNNEncoderGn(mode=0)(torch.zeros(1, 3, 100))
    with pytest.raises(RuntimeError):
        chamfer_distance_single_shape(torch.zeros(4, 3), torch.zeros(5, 3))


def test_src_package_exposes_the_reference_names():
    import importlib
    names = {
        "src.model": ["knn", "get_graph_feature", "DGCNNControlPoints", "PrimitivesEmbeddingDGCNGn"],
        "src.PointNet": ["knn", "knn_points_normals", "get_graph_feature", "get_graph_feature_with_normals",
                         "DGCNNEncoderGn", "PrimitivesEmbeddingDGCNGn"],
        "src.mean_shift": ["MeanShift"],
        "src.segment_loss": ["EmbeddingLoss", "evaluate_miou", "primitive_loss"],
        "src.utils": ["chamfer_distance", "chamfer_distance_one_side", "chamfer_distance_single_shape",
                      "rescale_input_outputs", "grad_norm"],
        "src.loss": ["control_points_permute_reg_loss", "control_points_permute_closed_reg_loss",
                     "spline_reconstruction_loss_one_sided", "spline_reconstruction_loss",
                     "uniform_knot_bspline", "laplacian_loss", "basis_function_one"],
        "src.fitting_utils": ["LeastSquares", "best_lambda", "weights_normalize", "match", "customsvd",
                              "standardize_points_torch", "sample_points_from_control_points_", "to_one_hot"],
        "src.primitive_forward": ["forward_pass_open_spline", "forward_closed_splines", "Fit",
                                  "fit_one_shape_torch", "initialize_open_spline_model"],
        "src.fitting_optimization": ["FittingModule"],
        "src.primitives": ["ResidualLoss", "ComputePrimitiveDistance"],
        "src.residual_utils": ["Evaluation"],
        "src.approximation": ["fit_bezier_surface_fit_kronecker", "BSpline", "uniform_knot_bspline_"],
        "src.guard": ["guard_exp", "guard_sqrt"],
    }
    for mod, syms in names.items():
        m = importlib.import_module(mod)
        for s in syms:
            assert hasattr(m, s), (mod, s)


def test_state_dict_keys_match_the_reference_layout():
    from parsenet_codebase_amd.encoders import DGCNNControlPoints, PrimitivesEmbeddingDGCNGn
    net = PrimitivesEmbeddingDGCNGn(embedding=True, emb_size=128, primitives=True, num_primitives=10, mode=5,
                                    num_channels=6)
    keys = set(net.state_dict().keys())
    for k in ("encoder.conv1.0.weight", "encoder.conv1.1.weight", "encoder.bn1.weight", "encoder.bn4.weight",
              "encoder.bn5.bias", "encoder.mlp1.weight", "encoder.bnmlp1.weight", "conv1.weight", "bn2.bias",
              "mlp_seg_prob2.weight", "mlp_prim_prob2.bias", "bn_prim_prob1.weight"):
        assert k in keys, k
    assert sum(p.numel() for p in net.parameters()) == 1250442
    assert sum(p.numel() for p in DGCNNControlPoints(20, 10, 0).parameters()) == 3951152
    assert sum(p.numel() for p in DGCNNControlPoints(20, 10, 1).parameters()) == 4976816
    assert tuple(net.encoder.conv1[0].weight.shape) == (64, 12, 1, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from parsenet_codebase_amd.dp import FlatGradBucket, init_from_env
    r, w, dev = init_from_env(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    unused = torch.nn.Parameter(torch.zeros(5))            # like the reference's encoder.bn4/bn5
    params = list(model.parameters()) + [unused]
    bucket = FlatGradBucket(params)
    torch.manual_seed(100 + rank)                           # every rank sees its own shard
    x = torch.randn(6, 8)
    bucket.zero()
    model(x).pow(2).mean().backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(w)]
    dist.all_gather(gathered, local)
    expect = torch.stack(gathered).mean(0)
    ok = torch.allclose(bucket.flat, expect, atol=1e-7) and all(p.grad.data_ptr() >= bucket.flat.data_ptr()
                                                               for p in params)
    ok = ok and float(unused.grad.abs().sum()) == 0.0
    # the trainer's rank-consistent validation statistic: nan-aware mean over ranks
    from parsenet_codebase_amd.trainer import _mean_over_ranks
    m = _mean_over_ranks(1.0 + rank, dev)                   # (1 + 2) / 2
    n = _mean_over_ranks(float("nan") if rank == 0 else 4.0, dev)
    ok = ok and abs(m - 1.5) < 1e-12 and abs(n - 4.0) < 1e-12
    # one set of weights on every rank (the reference's DataParallel): rank-dependent init, then
    # sync_module_from_rank0 -> all ranks equal rank 0's parameters AND buffers
    from parsenet_codebase_amd.trainer import sync_module_from_rank0
    torch.manual_seed(500 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    net[1].running_mean.add_(rank + 1.0)
    sync_module_from_rank0(net)
    state = torch.cat([t.reshape(-1).double() for t in list(net.parameters()) + list(net.buffers())])
    both = [torch.zeros_like(state) for _ in range(w)]
    dist.all_gather(both, state)
    ok = ok and torch.equal(both[0], both[1]) and float(net[1].running_mean[0]) == 1.0
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` without a launcher environment starts two ranks as children
    (torch.distributed.run) and rank 0 reports the world size it observed."""
    import json
    import subprocess
    if torch.cuda.device_count() == 1:
        pytest.skip("a one-GPU box: rank 1 has no device (the launch path is covered on CPU ranks and by the driver)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["world_size_observed"] == 2
    # a launcher that started another number of ranks than --gpus is an error, not a warning
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode != 0


def test_bench_whole_control_flow_on_two_gloo_ranks():
    """bench.py's WHOLE control flow on two CPU ranks with the stub step: rank 0 pre-trains alone
    (gradient all-reduce off) and broadcasts, timed loop, the second (dense) timed run from the
    restored state, profiled steps, MAX over ranks, rank 0's single JSON line.  A rank mismatch or
    an unpaired collective anywhere in that sequence deadlocks here — not first on the 8-GPU node."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub",
                        "--steps", "6", "--warmup", "2", "--pretrain", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_observed"] == 2 and rec["steps"] == 6
    assert rec["value"] > 0 and rec["value_dense"] > 0 and rec["config"]["global_batch"] == 8
    # the same on one rank (no process group at all)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--steps", "3",
                        "--warmup", "1", "--pretrain", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["n_gpus"] == 1


def test_bench_forced_collective_on_one_gloo_rank():
    """PARSENET_FORCE_COLLECTIVE=1 with WORLD_SIZE=1: the process group is created, the rank-0 pre-training +
    broadcast, the status agreement, the gradient all-reduce and the barriers all run on the single rank (the CPU
    twin of tests/test_rccl_world1_gpu.py, which does the same with RCCL on the GPU box) and the line says so."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PARSENET_FORCE_COLLECTIVE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--steps", "3",
                        "--warmup", "1", "--pretrain", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["world_size_observed"] == 1 and rec["value"] > 0
    assert rec["config"]["collective"].startswith("forced on one rank")


def test_bench_step_is_dropped_on_every_rank_when_one_rank_raises():
    """bench.py --workload stub on two gloo ranks with a failure injected into rank 1's status check
    (PN_STUB_FAIL): the step is dropped on BOTH ranks — nobody enters the gradient all-reduce alone —
    and the run continues to its JSON line, which counts the skipped step."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PN_STUB_FAIL"] = "1:4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub",
                        "--steps", "6", "--warmup", "2", "--pretrain", "3"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["skipped_steps"] == 1 and rec["value"] > 0


def test_bench_control_flow_on_four_gloo_ranks():
    """bench.py --workload stub with FOUR ranks (the driver's N = 4 leg; 8 CPU ranks do not fit this
    container's test budget): pre-training on rank 0 + broadcast, timed loop, dense re-run, profiled steps,
    MAX over ranks, the CPU-baseline leg on rank 0 with the others at the barrier, one line with the
    per-rank spread."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "stub",
                        "--steps", "5", "--warmup", "2", "--pretrain", "3"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 4 and rec["world_size_observed"] == 4 and rec["value"] > 0
    assert rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp4"
    assert rec["cpu_baseline"] is not None and rec["cpu_baseline"]["kind"] == "port"
    pr = rec["per_rank_ms"]
    assert 0 < pr["min"] <= pr["max"] <= rec["ms_per_step"] * 1.001


def _status_worker(rank, w, port, out):
    import torch
    import torch.distributed as dist
    from parsenet_codebase_amd.dp import FitStatusError, FlatGradBucket
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=w)
    torch.manual_seed(0)
    net = torch.nn.Linear(3, 1, bias=False)
    bucket = FlatGradBucket(net.parameters())
    opt = torch.optim.SGD(net.parameters(), lr=0.5)
    w0 = net.weight.detach().clone()
    x = torch.ones(1, 3) * (rank + 1)

    def one(fail):
        bucket.zero()
        net(x).sum().backward()

        def finish():
            if fail:
                raise fail
            return "metrics"
        return bucket.finish_or_skip(finish, opt)
    res, err, took = one(FitStatusError("degenerate segment") if rank == 1 else None)   # rank 1 raises: BOTH drop the step
    ok = (not took) and res is None and torch.equal(net.weight.detach(), w0) and ((err is not None) == (rank == 1))
    res, err, took = one(None)                           # next step: mean over ranks, one optimizer move
    ok = ok and took and res == "metrics" and err is None
    ok = ok and torch.allclose(net.weight.detach(), w0 - 0.5 * torch.full((1, 3), 1.5))
    # anything that is NOT a fit status (out of memory, a launch error, a bug) goes through the agreement — the
    # other rank is not left in the gradient all-reduce — and is then re-raised on the rank it happened on
    w1 = net.weight.detach().clone()
    try:
        res, err, took = one(MemoryError("out of memory") if rank == 1 else None)
        ok = ok and rank == 0 and not took and err is None
    except MemoryError:
        ok = ok and rank == 1
    ok = ok and torch.equal(net.weight.detach(), w1)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_step_status_is_agreed_upon_before_the_gradient_all_reduce():
    """dp.FlatGradBucket.finish_or_skip on two gloo ranks (workloads.ParsenetE2EStep.step's tail): a
    fit status that raises on ONE rank drops the step on both, the next step reduces and moves; an error of
    any other kind also keeps the ranks in step, but propagates on the rank it happened on."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_status_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_pinned_ring_never_hands_out_a_slot_a_host_reader_still_holds():
    """_lib._PinnedRing: a download slot (hold=True) stays out of the rotation until released, however
    many uploads come by; when every slot is held the caller gets (None, None) and allocates."""
    from parsenet_codebase_amd import _lib

    class Ev:
        def synchronize(self):
            pass

        def record(self):
            pass
    ring = _lib._PinnedRing(slots=4, nbytes=64)
    ring.slots = [{"buf": torch.zeros(64, dtype=torch.uint8), "event": Ev(), "armed": False, "held": False, "gen": 0}
                  for _ in range(4)]
    _, held = ring.take(8, hold=True)
    seen = [ring.take(8)[1] for _ in range(40)]
    assert all(s.slot is not held.slot for s in seen) and len({id(s.slot) for s in seen}) == 3
    _lib._PinnedRing.release(held)
    assert any(ring.take(8)[1].slot is held.slot for _ in range(4))
    taken = [ring.take(8, hold=True)[1] for _ in range(4)]
    assert all(t is not None for t in taken) and ring.take(8) == (None, None)
    for t in taken:
        _lib._PinnedRing.release(t)
    assert ring.take(8)[1] is not None
    # a slot whose reader never comes back (an abandoned step) is reclaimed after ABANDONED_AFTER further takes
    _, lost = ring.take(8, hold=True)
    with pytest.warns(UserWarning):
        seen = [ring.take(8)[1] for _ in range(ring.ABANDONED_AFTER + 8)]
    assert any(s.slot is lost.slot for s in seen[-8:])
    assert all(s.slot is not lost.slot for s in seen[:ring.ABANDONED_AFTER - 4])


def test_a_late_reader_of_a_reclaimed_pinned_slot_fails_loudly_and_cannot_release_the_new_owner():
    """Round-5 advisor finding: the reader of a reclaimed slot was merely late.  Its handle carries the slot's
    generation: releasing it raises, and the hold of the slot's new owner stays in place."""
    from parsenet_codebase_amd import _lib

    class Ev:
        def synchronize(self):
            pass

        def record(self):
            pass
    ring = _lib._PinnedRing(slots=2, nbytes=64)
    ring.slots = [{"buf": torch.zeros(64, dtype=torch.uint8), "event": Ev(), "armed": False, "held": False, "gen": 0}
                  for _ in range(2)]
    _, late = ring.take(8, hold=True)
    with pytest.warns(UserWarning):
        for _ in range(ring.ABANDONED_AFTER + 2):
            ring.take(8)
    # the reclaimed slot goes to a new download
    owner = None
    for _ in range(4):
        _, h = ring.take(8, hold=True)
        if h.slot is late.slot:
            owner = h
            break
        _lib._PinnedRing.release(h)
    assert owner is not None
    with pytest.raises(RuntimeError, match="reclaimed"):
        _lib._PinnedRing.release(late)
    assert owner.slot["held"]                        # still the new owner's
    _lib._PinnedRing.release(owner)
    assert not owner.slot["held"]


def test_wait_event_spins_for_a_bounded_time_then_blocks(monkeypatch):
    from parsenet_codebase_amd import _lib

    class Ev:
        def __init__(self, ready_after):
            self.n, self.ready_after, self.blocked = 0, ready_after, False

        def query(self):
            self.n += 1
            return self.n > self.ready_after

        def synchronize(self):
            self.blocked = True
    monkeypatch.setattr(_lib, "_SPIN_SECONDS", 1e-3)
    quick, never = Ev(3), Ev(10 ** 12)
    _lib.wait_event(quick)
    _lib.wait_event(never)
    assert not quick.blocked and never.blocked


def _pretrain_worker(rank, w, port, out):
    import torch
    import torch.distributed as dist
    from parsenet_codebase_amd.dp import FlatGradBucket
    from parsenet_codebase_amd.workloads import train_on_rank0_then_broadcast
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=w)
    torch.manual_seed(rank)                       # different weights per rank before
    net = torch.nn.Linear(3, 2)
    bucket = FlatGradBucket(net.parameters())
    ran = []

    def train():
        ran.append(rank)
        for _ in range(3):                        # steps with the bucket's reduction call in them, like seg_step
            bucket.zero()
            net(torch.ones(1, 3)).sum().backward()
            bucket.all_reduce_mean()              # must NOT be a collective here: rank 1 is not in these steps
            with torch.no_grad():
                net.weight -= 0.1 * net.weight.grad
    train_on_rank0_then_broadcast(net, bucket, train)
    state = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.zeros_like(state) for _ in range(w)]
    dist.all_gather(both, state)
    out[rank] = bool(torch.equal(both[0], both[1])) and ran == ([0] if rank == 0 else []) and bucket.collective
    dist.destroy_process_group()


def test_pretraining_runs_on_rank0_only_and_is_broadcast():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pretrain_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]


def _accum_worker(rank, w, port, out):
    import torch
    import torch.distributed as dist
    from parsenet_codebase_amd.dp import FlatGradBucket
    from parsenet_codebase_amd.trainer import accumulate_or_skip, guarded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=w)
    torch.manual_seed(0)
    net = torch.nn.Linear(3, 1, bias=False)
    bucket = FlatGradBucket(net.parameters())
    opt = torch.optim.SGD(net.parameters(), lr=0.5)
    data = torch.arange(30, dtype=torch.float32).reshape(10, 3) * (rank + 1)     # rank-specific micro-batches
    w0 = net.weight.detach().clone()
    seen = []

    def micro_factory(first, fail_at):
        def fitting_loss(i):
            if i == fail_at:
                raise RuntimeError("degenerate segment")

        def micro(i):
            guarded(fitting_loss, i)                # what the reference's try covers
            net(data[first + i:first + i + 1]).sum().backward()
        return micro
    # step 1: rank 1 fails in its third micro-batch -> EVERY rank drops the step, no optimizer move
    took = accumulate_or_skip(bucket, opt, 5, micro_factory(0, 2 if rank == 1 else -1), w, None,
                              lambda m, f: seen.append(f.clone()), net)
    ok = (not took) and torch.equal(net.weight.detach(), w0) and not seen
    ok = ok and float(bucket.flat.abs().sum()) > 0          # the partial sums are still there ...
    # step 2: five micro-batches on both ranks -> sum over micro-batches, mean over ranks, one step
    took = accumulate_or_skip(bucket, opt, 5, micro_factory(5, -1), w, None, lambda m, f: seen.append(f.clone()), net)
    want = sum(data[5:10].sum(0) * s for s in (1.0,)) * (1 + 2) / (rank + 1) / 2.0   # (g_rank0 + g_rank1) / 2
    ok = ok and took and len(seen) == 1 and torch.allclose(seen[0], want.reshape(-1))     # ... and were zeroed
    ok = ok and torch.allclose(net.weight.detach().reshape(-1), w0.reshape(-1) - 0.5 * want)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_only_the_guarded_part_of_a_micro_batch_can_skip_a_step():
    """trainer.accumulate_or_skip drops a step on StepSkipped (fitting loss / backward, like the
    reference's try at train_parsenet_e2e.py:232-257) and lets everything else through: an exhausted
    data iterator or a programming error must not be recorded as a "mistake" step after step."""
    import torch
    from parsenet_codebase_amd.dp import FlatGradBucket
    from parsenet_codebase_amd.trainer import accumulate_or_skip, guarded
    net = torch.nn.Linear(2, 1)
    bucket, opt = FlatGradBucket(net.parameters()), torch.optim.SGD(net.parameters(), lr=0.1)
    it = iter([torch.ones(1, 2)])
    seen = []

    def micro(i):
        x = next(it)                                   # StopIteration in the second micro-batch
        guarded(lambda: net(x).sum().backward())
    with pytest.raises((StopIteration, RuntimeError)):
        accumulate_or_skip(bucket, opt, 2, micro)

    def bad(i):
        guarded(lambda: (_ for _ in ()).throw(ValueError("no full-rank ridge system")))
    assert accumulate_or_skip(bucket, opt, 2, bad, on_exception=seen.append) is False
    assert seen and "no full-rank ridge system" in seen[0]


def test_e2e_accumulation_skip_and_rank_mean_on_two_gloo_ranks():
    """train_parsenet_e2e's step logic (trainer.accumulate_or_skip; train_parsenet_e2e.py:174-277)
    on two ranks: an exception in micro-batch 3 of ONE rank drops the step on BOTH (weights
    untouched, nobody enters the gradient all-reduce alone); the next step accumulates five
    micro-batches per rank, averages over the ranks and moves the weights once."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_accum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_flat_gradient_bucket_allreduce_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_plateau_scheduler_matches_torch():
    """trainer.ReduceLROnPlateau reproduces torch.optim.lr_scheduler.ReduceLROnPlateau (mode min,
    relative threshold 1e-4) on a noisy plateau."""
    import torch
    from parsenet_codebase_amd.trainer import ReduceLROnPlateau
    rng = np.random.RandomState(0)
    metrics = np.concatenate([np.linspace(1, 0.5, 6), 0.5 + 0.01 * rng.rand(40)])
    p1, p2 = [torch.nn.Parameter(torch.zeros(1))], [torch.nn.Parameter(torch.zeros(1))]
    o1, o2 = torch.optim.Adam(p1, lr=1e-2), torch.optim.Adam(p2, lr=1e-2)
    mine = ReduceLROnPlateau(o1, factor=0.5, patience=4, min_lr=1e-4)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(o2, mode="min", factor=0.5, patience=4, min_lr=1e-4)
    for m in metrics:
        mine.step(m)
        ref.step(m)
        assert abs(o1.param_groups[0]["lr"] - o2.param_groups[0]["lr"]) < 1e-12
    assert o1.param_groups[0]["lr"] < 1e-2


def test_train_config_reads_reference_format(tmp_path):
    from parsenet_codebase_amd.trainer import TrainConfig
    text = """comment=""

[train]
model_path = "train_parsenet_e2e_{}"
pretrain_model_path = "parsenet_with_normals.pth"
normals = True
num_train=24000
num_val=4000
num_test=4000
num_points=1000
loss_weight=100
num_epochs = 100
batch_size = 1
# Learing rate
lr = 0.0001
patience = 8
mode = 5
"""
    f = tmp_path / "c.yml"
    f.write_text(text)
    c = TrainConfig.from_file(str(f))
    assert (c.num_train, c.num_test, c.epochs, c.batch_size, c.mode, c.patience) == (24000, 4000, 100, 1, 5, 8)
    assert c.lr == 1e-4 and c.loss_weight == 100.0 and c.normals is True
    assert c.model_path == "train_parsenet_e2e_{}" and c.pretrain_model_path == "parsenet_with_normals.pth"


def test_checkpoint_names_of_the_reference_config_templates():
    """The ``model_path`` templates of the reference's configs/*.yml carry 6 (segmentation, e2e) or 8
    (SplineNet) fields, filled as its scripts do (train_parsenet.py:28-35, train_open_splines.py:33-42)."""
    from parsenet_codebase_amd.trainer import TrainConfig, model_name
    c = TrainConfig(batch_size=2, lr=0.01, num_train=24000, num_test=4000, loss_weight=100.0, mode=5)
    c.model_path = "train_parsenet_{}_lr_{}_trsz_{}_tsz_{}_wght_{}_mode_{}"
    assert model_name(c, "seg") == "train_parsenet_2_lr_0.01_trsz_24000_tsz_4000_wght_100.0_mode_5"
    c.model_path = "train_parsenet_e2e_{}_lr_{}_trsz_{}_tsz_{}_wght_{}_mode_{}"
    assert model_name(c, "seg").startswith("train_parsenet_e2e_2_lr_0.01_")
    s = TrainConfig(batch_size=36, lr=0.001, num_train=3200, num_test=3000, loss_weight=0.9, mode=0, num_points=700)
    s.model_path = "temp_{}_{}_{}_bt_{}_lr_{}_trsz_{}_tsz_{}_wght_{}"
    assert model_name(s, "spline") == "temp_0_700_0.9_bt_36_lr_0.001_trsz_3200_tsz_3000_wght_0.9"
    s.model_path = "train_closed_spline_{}_{}_{}_bt_{}_lr_{}_trsz_{}_tsz_{}_wght_{}"
    assert model_name(s, "spline").startswith("train_closed_spline_0_700_0.9_bt_36_")
    s.model_path = "one_field_{}"
    assert model_name(s, "spline") == "one_field_0" and model_name(s, "seg") == "one_field_36"


def test_host_iou_matrix_matches_one_hot_products_on_cpu():
    """fitting._relaxed_iou_of_labels (host bincount form) == relaxed_iou_fast on one-hot
    encodings (the reference's form), evaluated here with torch on the CPU: bit for bit."""
    from parsenet_codebase_amd.fitting import _relaxed_iou_of_labels, relaxed_iou_fast
    rng = np.random.RandomState(11)
    for n, k in ((10000, 9), (5000, 49)):
        gt = rng.randint(0, k, n)
        pred = (gt + (rng.rand(n) < 0.3) * rng.randint(0, k, n)) % k
        oh = lambda a: torch.zeros(n, 50).scatter_(1, torch.from_numpy(a).long().unsqueeze(1), 1)  # noqa: E731
        want = relaxed_iou_fast(oh(pred).unsqueeze(0), oh(gt).unsqueeze(0))[0].numpy()
        got = _relaxed_iou_of_labels(pred, gt)
        assert got.dtype == np.float32 and np.array_equal(got, want)
    with pytest.raises(ValueError):
        _relaxed_iou_of_labels(np.array([0, 50]), np.array([0, 1]))


def _device_dataset_equals_host(device, atol):
    """data.Dataset(device=...) keeps the split on the device and does the per-batch work there
    (gather, augmentation map, normal noise, canonical frame); the draws come from numpy's
    generator in the same order, so the batches equal the host mode's to rounding — of the maps
    (1e-7) and of the 3 x 3 second-moment matrix whose eigenvector is the canonical axis (fp32
    accumulation in another order turns the axis by ~1e-5 rad on these shapes; the reference's own frame has the
    same sensitivity)."""
    import torch
    from parsenet_codebase_amd import data as D, synthetic
    pts, nrm, lab, prim = synthetic.make_batch(3, 6, 1500)
    raw = {"points": pts, "normals": nrm, "labels": lab, "prim": prim}
    kw = dict(train=dict(raw), val=dict(raw), train_size=6, val_size=6, normals=True, primitives=True)
    host, dev = D.Dataset(2, **kw), D.Dataset(2, device=device, **kw)
    for seed, flags in ((5, dict(randomize=True, augment=True, align_canonical=True, anisotropic=False,
                                 if_normal_noise=True)),
                        (6, dict(randomize=True, augment=True, align_canonical=True, anisotropic=True,
                                 if_normal_noise=False))):
        np.random.seed(seed)
        gh = host.get_train(**flags)
        want = [next(gh) for _ in range(5)]
        state_h = np.random.get_state()[1][:4].copy()
        np.random.seed(seed)
        gd = dev.get_train(**flags)
        got = [next(gd) for _ in range(5)]
        assert np.array_equal(np.random.get_state()[1][:4], state_h)           # same draws consumed
        for (p0, l0, n0, t0), (p1, l1, n1, t1) in zip(want, got):
            assert isinstance(p1, torch.Tensor) and p1.device.type == torch.device(device).type
            assert np.allclose(p1.cpu().numpy(), p0, rtol=0, atol=atol)
            assert np.allclose(n1.cpu().numpy(), n0, rtol=0, atol=atol)
            assert np.array_equal(l0, l1) and np.array_equal(t0, t1)


def test_device_resident_dataset_on_the_cpu_device():
    _device_dataset_equals_host("cpu", 3e-5)


def test_batched_host_rotations_equal_the_per_segment_function_bit_for_bit():
    """fitting_batch.host_minor_axis_rotations (one batched geev call, cross / dot products of
    rotation_matrix_a_to_b written out for B = e_x) against the per-matrix function whose bits the
    spline fixtures pin (fitting_utils.py:532-577): the float32 bits of R decide kNN near-ties of
    the SplineNets downstream, so the fast path must not move a single one."""
    import torch
    from parsenet_codebase_amd.fitting_batch import _host_minor_axis_rotation, host_minor_axis_rotations
    torch.manual_seed(0)
    for trial in range(6):
        c = torch.randn(48, int(torch.randint(20, 3000, (1,))), 3) * torch.rand(48, 1, 3)
        c = c - c.mean(1, keepdim=True)
        cov = torch.bmm(c.transpose(1, 2), c)
        cov[0] = torch.eye(3)                                     # degenerate: singular basis -> identity
        cov[1] = torch.diag(torch.tensor([1.0, 2.0, 3.0]))        # minor axis already +x
        cov[2] = torch.diag(torch.tensor([3.0, 2.0, 1.0]))
        a = np.stack([_host_minor_axis_rotation(cov[s]) for s in range(48)])
        b = host_minor_axis_rotations(cov)
        assert a.dtype == b.dtype == np.float32 and np.array_equal(a.view(np.int32), b.view(np.int32))


def test_auto_mode_of_the_block_sparse_iterations():
    """mean_shift.use_sparse / auto_report: plan AUTO_SAMPLES calls; when their plans kept more than
    AUTO_DENSE_ABOVE of the tile pairs on average launch dense for AUTO_DENSE_STEPS calls, then
    probe again; PARSENET_MS_SPARSE=1 / 0 (SPARSE True / False) override."""
    from parsenet_codebase_amd import mean_shift as MSM
    saved, MSM.SPARSE = MSM.SPARSE, "auto"
    MSM._AUTO.clear()
    try:
        key = (4, 10000)
        for share in (0.99, 0.97, 0.98):                 # one dense-looking batch decides nothing
            assert MSM.use_sparse(*key)
            MSM.auto_report(*key, share)
        assert MSM.use_sparse(*key)
        MSM.auto_report(*key, 0.96)                      # four in a row: a converged embedding
        seq = [MSM.use_sparse(*key) for _ in range(MSM.AUTO_DENSE_STEPS + 1)]
        assert seq == [False] * MSM.AUTO_DENSE_STEPS + [True]
        for share in (0.95, 0.7, 0.8, 0.75):             # mean 0.8: keep planning
            MSM.auto_report(*key, share)
            assert MSM.use_sparse(*key)
        assert MSM.use_sparse(2, 5000)                   # another problem size has its own memory
        MSM.SPARSE = False
        assert not MSM.use_sparse(*key)
        MSM.SPARSE = True
        MSM.auto_report(*key, 0.99)                      # ignored outside auto mode
        assert MSM.use_sparse(*key)
    finally:
        MSM.SPARSE = saved
        MSM._AUTO.clear()


def test_workload_pool_rotation_on_the_host():
    """ParsenetSegStep: the resident pool and the order in which steps walk through it (shapes
    [s * batch mod pool, + batch)); explicit shape ids; a pool that is not a multiple of the batch."""
    import torch
    from parsenet_codebase_amd import synthetic, workloads
    step = workloads.ParsenetSegStep(torch.device("cpu"), batch=2, num_points=1500, nn_nb=8, first_shape=5, pool=6)
    want = synthetic.make_batch(5, 6, 1500)
    seen = []
    for _ in range(4):
        step.next_batch()
        seen.append(step.cursor)
        lo = (len(seen) - 1) * 2 % 6
        assert np.array_equal(step.points.numpy(), want[0][lo:lo + 2]) and np.array_equal(step.labels, want[2][lo:lo + 2])
        assert tuple(step.x.shape) == (2, 6, 1500) and step.prim.dtype == torch.int64
    assert seen == [2, 4, 0, 2]
    ids = [68, 160]
    step = workloads.ParsenetSegStep(torch.device("cpu"), batch=2, num_points=1500, nn_nb=8, shape_ids=ids)
    assert step.pool == 2 and np.array_equal(step.labels, synthetic.make_batch_ids(ids, 1500)[2])
    with pytest.raises(ValueError):
        workloads.ParsenetSegStep(torch.device("cpu"), batch=4, num_points=1500, nn_nb=8, pool=6)


def test_host_thread_cap_and_usable_cpus(monkeypatch):
    """dp.limit_host_threads: one intra-op thread by default (the GPU legs compute nothing on the CPU; the
    default pool gets the process throttled under a CFS quota), PARSENET_HOST_THREADS overrides, 0 leaves
    torch alone; dp.usable_cpus never exceeds the visible cores and honours the cgroup quota."""
    import torch
    from parsenet_codebase_amd import dp
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv("PARSENET_HOST_THREADS", raising=False)
        assert dp.limit_host_threads() == before
        assert torch.get_num_threads() == 1
        monkeypatch.setenv("PARSENET_HOST_THREADS", "2")
        dp.limit_host_threads()
        assert torch.get_num_threads() == 2
        monkeypatch.setenv("PARSENET_HOST_THREADS", "0")
        dp.limit_host_threads()
        assert torch.get_num_threads() == 2          # 0: untouched
        dp.limit_host_threads(3)
        assert torch.get_num_threads() == 3          # an explicit count wins
    finally:
        torch.set_num_threads(before)
    n = dp.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            assert n <= max(1, int(float(q) / float(p)))
    except OSError:
        pass


def test_segmentation_metric_helpers():
    """metrics.* (src/segment_utils.py helpers): host numpy, checked on hand-made label sets."""
    from parsenet_codebase_amd import metrics as M
    gt = np.array([0, 0, 1, 1, 2, 2, 2, 5])
    pred = np.array([0, 1, 1, 1, 2, 2, 0, 5])
    # classes 0..5: IoUs 1/3, 2/3, 2/3, 1 (empty), 1 (empty), 1
    assert abs(M.mean_IOU_one_sample(pred, gt, 6) - (1 / 3 + 2 / 3 + 2 / 3 + 3) / 6) < 1e-6
    a, b = np.array([0, 6, 7, 8, 1]), np.array([9, 9, 9, 2, 1])
    assert abs(M.iou_segmentation(a.copy(), b.copy()) - 1.0) < 1e-6 and a[0] == 0      # arguments untouched
    perm = np.array([2, 0, 1, 3, 4, 5])
    assert abs(M.SIOU(gt, perm[gt]) - 1.0) < 1e-6                                      # a relabelling is perfect
    assert 0.3 < M.SIOU(gt, pred) < 1.0
    p = np.eye(3)[[0, 1, 2, 2]]
    w = np.array([[1.0, 0], [1, 0], [0, 1], [0, 1]])
    assert list(M.primitive_type_segment(p, w)) == [0, 2]
    assert list(M.primitive_type_segment_torch(torch.from_numpy(p), torch.from_numpy(w)).numpy()) == [0, 2]
    emb = np.eye(4)[:3]
    cen = np.eye(4)[:2]
    pm = M.cluster_prob_mutual(emb, cen, 0.5)
    assert pm.shape == (2, 3) and np.allclose(pm.sum(0), 1.0)
    assert M.cluster_prob(emb, cen, 0.5).shape == (2, 3)


# ---------------------------------------------------------------------------------------------------------------
# Hungarian matching: which optimum among EQUAL-cost assignments (src/fitting_utils.py:362-376 uses
# lapsolver.solve_dense, absent here; scipy's linear_sum_assignment stands in — SURVEY 8c "tie order unpinned")
# ---------------------------------------------------------------------------------------------------------------
def _shuffled_solver(rng):
    """An equally exact assignment solver with ANOTHER tie order: scipy on a row- and column-permuted matrix,
    mapped back.  Every optimum of a tied cost matrix is reachable this way."""
    from scipy.optimize import linear_sum_assignment

    def solve(cost):
        cost = np.asarray(cost)
        pr, pc = rng.permutation(cost.shape[0]), rng.permutation(cost.shape[1])
        r, c = linear_sum_assignment(cost[pr][:, pc])
        rows, cols = pr[r], pc[c]
        o = np.argsort(rows)
        return rows[o], cols[o]
    return solve


@pytest.mark.parametrize("n_pred,n_gt", [(5, 9), (9, 5), (7, 7), (1, 6), (12, 3)])
def test_matching_results_do_not_depend_on_the_tie_order_of_the_assignment_solver(monkeypatch, n_pred, n_gt):
    """The 50 x 50 relaxed-IoU matrix of a shape is mostly ties: every empty one-hot column / row costs 1.0 against
    everything, and so does a predicted cluster against a ground-truth segment it does not touch.  Whatever optimum
    the solver returns, the things the loss is built from are the same: the segments that get fitted with their
    ground-truth partners and types (training: fitting_batch.build_segment_table; evaluation:
    fitting_eval.eval_segments), and the segment IoU / type accuracy of the matched pairs."""
    from parsenet_codebase_amd import fitting, fitting_batch as FB, fitting_eval as FE
    rng = np.random.RandomState(100 * n_pred + n_gt)
    N = 6000
    gt = rng.randint(0, n_gt, N)
    prim_of_gt = rng.choice([1, 3, 4, 5, 2, 8, 0, 6], n_gt)
    prim = prim_of_gt[gt]
    # predicted clusters: a noisy relabelling with another number of clusters (merges / splits / strays)
    pred = (rng.permutation(50)[:n_gt][gt] % n_pred + (rng.rand(N) < 0.15) * rng.randint(0, n_pred, N)) % n_pred
    prim_pred = np.where(rng.rand(N) < 0.9, prim, rng.randint(0, 10, N))
    ref_segs = ref_metrics = ref_eval = None
    for trial in range(25):
        solver = fitting.solve_dense if trial == 0 else _shuffled_solver(np.random.RandomState(trial))
        monkeypatch.setattr(fitting, "solve_dense", solver)
        segs, match = FB.build_segment_table(gt, prim, pred, N)
        segs = [(s["row"], s["key"], s["type"], s["kind"], s["gt"].tobytes()) for s in segs]
        modal_pred = np.asarray([np.bincount(prim_pred[pred == c], minlength=10).argmax() if (pred == c).any() else 0
                                 for c in range(50)])
        siou, pacc, _, pairs = FB.siou_matched_segments_fast(match, modal_pred, prim)
        esegs, _ = FE.eval_segments(gt, pred, prim_pred)
        esegs = [(s["index"], s["key"], s["type"], s["kind"], s["pred"].tobytes(), s["gt"].tobytes(), float(s["wv"]),
                  s["fit"]) for s in esegs]
        metrics = (siou, pacc, sorted(map(tuple, pairs)))
        if trial == 0:
            ref_segs, ref_metrics, ref_eval = segs, metrics, esegs
            assert segs                                  # something IS fitted
        else:
            assert segs == ref_segs and esegs == ref_eval
            assert metrics[0] == ref_metrics[0] and metrics[1] == ref_metrics[1] and metrics[2] == ref_metrics[2]


def test_a_genuine_tie_between_occupied_segments_keeps_the_matched_iou(monkeypatch):
    """The one tie that CAN change the pairing: two predicted clusters that each cover exactly half of two equally
    large ground-truth segments (equal integer counts — a measure-zero event on real clusterings).  Both pairings are
    optima; the matched segment IoU (s_iou) is the same under either, the total cost is the optimum of all 2 x 2
    pairings, and scipy's choice is one of the two — the reference's lapsolver may return the other one, which is why
    SURVEY 8c calls the tie order unpinned."""
    from parsenet_codebase_amd import fitting, fitting_batch as FB
    gt = np.repeat([0, 1, 2], [400, 400, 300])
    pred = np.concatenate([np.tile([0, 1], 200), np.tile([1, 0], 200), np.full(300, 2)])
    prim = np.repeat([1, 1, 5], [400, 400, 300])
    seen = set()
    sious = set()
    for trial in range(40):
        solver = fitting.solve_dense if trial == 0 else _shuffled_solver(np.random.RandomState(trial))
        monkeypatch.setattr(fitting, "solve_dense", solver)
        segs, match = FB.build_segment_table(gt, prim, pred, gt.size)
        rids, cids = match[0], match[1]
        seen.add((int(cids[0]), int(cids[1]), int(cids[2])))
        sious.add(FB.siou_matched_segments_fast(match, np.zeros(50, np.int64), prim)[0])
        assert [s["kind"] for s in segs] == ["prim"] * 3
    assert seen == {(0, 1, 2), (1, 0, 2)}               # both optima occur, nothing else
    assert len(sious) == 1                               # the matched IoU does not depend on which


def test_edge_weight_function_matches_the_autograd_composition_bit_for_bit():
    """graph._EdgeWeight: [Wa | Wb] -> [Wa ; Wb - Wa]^T with a two-launch backward pass against slicing, subtracting
    and concatenating through autograd (ten launches per layer): operand and weight gradient identical."""
    from parsenet_codebase_amd.graph import _EdgeWeight
    g = torch.Generator().manual_seed(5)
    for Cout, C in ((64, 3), (64, 64), (128, 64)):
        w0 = torch.randn(Cout, 2 * C, generator=g)
        up = torch.randn(C, 2 * Cout, generator=g)
        wa = w0.clone().requires_grad_(True)
        ref = torch.cat([wa[:, :C], wa[:, C:] - wa[:, :C]], 0).t()
        (ref * up).sum().backward()
        wb = w0.clone().requires_grad_(True)
        got = _EdgeWeight.apply(wb, C)
        (got * up).sum().backward()
        assert torch.equal(got, ref) and torch.equal(wb.grad, wa.grad)


def test_shared_normalisation_of_the_embedding():
    """losses.normalized_rows: the second normalisation of the SAME view of the same tensor returns the first
    one's result (one autograd node); another tensor, another autograd mode, an in-place write or a freed result
    give a fresh evaluation; gradients through the shared node equal the sum of two separate normalisations
    within rounding."""
    import torch.nn.functional as F
    from parsenet_codebase_amd import losses as L
    old_default, L.SHARE_NORMALIZE = L.SHARE_NORMALIZE, True
    try:
        _shared_normalisation_cases(L, F)
    finally:
        L.SHARE_NORMALIZE = old_default


def _shared_normalisation_cases(L, F):
    e = torch.randn(2, 16, 50, requires_grad=True)
    y = e * 1.0
    a = L.normalized_rows(y.permute(0, 2, 1))
    b = L.normalized_rows(y.permute(0, 2, 1))
    assert a is b and torch.equal(a, F.normalize(y.permute(0, 2, 1), p=2, dim=2))
    w1, w2 = torch.randn(2, 50, 16), torch.randn(2, 50, 16)
    ((a * w1).sum() + (b * w2).sum()).backward()
    shared = e.grad.clone()
    e.grad = None
    y2 = e * 1.0
    c = L.normalized_rows(y2.permute(0, 2, 1))
    assert c is not a                                            # another tensor
    ((F.normalize(y2.permute(0, 2, 1), p=2, dim=2) * w1).sum() + (c * w2).sum()).backward()
    assert float((e.grad - shared).abs().max()) < 1e-6
    with torch.no_grad():
        d = L.normalized_rows(y2.permute(0, 2, 1))
    assert d is not c and not d.requires_grad                    # another autograd mode
    y3 = torch.randn(2, 16, 50)
    f = L.normalized_rows(y3.permute(0, 2, 1))
    assert L.normalized_rows(y3.permute(0, 2, 1)) is f
    y3.add_(1.0)                                                 # an in-place write: a fresh evaluation
    g = L.normalized_rows(y3.permute(0, 2, 1))
    assert g is not f and torch.equal(g, F.normalize(y3.permute(0, 2, 1), p=2, dim=2))
    assert L.normalized_rows(y3) is not g                        # another view geometry of the same tensor
    del f, g
    L.SHARE_NORMALIZE = False
    h = L.normalized_rows(y3.permute(0, 2, 1))
    assert L.normalized_rows(y3.permute(0, 2, 1)) is not h
