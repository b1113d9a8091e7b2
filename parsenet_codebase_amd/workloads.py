"""The BASELINE.json configurations as callable training steps on synthetic data (used by
bench.py, __graft_entry__.smoke() and the tests).  Everything a step touches is resident on
the GPU before the timed region starts."""
import functools
import operator
import os

import numpy as np
import torch

from . import synthetic
from .dp import FlatGradBucket
from .encoders import PrimitivesEmbeddingDGCNGn
from .losses import EmbeddingLoss, primitive_loss


FLAT_ADAM = os.environ.get("PARSENET_FLAT_ADAM", "1") != "0"


def _adam(params, lr, bucket=None):
    """Adam with torch's defaults.  With the model's gradient bucket on the GPU: optim.FlatAdam — parameters,
    gradients and moments in flat buffers, ONE launch per step (round 6).  PARSENET_FLAT_ADAM=0 (developer A/B) or
    no bucket: torch.optim.Adam with the multi-tensor fused kernel."""
    params = list(params)
    if FLAT_ADAM and bucket is not None and params and params[0].is_cuda:
        from .optim import FlatAdam
        return FlatAdam(bucket, lr=lr)
    return torch.optim.Adam(params, lr=lr, fused=bool(params and params[0].is_cuda))


def train_on_rank0_then_broadcast(model, bucket, train):
    """Every rank must start the timed region from ONE set of weights.  Rank 0 runs ``train()``
    alone (the bucket's gradient all-reduce is switched off meanwhile: the other ranks are not
    in those steps), then all parameters and buffers reach the other ranks with one flat
    broadcast per dtype (trainer.sync_module_from_rank0).  On a single rank: just ``train()``."""
    import torch.distributed as dist
    from .trainer import sync_module_from_rank0
    from .dp import multi_rank
    multi = multi_rank()
    if not multi:
        return train()
    if dist.get_rank() == 0:
        bucket.collective = False
        try:
            train()
        finally:
            bucket.collective = True
    sync_module_from_rank0(model)


PRETRAIN_FIRST_SHAPE = 1000000     # ids of the pre-training shapes: disjoint from every timed pool


class ParsenetSegStep:
    """cfg4: ParSeNet segmentation-only training step (train_parsenet.py:151-198): points +
    normals (6 channels), first graph on the points+normals metric, k = 80, triplet embedding
    loss + NLL primitive loss, forward + backward + one gradient all-reduce + Adam.

    ``pool`` (a multiple of ``batch``, default = ``batch``): number of distinct shapes resident in
    HBM; step s works on shapes [s * batch mod pool, + batch) of the pool — the reference's loop
    draws a new batch every iteration (train_parsenet.py:151-160)."""

    def __init__(self, device, batch=4, num_points=10000, nn_nb=80, first_shape=0, seed=0, lr=1e-2, pool=None,
                 shape_ids=None):
        torch.manual_seed(seed)
        self.device = device
        self.batch = batch
        self.num_points = num_points
        self.loss = EmbeddingLoss(margin=1.0, if_mean_shift=False)
        self.model = PrimitivesEmbeddingDGCNGn(embedding=True, emb_size=128, primitives=True,
                                               num_primitives=10, loss_function=self.loss.triplet_loss,
                                               mode=5, num_channels=6, nn_nb=nn_nb).to(device)
        self.bucket = FlatGradBucket(self.model.parameters())
        self.opt = _adam(self.model.parameters(), lr, self.bucket)
        self.rng_seed = seed
        self.shape_ids = None if shape_ids is None else list(shape_ids)
        if shape_ids is not None:
            self.load_pool(first_shape, len(shape_ids), ids=shape_ids)
        else:
            self.load_pool(first_shape, batch if pool is None else pool)

    def load_pool(self, first_shape, pool, ids=None):
        """Make shapes first_shape .. first_shape + pool - 1 (or the explicit ``ids``) resident
        (everything a step touches is in HBM before the timed region starts; labels stay host
        integers like the reference's numpy)."""
        if pool % self.batch:
            raise ValueError("pool (%d) must be a multiple of the batch (%d)" % (pool, self.batch))
        pts, nrm, lab, prim = (synthetic.make_batch_ids(ids, self.num_points) if ids is not None else
                               synthetic.make_batch(first_shape, pool, self.num_points))
        x = np.concatenate([pts, nrm], 2).transpose(0, 2, 1)           # (P,6,N)
        self.pool, self.first_shape = pool, first_shape
        self.pool_x = torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        self.pool_points = torch.from_numpy(pts).to(self.device)
        self.pool_normals = torch.from_numpy(nrm).to(self.device)
        self.pool_labels = lab                                          # host ints (reference: numpy)
        self.pool_prim_np = prim
        self.pool_prim = torch.from_numpy(prim).to(self.device)
        self.cursor = 0
        self.select(0)

    def select(self, start):
        """Views of the batch that starts at pool position ``start`` (no copies)."""
        sl = slice(start, start + self.batch)
        self.x, self.labels, self.prim = self.pool_x[sl], self.pool_labels[sl], self.pool_prim[sl]
        self.points, self.normals, self.prim_np = self.pool_points[sl], self.pool_normals[sl], self.pool_prim_np[sl]

    def next_batch(self):
        self.select(self.cursor)
        self.cursor = (self.cursor + self.batch) % self.pool

    def shapes_per_step(self):
        return self.batch

    def seg_step(self):
        self.next_batch()
        self.bucket.begin()          # one backward pass: gradients are handed over, then gathered (dp.FlatGradBucket)
        embedding, log_prob, embed_loss = self.model(self.x, self.labels, True)
        loss = torch.mean(embed_loss) + primitive_loss(log_prob, self.prim)
        loss.backward()
        self.bucket.gather()
        self.bucket.all_reduce_mean()
        self.opt.step()
        return loss

    def step(self):
        return self.seg_step()


class ParsenetE2EStep(ParsenetSegStep):
    """cfg5: the end-to-end step of train_parsenet_e2e.py:190-277 per shape — segmentation
    network (norm layers frozen: model.eval()), then for every shape of the batch mean-shift
    clustering of the embedding (quantile 0.025, 10 iterations), Hungarian matching, weighted
    primitive / SplineNet fits and residual losses (lamb 0.1); loss = triplet + NLL + residual,
    backward through everything, one gradient all-reduce, Adam.  The SplineNets are frozen
    random-init DGCNNControlPoints (no pretrained weights ship with the reference).

    Initial state.  The reference starts end-to-end training from a segmentation network
    pre-trained by train_parsenet.py (train_parsenet_e2e.py:82-84 loads parsenet_with_normals.pth)
    and draws a new shape every iteration (:190-241).  No checkpoint ships with it, so the
    stand-in is ``pretrain_steps`` deterministic segmentation-only steps (triplet + NLL, Adam
    ``pretrain_lr``) over ``pretrain_pool`` shapes with ids from PRETRAIN_FIRST_SHAPE on — DISJOINT
    from the ``pool`` shapes the end-to-end steps then rotate through (held-out data, like the
    reference's fresh batches).  ``pretrain_pool=None`` keeps the round-2 behaviour (pre-training
    on the step's own batch; tests and tools that want an over-fitted embedding).  With several
    ranks only rank 0 trains; the weights reach the others with one flat broadcast."""

    def __init__(self, device, batch=4, num_points=10000, nn_nb=80, first_shape=0, seed=0, lr=1e-4,
                 pretrain_steps=0, pool=None, pretrain_pool=None, pretrain_lr=1e-2, shape_ids=None):
        super().__init__(device, batch, num_points, nn_nb, first_shape, seed, lr=pretrain_lr, pool=pool,
                         shape_ids=shape_ids)
        from .encoders import DGCNNControlPoints
        from .fitting import Evaluation
        self.pretrain_steps = int(pretrain_steps)
        self.pretrain_pool = pretrain_pool
        self.pretrain_loss = None
        if self.pretrain_steps:
            self._pretrain(seed, first_shape, pretrain_lr)
        self.opt = _adam(self.model.parameters(), lr, self.bucket)
        torch.manual_seed(seed + 1)
        open_net = DGCNNControlPoints(20, num_points=10, mode=0)
        closed_net = DGCNNControlPoints(20, num_points=10, mode=1)
        self.evaluation = Evaluation(closed_path=closed_net, open_path=open_net)
        self.model.eval()
        self.last_res = None
        # The clustering of shape b+1 (few large kernels) is queued on a side stream underneath the
        # fitting stage of shape b (hundreds of tiny launches and the host synchronisations of the
        # Hungarian matching): the reference processes one shape after the other.
        self.batched = True       # stage-wise over the whole batch (fitting_batch.py); False: shape by shape
        # groups of shapes whose clustering is queued ahead of the host's matching work (PARSENET_FIT_CHUNKS).
        # Default 1 = the whole batch as one group: the pipelined form gives identical results
        # (tests/test_fitting_batch_gpu.py) but MEASURED slower on this stack (2 groups: 102 instead of
        # 77 ms per step on the same box; smaller launches, twice the fitting-stage launches, and uploads
        # that queue behind the other group's iterations) — kept as an option, not used
        import os
        self.chunks = int(os.environ.get("PARSENET_FIT_CHUNKS", "1"))
        self.overlap = True       # (shape-by-shape mode) clustering of shape b+1 on a side stream
        self.side = torch.cuda.Stream(device=device)
        self._warmed = False
        self.skipped_steps, self.last_error = 0, None    # steps dropped because a rank's fitting stage raised

    def _pretrain(self, seed, first_shape, lr):
        """See the class docstring.  PARSENET_PRETRAIN_CACHE=<file> keeps the pre-trained weights
        across processes (profiling runs: the trace then holds end-to-end steps only); the file is
        tied to the recipe."""
        import os
        held_out = self.pretrain_pool is not None
        cache = os.environ.get("PARSENET_PRETRAIN_CACHE")
        # (the ids of an explicit shape list belong to the recipe: two steps that differ in nothing else must
        # not load each other's weights — round 4: a test run with the cache variable set did exactly that)
        tag = "seed%d_first%d_B%d_N%d_steps%d_pool%s_lr%g_ids%s" % (
            seed, PRETRAIN_FIRST_SHAPE if held_out else first_shape, self.batch, self.num_points,
            self.pretrain_steps, self.pretrain_pool, lr,
            "-".join(str(int(i)) for i in self.shape_ids) if self.shape_ids is not None else "pool")

        def train():
            state = torch.load(cache, map_location=self.device) if cache and os.path.exists(cache) else None
            if state is not None and state.get("tag") == tag:
                self.model.load_state_dict(state["model"])
                self.pretrain_loss = state.get("loss")
                return
            timed = (self.first_shape, self.pool)
            if held_out:
                self.load_pool(PRETRAIN_FIRST_SHAPE, self.pretrain_pool)
            np.random.seed(4321 + (0 if held_out else first_shape))
            tail = []
            for it in range(self.pretrain_steps):
                loss = self.seg_step()
                if it >= self.pretrain_steps - 8:
                    tail.append(loss.detach())
            self.pretrain_loss = float(torch.stack(tail).mean())
            if held_out:
                self.load_pool(*timed)
            if cache:
                torch.save({"tag": tag, "model": self.model.state_dict(), "loss": self.pretrain_loss}, cache)
        if held_out:
            train_on_rank0_then_broadcast(self.model, self.bucket, train)
        else:
            train()          # round-2 recipe: every rank on its own batch, gradients all-reduced

    def segments_per_shape(self):
        st = self.evaluation.stats
        n = max(st["shapes"], 1)
        return {"clusters": st["clusters"] / n, "fitted": st["fitted"] / n}

    def warm_paths(self):
        """One pass of the clustering + fitting stage (forward and backward) on an embedding that HAS
        cluster structure — a noisy code of the ground-truth segments — so that every per-primitive
        code path (kernels torch loads lazily, rocBLAS / solver heuristics, allocator pools) has run
        once before anything is timed.  With random-init weights those paths are first reached about
        ten optimizer steps in, when the embedding starts to separate, and their one-time set-up cost
        (15-80 ms) would otherwise land in the middle of a measurement.  Nothing of the training state
        is touched: no optimizer step, gradients are discarded, numpy's RNG state is restored."""
        import numpy as np
        state = np.random.get_state()
        g = torch.Generator().manual_seed(12345)
        code = torch.nn.functional.normalize(torch.randn(64, 128, generator=g), dim=1).to(self.device)
        log_prob = torch.log_softmax(torch.randn(self.batch, 10, self.num_points, generator=g), 1).to(self.device)
        stats = dict(self.evaluation.stats)
        embs = []
        for b in range(self.batch):
            lab = torch.from_numpy(np.asarray(self.labels[b]).astype(np.int64) % 64).to(self.device)
            emb = code[lab] + 0.01 * torch.randn(self.num_points, 128, generator=g).to(self.device)
            embs.append(torch.nn.functional.normalize(emb, dim=1))
        emb = torch.stack(embs, 0).requires_grad_(True)
        self.evaluation.batched = self.batched
        res = self.evaluation.fitting_losses(emb, self.points, self.normals, self.labels, self.prim_np, log_prob,
                                             quantile=0.025, iterations=10, lamb=0.1) if self.batched else [
            self.evaluation.fitting_loss(emb[b:b + 1], self.points[b:b + 1], self.normals[b:b + 1],
                                         self.labels[b:b + 1], self.prim_np[b:b + 1], log_prob[b:b + 1],
                                         quantile=0.025, iterations=10, lamb=0.1) for b in range(self.batch)]
        tot = sum(r[0][0].sum() for r in res)
        if torch.is_tensor(tot) and tot.requires_grad:
            tot.backward()
        torch.cuda.synchronize(self.device)
        np.random.set_state(state)
        self.evaluation.stats.update(stats)
        self._warmed = True

    def step(self):
        if not self._warmed:
            self.warm_paths()
        self.next_batch()
        self.bucket.begin()
        embedding, log_prob, embed_loss = self.model(self.x, self.labels, True)
        loss = torch.mean(embed_loss) + primitive_loss(log_prob, self.prim)
        emb = embedding.permute(0, 2, 1)
        res_total = 0
        self.evaluation.batched = self.batched
        if self.batched:
            try:
                loss_b, finish = self.evaluation.fitting_losses_pipelined(
                    emb, self.points, self.normals, self.labels, self.prim_np, log_prob, quantile=0.025,
                    iterations=10, lamb=0.1, chunks=self.chunks)
            except Exception as e:
                # the host part of the stage (matching, segment tables) can raise before anything is
                # queued: with several ranks this rank still has to join the status agreement below
                if not self.bucket._multi():
                    raise
                stage_error = e

                def finish():
                    raise stage_error
                loss_b = [0.0] * self.batch
            # the association of the per-shape sum, ((l0 + l1) + l2) + l3; unbind = ONE backward node (a stack) where
            # indexing shape by shape gave four zero fills, four copies and three accumulations for the same bits
            parts = loss_b.unbind(0) if torch.is_tensor(loss_b) else loss_b
            res_total = functools.reduce(operator.add, parts[:self.batch])
            loss = loss + res_total / self.batch
            try:
                loss.backward()
            except Exception as e:
                # a backward pass that raises on ONE rank: that rank joins the status agreement like the
                # others (who would wait in it for good) and the error propagates from there
                if not self.bucket._multi():
                    raise
                bwd_error = e

                def finish():                                      # noqa: F811
                    raise bwd_error
            # The fit status (lstsq failure, non-finite residual) rides in the deferred download: it
            # is read BEFORE the optimizer step, so a degenerate segment leaves the weights
            # untouched, like the reference's skipped batch (train_parsenet_e2e.py:243-257).  The
            # copy was queued behind the forward kernels; the device is still busy with the backward
            # pass while the host waits for it.  With several ranks the status is agreed upon first
            # (dp.FlatGradBucket.finish_or_skip): either every rank reduces and steps or none does —
            # a rank that raised alone would leave the others in the gradient all-reduce for good.
            self.bucket.gather()
            self.last_metrics, err, took = self.bucket.finish_or_skip(finish, self.opt)
            self.last_res = res_total
            if not took:
                self.skipped_steps += 1
                self.last_error = err
                if err is not None and not self.bucket._multi():
                    raise err                 # one rank: the caller decides (the reference's loop catches it)
            return loss
        main = torch.cuda.current_stream(self.device)
        handles, events = {}, {}

        def prefetch(b):
            self.side.wait_stream(main)            # the embedding (and everything before) is on main
            with torch.cuda.stream(self.side):
                h = self.evaluation.prefetch_clustering(emb[b], 0.025, 10)
                ev = torch.cuda.Event()
                ev.record(self.side)
            for t in h.values():                   # produced on the side stream, consumed on main
                t.record_stream(main)
            handles[b], events[b] = h, ev
        if self.overlap:
            prefetch(0)
        for b in range(self.batch):     # the fitting stage is per shape (reference: batch 1)
            pre = None
            if self.overlap:
                if b + 1 < self.batch:
                    prefetch(b + 1)
                main.wait_event(events[b])
                pre = [handles.pop(b)]
            res, _ = self.evaluation.fitting_loss(emb[b:b + 1], self.points[b:b + 1], self.normals[b:b + 1],
                                                  self.labels[b:b + 1], self.prim_np[b:b + 1],
                                                  log_prob[b:b + 1], quantile=0.025, iterations=10, lamb=0.1,
                                                  prefetched=pre)
            res_total = res_total + res[0]
        loss = loss + res_total / self.batch
        loss.backward()
        self.bucket.gather()
        self.bucket.all_reduce_mean()
        self.opt.step()
        self.last_res = res_total
        return loss


class _SplineCfg:
    def __init__(self, batch_size, grid_size):
        self.batch_size = batch_size
        self.grid_size = grid_size


class SplineNetStep:
    """cfg2 (open, mode 0) / cfg3 (closed, mode 1): one SplineNet training step of
    train_open_splines.py:140-186 / train_closed_control_points.py:141-176 — DGCNNControlPoints
    (k = 10, training-mode BatchNorm) on 700-point patches, the predicted 20x20 grid evaluated on
    the 40x40 (open) / 30x30 (closed) parameter lattice, one-sided Chamfer to the input points,
    permutation regression (8 / 80 candidate orderings), Laplacian (open only);
    loss = 0.9 reg + 0.1 (cd [+ lap]); backward, one gradient all-reduce, Adam (lr 1e-3)."""

    def __init__(self, device, closed=False, batch=32, num_points=700, first_shape=0, seed=0, lr=1e-3,
                 loss_weight=0.9):
        from .bspline import uniform_knot_bspline
        from .encoders import DGCNNControlPoints
        torch.manual_seed(seed)
        self.device = device
        self.closed = closed
        self.batch = batch
        self.num_points = num_points
        self.loss_weight = loss_weight
        self.model = DGCNNControlPoints(20, num_points=10, mode=1 if closed else 0).to(device)
        self.bucket = FlatGradBucket(self.model.parameters())
        self.opt = _adam(self.model.parameters(), lr, self.bucket)
        nu, nv = uniform_knot_bspline(20, 20, 3, 3, 30 if closed else 40)
        self.nu = torch.from_numpy(nu.astype(np.float32)).to(device)
        self.nv = torch.from_numpy(nv.astype(np.float32)).to(device)
        pts, ctrl = synthetic.make_spline_patches(first_shape, batch, num_points, 20, closed)
        self.points = torch.from_numpy(np.ascontiguousarray(pts.transpose(0, 2, 1))).to(device)   # (B,3,P)
        self.control_points = torch.from_numpy(ctrl).to(device)                                  # (B,20,20,3)
        self.cfg = _SplineCfg(batch, 20)
        self.last = None
        self.scales = None      # per-patch anisotropic scales of a file-backed loader (trainer)

    def shapes_per_step(self):
        return self.batch

    def losses(self, output):
        from . import spline_losses as SL
        points, control_points = self.points, self.control_points
        if self.scales is not None:
            # anisotropic canonicalisation: back to a common scale before the losses
            # (train_open_splines.py:156-159, src/utils.py:361-390)
            sc = torch.from_numpy(np.stack(self.scales, 0).astype(np.float32)).to(output.device).reshape((-1, 1, 3))
            smax = torch.max(sc.reshape((-1, 3)), 1)[0]
            output = output * sc / smax.reshape((-1, 1, 1))
            points = points * sc.reshape((-1, 3, 1)) / smax.reshape((-1, 1, 1))
            control_points = control_points * sc.reshape((-1, 1, 1, 3)) / smax.reshape((-1, 1, 1, 1))
        cd, _ = SL.spline_reconstruction_loss_one_sided(self.nu, self.nv, output, points, self.cfg)
        if self.closed:
            l_reg, _ = SL.control_points_permute_closed_reg_loss(output, control_points, 20, 20)
            loss = l_reg * self.loss_weight + cd * (1 - self.loss_weight)
            return loss, cd, l_reg, None
        l_reg, permute_cp = SL.control_points_permute_reg_loss(output, control_points, 20)
        lap = SL.laplacian_loss(output.reshape((self.batch, 20, 20, 3)), permute_cp, dist_type="l2")
        loss = l_reg * self.loss_weight + (cd + lap) * (1 - self.loss_weight)
        return loss, cd, l_reg, lap

    def step(self):
        # (this step is bound by the host's launch rate, not by the device: autograd adding into the bucket's
        # views costs the host less than begin() / gather() — 4.15 against 4.33 ms per cfg2 step, alternating on
        # one box, tools/jobs/r5n.sh; the device-bound steps above gain from the gathered form)
        self.bucket.zero()
        output = self.model(self.points)
        loss, cd, l_reg, lap = self.losses(output)
        loss.backward()
        self.bucket.all_reduce_mean()
        self.opt.step()
        self.last = (cd, l_reg, lap)
        return loss
