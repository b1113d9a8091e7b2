"""The BASELINE.json configurations as callable training steps on synthetic data (used by
bench.py, __graft_entry__.smoke() and the tests).  Everything a step touches is resident on
the GPU before the timed region starts."""
import numpy as np
import torch

from . import synthetic
from .dp import FlatGradBucket
from .encoders import PrimitivesEmbeddingDGCNGn
from .losses import EmbeddingLoss, primitive_loss


class ParsenetSegStep:
    """cfg4: ParSeNet segmentation-only training step (train_parsenet.py:151-198): points +
    normals (6 channels), first graph on the points+normals metric, k = 80, triplet embedding
    loss + NLL primitive loss, forward + backward + one gradient all-reduce + Adam."""

    def __init__(self, device, batch=4, num_points=10000, nn_nb=80, first_shape=0, seed=0, lr=1e-2):
        torch.manual_seed(seed)
        self.device = device
        self.batch = batch
        self.num_points = num_points
        self.loss = EmbeddingLoss(margin=1.0, if_mean_shift=False)
        self.model = PrimitivesEmbeddingDGCNGn(embedding=True, emb_size=128, primitives=True,
                                               num_primitives=10, loss_function=self.loss.triplet_loss,
                                               mode=5, num_channels=6, nn_nb=nn_nb).to(device)
        self.bucket = FlatGradBucket(self.model.parameters())
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        pts, nrm, lab, prim = synthetic.make_batch(first_shape, batch, num_points)
        x = np.concatenate([pts, nrm], 2).transpose(0, 2, 1)           # (B,6,N)
        self.x = torch.from_numpy(np.ascontiguousarray(x)).to(device)
        self.labels = lab                                              # host ints (reference: numpy)
        self.prim = torch.from_numpy(prim).to(device)
        self.rng_seed = seed

    def shapes_per_step(self):
        return self.batch

    def step(self):
        self.bucket.zero()
        embedding, log_prob, embed_loss = self.model(self.x, self.labels, True)
        loss = torch.mean(embed_loss) + primitive_loss(log_prob, self.prim)
        loss.backward()
        self.bucket.all_reduce_mean()
        self.opt.step()
        return loss


class ParsenetE2EStep(ParsenetSegStep):
    """cfg5: the end-to-end step of train_parsenet_e2e.py:190-277 per shape — segmentation
    network (norm layers frozen: model.eval()), then for every shape of the batch mean-shift
    clustering of the embedding (quantile 0.025, 10 iterations), Hungarian matching, weighted
    primitive / SplineNet fits and residual losses (lamb 0.1); loss = triplet + NLL + residual,
    backward through everything, one gradient all-reduce, Adam.  The SplineNets are frozen
    random-init DGCNNControlPoints (no pretrained weights ship with the reference)."""

    def __init__(self, device, batch=4, num_points=10000, nn_nb=80, first_shape=0, seed=0, lr=1e-4):
        super().__init__(device, batch, num_points, nn_nb, first_shape, seed, lr)
        from .encoders import DGCNNControlPoints
        from .fitting import Evaluation
        torch.manual_seed(seed + 1)
        open_net = DGCNNControlPoints(20, num_points=10, mode=0)
        closed_net = DGCNNControlPoints(20, num_points=10, mode=1)
        self.evaluation = Evaluation(closed_path=closed_net, open_path=open_net)
        self.model.eval()
        pts, nrm, lab, prim = synthetic.make_batch(first_shape, batch, num_points)
        self.points = torch.from_numpy(pts).to(device)
        self.normals = torch.from_numpy(nrm).to(device)
        self.prim_np = prim
        self.last_res = None

    def step(self):
        self.bucket.zero()
        embedding, log_prob, embed_loss = self.model(self.x, self.labels, True)
        loss = torch.mean(embed_loss) + primitive_loss(log_prob, self.prim)
        emb = embedding.permute(0, 2, 1)
        res_total = 0
        for b in range(self.batch):     # the fitting stage is per shape (reference: batch 1)
            res, _ = self.evaluation.fitting_loss(emb[b:b + 1], self.points[b:b + 1], self.normals[b:b + 1],
                                                  self.labels[b:b + 1], self.prim_np[b:b + 1],
                                                  log_prob[b:b + 1], quantile=0.025, iterations=10, lamb=0.1)
            res_total = res_total + res[0]
        loss = loss + res_total / self.batch
        loss.backward()
        self.bucket.all_reduce_mean()
        self.opt.step()
        self.last_res = res_total
        return loss
