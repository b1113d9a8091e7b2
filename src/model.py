"""src/model.py of the reference (SplineNet) on the HIP kernels.  The north-star signature
``src.model.PrimitivesEmbeddingDGCNGn`` is exported here as well as from src.PointNet,
where the reference defines it."""
import numpy as np

from parsenet_codebase_amd import graph as _graph
from parsenet_codebase_amd.encoders import (DGCNNControlPoints, DGCNNEncoderGn,  # noqa: F401
                                            PrimitivesEmbeddingDGCNGn)

EPS = np.finfo(np.float32).eps


def knn(x, k):
    """(B,C,N) -> (B,N,k) int64 neighbour indices, nearest first, the point itself included."""
    return _graph.knn(x, k)


def get_graph_feature(x, k=20, idx=None):
    """(B,C,N) -> (B,2C,N,k) edge features cat(x_j - x_i, x_i); ``idx`` (B,N,k) optional."""
    batch_size, num_points = x.size(0), x.size(2)
    x = x.contiguous().view(batch_size, -1, num_points)
    if idx is None:
        idx = knn(x, k=k)
    return _graph.graph_feature(x, idx)
