"""src/PointNet.py of the reference (ParSeNet segmentation network) on the HIP kernels."""
from parsenet_codebase_amd import graph as _graph
from parsenet_codebase_amd.encoders import (DGCNNEncoderGn, PrimitivesEmbeddingDGCNGn,  # noqa: F401
                                            PrimitivesEmbeddingDGCNGne2e)


def knn(x, k1, k2):
    """Top-k2 neighbours, keeping every (k2 // k1)-th (dilation); identity when k1 == k2."""
    return _graph.knn_dilated(x, k1, k2)


def knn_points_normals(x, k1, k2):
    """First-layer graph on (B,6,N) points+normals: |dp|^2 * (1 + (2 - 2 ni.nj))."""
    return _graph.knn_points_normals(x, k1, k2)


def get_graph_feature(x, k1=20, k2=20, idx=None):
    batch_size, num_points = x.size(0), x.size(2)
    x = x.view(batch_size, -1, num_points)
    if idx is None:
        idx = knn(x, k1=k1, k2=k2)
    return _graph.graph_feature(x, idx)


def get_graph_feature_with_normals(x, k1=20, k2=20, idx=None):
    """Edge features on all 6 channels, graph from the points+normals metric."""
    batch_size, num_points = x.size(0), x.size(2)
    x = x.view(batch_size, -1, num_points)
    if idx is None:
        idx = knn_points_normals(x, k1=k1, k2=k2)
    return _graph.graph_feature(x, idx)
