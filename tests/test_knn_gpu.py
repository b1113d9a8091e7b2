"""Parity of the HIP kNN kernels against the C oracle: indices bit-exact on arbitrary inputs
(same fma chain, same tie rule), including ragged sizes, duplicates and the k limits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu_knn(x, k, gpu, metric="feature"):
    from parsenet_codebase_amd import kernels
    return kernels.knn(torch.from_numpy(x).to(gpu), k, metric).cpu().numpy()


@pytest.mark.parametrize("B,C,N,k", [(1, 3, 64, 10), (2, 3, 700, 10), (3, 64, 333, 10),
                                     (1, 128, 700, 10), (2, 6, 1000, 80), (1, 64, 2500, 80),
                                     (1, 256, 450, 10), (1, 3, 130, 128), (1, 5, 40, 1)])
def test_feature_metric_bit_exact(gpu, B, C, N, k):
    from oracle import cbind
    rng = np.random.RandomState(100 * C + N)
    x = rng.uniform(-1, 1, (B, C, N)).astype(np.float32)
    got = _gpu_knn(x, k, gpu)
    want = cbind.knn(x, k, 0)
    assert got.shape == (B, N, k)
    assert np.array_equal(got, want), "mismatching rows: %d" % (got != want).any(-1).sum()


@pytest.mark.parametrize("B,N,k", [(2, 700, 80), (1, 3000, 80), (1, 97, 20)])
def test_points_normals_metric_bit_exact(gpu, B, N, k):
    from oracle import cbind
    rng = np.random.RandomState(N)
    p = rng.uniform(-0.5, 0.5, (B, 3, N)).astype(np.float32)
    n = rng.normal(size=(B, 3, N)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    x = np.concatenate([p, n], 1).astype(np.float32)
    got = _gpu_knn(x, k, gpu, "points_normals")
    want = cbind.knn(x, k, 1)
    assert np.array_equal(got, want), "mismatching rows: %d" % (got != want).any(-1).sum()


def test_duplicate_points_tie_rule(gpu):
    """Coincident points give exactly equal values: the smaller index must come first."""
    from oracle import cbind
    rng = np.random.RandomState(3)
    base = rng.uniform(-1, 1, (1, 3, 50)).astype(np.float32)
    x = np.concatenate([base, base, base, base], 2)  # every point 4 times
    got = _gpu_knn(x, 8, gpu)
    want = cbind.knn(x, 8, 0)
    assert np.array_equal(got, want)


def test_self_is_first_and_sorted(gpu):
    """Known answers of the reference: for distinct points the point itself is
    neighbour 0 and values are non-increasing along k."""
    from oracle import cbind
    rng = np.random.RandomState(11)
    x = rng.uniform(-0.5, 0.5, (1, 3, 10000)).astype(np.float32)
    idx = _gpu_knn(x, 80, gpu)[0]
    assert np.array_equal(idx[:, 0], np.arange(10000))
    for i in (0, 1234, 9999):
        v = cbind.knn_row_values(x[0], i)
        vs = v[idx[i]]
        assert (np.diff(vs) <= 0).all()
        assert np.sort(v)[::-1][79] == vs[-1]


def test_argument_errors(gpu):
    from parsenet_codebase_amd import kernels
    x = torch.zeros(1, 3, 16, device=gpu)
    with pytest.raises(RuntimeError):
        kernels.knn(x, 17)
    with pytest.raises(RuntimeError):
        kernels.knn(x, 0)
    with pytest.raises(RuntimeError):
        kernels.knn(torch.zeros(1, 3, 300, device=gpu), 129)


def test_degenerate_input_takes_the_gated_fallback(gpu):
    """All points coincide: every value ties, the survivor lists of the MFMA path overflow and
    the flagged queries are recomputed by the generic scan kernel.  Result: indices 0..k-1."""
    from oracle import cbind
    x = np.zeros((2, 64, 2000), np.float32)
    x[1, :, 1000:] = 1.0  # two clusters of coincident points in the second item
    got = _gpu_knn(x, 20, gpu)
    want = cbind.knn(x, 20, 0)
    assert np.array_equal(got, want)


def test_order_independence_of_the_mfma_path(gpu):
    """Spatially sorted input (neighbours contiguous in memory) must not change the result:
    the internal candidate permutation only affects speed."""
    from oracle import cbind
    rng = np.random.RandomState(5)
    x = rng.uniform(-1, 1, (1, 3, 4000)).astype(np.float32)
    order = np.argsort(x[0, 0])
    xs = np.ascontiguousarray(x[:, :, order])
    got = _gpu_knn(xs, 40, gpu)
    want = cbind.knn(xs, 40, 0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("x3_level", ["1", "2"])
def test_near_ties_under_the_bf16_passes(gpu, x3_level, monkeypatch):
    """The passes on the bf16 matrix cores only pre-select: rows with large norms whose neighbours
    differ by far less than the error of an approximate distance (tight clumps far from the
    origin, exact duplicates among them) must still come out bit-exact — by the exact repairs of
    the final sort, or through the flag and the exact scan kernel when a window holds too many."""
    from oracle import cbind
    monkeypatch.setenv("PN_KNN_X3", x3_level)
    rng = np.random.RandomState(7)
    B, C, N, k = 2, 64, 4200, 80
    centres = rng.uniform(-3, 3, (B, C, 12)).astype(np.float32)
    lab = rng.randint(0, 12, (B, N))
    x = np.take_along_axis(centres, lab[:, None, :].repeat(C, 1), 2)
    x = x + rng.normal(0, 1.0, (B, C, N)).astype(np.float32) * np.where(lab[:, None, :] < 6, 1e-4, 0.3).astype(np.float32)
    x[:, :, 100:140] = x[:, :, 60:100]                     # exact duplicates
    x = np.ascontiguousarray(x.astype(np.float32))
    got = _gpu_knn(x, k, gpu)
    want = cbind.knn(x, k, 0)
    assert np.array_equal(got, want), "mismatching rows: %d" % (got != want).any(-1).sum()


@pytest.mark.parametrize("B,C,N,k", [(1, 128, 3000, 80), (3, 40, 2100, 128), (2, 100, 2049, 20), (1, 64, 4097, 1),
                                     (10, 256, 2500, 10), (12, 200, 2111, 80), (12, 256, 2049, 1)])
@pytest.mark.parametrize("x3_level", ["1", "2"])
def test_bf16_passes_on_odd_shapes(gpu, B, C, N, k, x3_level, monkeypatch):
    """The split passes (64-, 128- and 256-channel images, padded channels, tails of the last tile,
    k = 1 and k = 128) against the C oracle."""
    from oracle import cbind
    monkeypatch.setenv("PN_KNN_X3", x3_level)
    rng = np.random.RandomState(C + N)
    x = (rng.uniform(-1, 1, (B, C, N)) * rng.uniform(0.2, 3.0, (B, C, 1))).astype(np.float32)
    got = _gpu_knn(x, k, gpu)
    want = cbind.knn(x, k, 0)
    assert np.array_equal(got, want), "mismatching rows: %d" % (got != want).any(-1).sum()


@pytest.mark.parametrize("C,N,K", [(128, 5000, 125), (64, 2500, 60), (20, 3000, 75)])
def test_kth_dot_in_bf16x3_arithmetic(gpu, C, N, K):
    """pn_dot_kth_x3_f32 (the bandwidth statistic under the default arithmetic): the K-th largest
    dot product of every row to fp32 grade against the exact selection engine."""
    from parsenet_codebase_amd import kernels as K_
    g = torch.Generator().manual_seed(C * N)
    x = torch.nn.functional.normalize(torch.randn(2, N, C, generator=g), dim=2).to(gpu)
    exact, f0 = K_.dot_select(x, x, K, want_value=True)
    got = K_.dot_kth_x3(x, x, K)
    assert got is not None
    val, f1 = got
    assert int(f0.sum()) == 0 and int(f1.sum()) == 0
    assert float((val - exact).abs().max()) < 1e-6


@pytest.mark.parametrize("B,C,N,k", [(1, 3, 5000, 10), (2, 64, 5000, 10), (1, 128, 2600, 10), (1, 256, 2111, 10),
                                     (3, 3, 32, 10), (2, 7, 33, 5), (1, 64, 700, 16), (2, 20, 1500, 11), (4, 6, 97, 1),
                                     (1, 3, 20000, 10), (2, 200, 1500, 10), (3, 256, 5000, 8)])
def test_small_k_one_pass_kernel_bit_exact(gpu, B, C, N, k):
    """k <= 16 (the SplineNets' graphs, src/model.py:9-22): one distance pass with the k best of a lane in
    registers (csrc/knn_smallk.h) — sliced candidate ranges merged by the second kernel, a single slice written
    directly, tails of the last tile, KK = 10 and 16, every channel width, the 256-channel instance with four
    (N < 2 048) and with eight waves per workgroup — against the C oracle."""
    from oracle import cbind
    rng = np.random.RandomState(17 * C + N + k)
    x = (rng.uniform(-1, 1, (B, C, N)) * rng.uniform(0.2, 3.0, (B, C, 1))).astype(np.float32)
    got = _gpu_knn(x, k, gpu)
    want = cbind.knn(x, k, 0)
    assert got.shape == (B, N, k)
    assert np.array_equal(got, want), "mismatching rows: %d" % (got != want).any(-1).sum()


def test_small_k_ties_and_the_two_pass_engine_agree(gpu, monkeypatch):
    """Masses of equal values (coincident points, a lattice) order by the smaller index in every lane, between
    the two lanes of a query and between slices; and the graph equals the two-pass engine's (PN_KNN_SMALLK=0)."""
    from oracle import cbind
    rng = np.random.RandomState(23)
    lat = (rng.randint(-8, 9, (2, 3, 3000)) / 16.0).astype(np.float32)        # many exact ties and duplicates
    coin = np.zeros((1, 64, 2100), np.float32)
    coin[0, :, 1000:] = 0.5
    wide = np.zeros((1, 250, 2100), np.float32)                               # (the eight-wave instance)
    wide[0, :, 700:] = 0.25
    wide[0, :7, 1400:] = -0.5
    for x, k in ((lat, 10), (coin, 10), (lat[:, :, :257], 16), (wide, 10)):
        got = _gpu_knn(x, k, gpu)
        assert np.array_equal(got, cbind.knn(x, k, 0))
        monkeypatch.setenv("PN_KNN_SMALLK", "0")
        old = _gpu_knn(x, k, gpu)
        monkeypatch.delenv("PN_KNN_SMALLK")
        assert np.array_equal(got, old)


@pytest.mark.parametrize("np_", ["3", "6"])
@pytest.mark.parametrize("B,C,N,k", [(2, 64, 5000, 10), (1, 128, 5000, 10), (1, 256, 4100, 10), (3, 100, 2100, 7),
                                     (1, 40, 2049, 1), (2, 128, 2500, 10)])
def test_one_pass_bf16_graph_for_small_k_bit_exact(gpu, B, C, N, k, np_, monkeypatch):
    """PN_KNN_FUSED=1 (csrc/knn_x3.h, KIND 2): threshold and collection of the bf16 x 3 graph in ONE pass (running
    threshold from a lane's largest group maxima, exact repairs in the final sort) against the C oracle — features with
    a large common offset (what centring is for) and very different channel scales, three and six piece products,
    point counts with a ragged last tile, k below the list length the lanes keep."""
    from oracle import cbind
    rng = np.random.RandomState(31 * C + N + k)
    x = (rng.uniform(-1, 1, (B, C, N)) * rng.uniform(0.2, 3.0, (B, C, 1)) + rng.uniform(0.0, 4.0, (B, C, 1))).astype(np.float32)
    monkeypatch.setenv("PN_KNN_FUSED", "1")
    monkeypatch.setenv("PN_KNN_FUSED_NP", np_)
    got = _gpu_knn(x, k, gpu)
    want = cbind.knn(x, k, 0)
    assert got.shape == (B, N, k)
    assert np.array_equal(got, want), "mismatching rows: %d" % (got != want).any(-1).sum()


def test_one_pass_bf16_graph_on_ties_clusters_and_degenerate_input(gpu, monkeypatch):
    """The one-pass form on data the approximation cannot decide: coincident points (every distance tied: the rows
    overflow their lists, are flagged and redone by the gated scan), tight clusters (hundreds of candidates inside
    the 2-eps window: all re-evaluated exactly), a lattice with exact ties; and the same graph as the exact one-pass
    kernel (PN_KNN_FUSED=0) on a SplineNet-like input."""
    from oracle import cbind
    rng = np.random.RandomState(41)
    coin = np.zeros((1, 64, 2100), np.float32)
    coin[0, :, 1000:] = 0.5
    centres = rng.uniform(-1, 1, (1, 128, 12)).astype(np.float32)
    lab = rng.randint(0, 12, 3000)
    tight = (centres[:, :, lab] + 1e-4 * rng.normal(size=(1, 128, 3000))).astype(np.float32) + 3.0
    lat = np.repeat((rng.randint(-8, 9, (2, 4, 2500)) / 16.0).astype(np.float32), 16, axis=1)      # 64 channels
    feat = np.maximum(rng.normal(size=(2, 64, 5000)) + 0.5, 0.0).astype(np.float32)                # post-ReLU features
    for x, k in ((coin, 10), (tight, 10), (lat, 10), (feat, 10)):
        monkeypatch.setenv("PN_KNN_FUSED", "1")
        got = _gpu_knn(x, k, gpu)
        monkeypatch.setenv("PN_KNN_FUSED", "0")
        ref = _gpu_knn(x, k, gpu)
        assert np.array_equal(got, ref), "mismatching rows: %d" % (got != ref).any(-1).sum()
        if x.shape[2] <= 3000:
            assert np.array_equal(got, cbind.knn(x, k, 0))
