"""Parity of the HIP Chamfer nearest-neighbour kernel against the C oracle (bit-exact
minima and arg-mins) at sizes the oracle finishes in seconds, plus size-independent
properties at the 10k x 10k size of BASELINE.json."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["scalar", "mfma", "auto"], autouse=True)
def chamfer_path(request, monkeypatch):
    """Every test of this file on both kernels — PN_CHAMFER_MFMA=0: the scalar kernel, =1: the matrix-core
    pre-filter with the exact decision (round 6), whatever the size — and on the library's own choice."""
    if request.param == "auto":
        monkeypatch.delenv("PN_CHAMFER_MFMA", raising=False)
    else:
        monkeypatch.setenv("PN_CHAMFER_MFMA", "1" if request.param == "mfma" else "0")
    return request.param


def _run(a, b, gpu):
    from parsenet_codebase_amd import kernels
    ta = torch.from_numpy(a).to(gpu)
    tb = torch.from_numpy(b).to(gpu)
    return [t.cpu().numpy() for t in kernels.chamfer_nn(ta, tb)]


@pytest.mark.parametrize("B,Na,Nb", [(1, 1, 1), (1, 7, 300), (3, 257, 129), (32, 1600, 700),
                                     (1, 10000, 3001), (2, 900, 5000)])
def test_bit_exact_vs_oracle(gpu, B, Na, Nb):
    from oracle import cbind
    rng = np.random.RandomState(B * 1000 + Na + Nb)
    a = rng.uniform(-0.5, 0.5, (B, Na, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (B, Nb, 3)).astype(np.float32)
    minA, argA, minB, argB = _run(a, b, gpu)
    oA, oiA = cbind.chamfer_nn(a, b)
    oB, oiB = cbind.chamfer_nn(b, a)
    assert np.array_equal(argA, oiA) and np.array_equal(argB, oiB)
    assert np.array_equal(minA.view(np.uint32), oA.view(np.uint32))
    assert np.array_equal(minB.view(np.uint32), oB.view(np.uint32))


def test_ties_resolve_to_smallest_index(gpu):
    a = np.zeros((1, 5, 3), np.float32)
    b = np.zeros((1, 4000, 3), np.float32)
    b[0, :, 0] = 1.0
    minA, argA, minB, argB = _run(a, b, gpu)
    assert (argA == 0).all() and (argB == 0).all()
    assert np.allclose(minA, 1.0)


def test_full_size_properties(gpu):
    """10k x 10k (test.py:157-160 size): Chamfer(P,P)=0 with identity arg-min, and a
    permuted copy recovers the permutation."""
    rng = np.random.RandomState(7)
    p = rng.uniform(-0.5, 0.5, (1, 10000, 3)).astype(np.float32)
    perm = rng.permutation(10000)
    q = p[:, perm]
    minA, argA, minB, argB = _run(p, q, gpu)
    assert (minA == 0).all() and (minB == 0).all()
    assert np.array_equal(argB[0], perm)
    assert np.array_equal(perm[argA[0]], np.arange(10000))


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (40.0, 0.0), (1.0, 300.0), (1e-3, 0.0), (1.0, -2.5)])
def test_prefilter_bound_holds_far_from_the_origin_and_on_clumps(gpu, scale, offset):
    """The certified bound of the matrix-core pre-filter scales with |q||c|: clouds far from the origin (where the
    norms dwarf the distances and almost every candidate passes the filter), tiny and large extents, tight clumps
    and exact duplicates — minima and arg-mins stay bit-identical to the oracle's."""
    from oracle import cbind
    rng = np.random.RandomState(17)
    a = rng.uniform(-0.5, 0.5, (2, 1500, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (2, 2100, 3)).astype(np.float32)
    b[:, 1000:1400] = b[:, 600:1000] + rng.normal(0, 1e-5, (2, 400, 3)).astype(np.float32)     # clumps
    b[:, 1400:1500] = b[:, 100:200]                                                            # exact duplicates
    a[:, :100] = b[:, 1400:1500]                                                               # distance 0, ties
    a = (a * scale + offset).astype(np.float32)
    b = (b * scale + offset).astype(np.float32)
    minA, argA, minB, argB = _run(a, b, gpu)
    oA, oiA = cbind.chamfer_nn(a, b)
    oB, oiB = cbind.chamfer_nn(b, a)
    assert np.array_equal(argA, oiA) and np.array_equal(argB, oiB)
    assert np.array_equal(minA.view(np.uint32), oA.view(np.uint32))
    assert np.array_equal(minB.view(np.uint32), oB.view(np.uint32))


def test_non_finite_coordinates_give_the_empty_key_on_both_kernels(gpu):
    """A query with a NaN / inf coordinate has no finite distance: both kernels leave the "empty" pattern; a
    non-finite CANDIDATE never wins."""
    rng = np.random.RandomState(3)
    a = rng.uniform(-0.5, 0.5, (1, 300, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, (1, 400, 3)).astype(np.float32)
    a[0, 5, 1] = np.nan
    a[0, 9, 0] = np.inf
    b[0, 17, 2] = np.nan
    minA, argA, minB, argB = _run(a, b, gpu)
    assert argA[0, 5] == 0xffffffff and argA[0, 9] == 0xffffffff
    ok = np.ones(300, bool)
    ok[[5, 9]] = False
    assert (argA[0, ok] != 17).all() and np.isfinite(minA[0, ok]).all()
    d = ((a[0, ok, None, :] - b[0, None, :, :]) ** 2)
    d = (d[..., 0] + d[..., 1]) + d[..., 2]
    d[:, 17] = np.inf
    assert np.array_equal(argA[0, ok], d.argmin(1))
