"""Parity of the HIP Chamfer nearest-neighbour kernel against the C oracle (bit-exact
minima and arg-mins) at sizes the oracle finishes in seconds, plus size-independent
properties at the 10k x 10k size of BASELINE.json."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
