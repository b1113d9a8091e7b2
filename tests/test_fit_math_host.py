"""CPU: the arithmetic of the batched primitive fits (parsenet_codebase_amd/csrc/fit_math.h — the
source the gfx950 kernels of fitbatch.hip are compiled from) against the oracle's restatement of
Fit.fit_*_torch + ComputePrimitiveDistance (oracle/ref_fitting.py, pinned by the reference
fixtures): parameters, residual distance and d(distance)/d(weights) through the custom-gradient
SVD, the rank test / ridge search and the clamps.  The header is compiled for the host by
tests/native/fit_math_host.cpp (test infrastructure, not shipped in the product library)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import ref_fitting as RF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS32 = float(np.finfo(np.float32).eps)
KINDS = {"plane": 0, "sphere": 1, "cylinder": 2, "cone": 3}


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fbh") / "libfbh.so")
    src = os.path.join(ROOT, "tests", "native", "fit_math_host.cpp")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", out, src], check=True)
    lib = ctypes.CDLL(out)
    lib.fbh_segment.restype = ctypes.c_int
    return lib


def run_harness(lib, P, Nrm, w, kind, gt, sqrt_flag=0):
    n = P.shape[0]
    P, Nrm, w, gt = [np.ascontiguousarray(a, dtype=np.float32) for a in (P, Nrm, w, gt)]
    params = np.zeros(16)
    mom = np.zeros(64)
    dist = ctypes.c_float()
    gw = np.zeros(n)
    jac = np.zeros((16, 64))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    st = lib.fbh_segment(p(P), p(Nrm), p(w), n, KINDS[kind], p(gt), gt.shape[0], sqrt_flag, p(params), p(mom),
                         ctypes.byref(dist), p(gw), p(jac))
    return st, params, float(dist.value), gw, mom, jac


def make_segment(kind, seed, n=600, noise=0.004):
    """Noisy samples of one primitive patch + soft memberships that favour it, like a segment of a
    normalised shape: coordinates within [-0.5, 0.5]."""
    rng = np.random.RandomState(seed)
    u, v = rng.rand(n), rng.rand(n)
    if kind == "plane":
        a = rng.randn(3)
        a /= np.linalg.norm(a)
        e1 = np.cross(a, [1, 0, 0.3])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(a, e1)
        P = 0.1 * a + np.outer(u - 0.5, e1) * 0.8 + np.outer(v - 0.5, e2) * 0.6
        Nn = np.tile(a, (n, 1))
    elif kind == "sphere":
        th, ph = u * np.pi * 0.7 + 0.3, v * 2 * np.pi
        d = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], 1)
        P = np.array([0.05, -0.1, 0.02]) + 0.3 * d
        Nn = d
    elif kind == "cylinder":
        ax = rng.randn(3)
        ax /= np.linalg.norm(ax)
        e1 = np.cross(ax, [0.2, 1, 0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(ax, e1)
        ph = u * 2 * np.pi
        d = np.outer(np.cos(ph), e1) + np.outer(np.sin(ph), e2)
        P = np.array([0.02, 0.03, -0.05]) + 0.2 * d + np.outer(v - 0.5, ax) * 0.7
        Nn = d
    else:
        ax = rng.randn(3)
        ax /= np.linalg.norm(ax)
        e1 = np.cross(ax, [0.2, 1, 0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(ax, e1)
        half = 0.5
        h = 0.15 + 0.5 * v
        ph = u * 2 * np.pi
        rad = np.outer(np.cos(ph), e1) + np.outer(np.sin(ph), e2)
        apex = np.array([-0.1, 0.05, -0.2])
        P = apex + h[:, None] * (np.cos(half) * ax + np.sin(half) * rad)
        Nn = np.cos(half) * rad - np.sin(half) * ax
    P = P + noise * rng.randn(n, 3)
    Nn = Nn + 0.05 * rng.randn(n, 3)
    Nn /= np.linalg.norm(Nn, axis=1, keepdims=True)
    w = np.clip(0.75 + 0.25 * rng.randn(n), 0.02, 1.0) + EPS32
    gt = P[rng.rand(n) < 0.7] + 0.002 * rng.randn(int((rng.rand(n) < 0.7).sum()), 3) if False else \
        P[::2] + 0.002 * rng.randn(P[::2].shape[0], 3)
    return P.astype(np.float32), Nn.astype(np.float32), w.astype(np.float32), gt.astype(np.float32)


def oracle(kind, P, Nn, w, gt, dtype, sqrt=False):
    """(params, distance, d distance / d w) of the oracle in ``dtype``; the rank test keeps the
    reference's fp32 tolerance (max(shape) * eps32) so that fp64 runs take the same branch."""
    Pt, Nt, gtt = [torch.tensor(a, dtype=dtype) for a in (P, Nn, gt)]
    wt = torch.tensor(w, dtype=dtype).reshape(-1, 1).requires_grad_(True)
    orig = torch.linalg.matrix_rank

    def rank32(A, *a, **k):
        sv = torch.linalg.svdvals(A)
        return int((sv > sv.max() * max(A.shape) * EPS32).sum())
    torch.linalg.matrix_rank = rank32
    try:
        if kind == "plane":
            params = RF.fit_plane(Pt, wt)
        elif kind == "sphere":
            params = RF.fit_sphere(Pt, wt)
        elif kind == "cylinder":
            params = RF.fit_cylinder(Pt, Nt, wt)
        else:
            params = RF.fit_cone(Pt, Nt, wt)
        d = RF.distance(kind, gtt, params, sqrt=sqrt)
    finally:
        torch.linalg.matrix_rank = orig
    d.backward()
    flat = torch.cat([torch.as_tensor(p).reshape(-1) for p in params]).detach().numpy()
    return flat, float(d), wt.grad[:, 0].numpy()


def align_sign(kind, mine, ref):
    """Singular vectors are defined up to sign: plane (a, d) ~ (-a, -d); cylinder axis ~ -axis."""
    mine = mine.copy()
    if kind == "plane" and np.dot(mine[:3], ref[:3]) < 0:
        mine[:4] *= -1
    if kind == "cylinder" and np.dot(mine[:3], ref[:3]) < 0:
        mine[:3] *= -1
    return mine


@pytest.mark.parametrize("kind", ["plane", "sphere", "cone"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fit_and_gradient_match_the_fp64_oracle(harness, kind, seed):
    P, Nn, w, gt = make_segment(kind, seed)
    st, params, dist, gw, _, _ = run_harness(harness, P, Nn, w, kind, gt)
    assert st == 0
    npar = {"plane": 4, "sphere": 4, "cone": 7}[kind]
    ref_p, ref_d, ref_g = oracle(kind, P, Nn, w, gt, torch.float64)
    mine = align_sign(kind, params[:npar], ref_p)
    # the fit itself is fp64 on both sides; the cone's half angle and the residual run in fp32
    tol = 2e-6 if kind == "cone" else 1e-9
    assert np.abs(mine - ref_p).max() < tol * max(1.0, np.abs(ref_p).max()), (mine, ref_p)
    assert abs(dist - ref_d) < 2e-6 * abs(ref_d) + 1e-12
    scale = np.abs(ref_g).max()
    assert np.abs(gw - ref_g).max() < 5e-5 * scale, np.abs(gw - ref_g).max() / scale


@pytest.mark.parametrize("kind", ["plane", "sphere", "cylinder", "cone"])
def test_fit_matches_the_fp32_oracle(harness, kind):
    """Against the oracle in the reference's own precision (what the fixtures pin): parameters to
    fp32 rounding; the cylinder goes through the ridge branch, whose fp32 solve in the reference
    is noisy along the (irrelevant) axial direction — compared through the distance and the
    components that matter."""
    P, Nn, w, gt = make_segment(kind, 5)
    st, params, dist, gw, _, _ = run_harness(harness, P, Nn, w, kind, gt)
    assert st == 0
    ref_p, ref_d, ref_g = oracle(kind, P, Nn, w, gt, torch.float32)
    if kind == "cylinder":
        mine = align_sign(kind, params[:7], ref_p)
        ax = ref_p[:3]
        assert np.abs(mine[:3] - ax).max() < 1e-4
        dc = mine[3:6] - ref_p[3:6]
        assert np.linalg.norm(dc - np.dot(dc, ax) * ax) < 2e-3      # centre, off the axis
        assert abs(mine[6] - ref_p[6]) < 2e-3 * ref_p[6]
        assert abs(dist - ref_d) < 5e-2 * ref_d
        return
    npar = {"plane": 4, "sphere": 4, "cone": 7}[kind]
    mine = align_sign(kind, params[:npar], ref_p)
    assert np.abs(mine - ref_p).max() < 2e-4 * max(1.0, np.abs(ref_p).max())
    assert abs(dist - ref_d) < 1e-3 * ref_d
    assert np.abs(gw - ref_g).max() < 2e-2 * np.abs(ref_g).max()


def test_cylinder_against_fp64_oracle_with_the_same_ridge(harness):
    """The cylinder's circle fit is rank deficient along the axis: both sides take the ridge
    branch (lambda from the fp32-tolerance rank test) and then agree in fp64."""
    for seed in (0, 3):
        P, Nn, w, gt = make_segment("cylinder", seed)
        st, params, dist, gw, _, _ = run_harness(harness, P, Nn, w, "cylinder", gt)
        assert st == 0 and params[15] > 0          # ridge parameter in use
        ref_p, ref_d, ref_g = oracle("cylinder", P, Nn, w, gt, torch.float64)
        mine = align_sign("cylinder", params[:7], ref_p)
        assert np.abs(mine[:3] - ref_p[:3]).max() < 1e-8
        assert abs(mine[6] - ref_p[6]) < 1e-6 * ref_p[6]
        assert abs(dist - ref_d) < 1e-4 * ref_d
        assert np.abs(gw - ref_g).max() < 1e-3 * np.abs(ref_g).max()


def test_sqrt_residual_and_clamps(harness):
    """Evaluation-mode residual (guard_sqrt before the mean) and the clamp masks: a sphere whose
    weighted radius falls under the 1e-3 clamp has no radius gradient."""
    P, Nn, w, gt = make_segment("sphere", 7)
    st, params, dist, gw, _, _ = run_harness(harness, P, Nn, w, "sphere", gt, sqrt_flag=1)
    ref_p, ref_d, ref_g = oracle("sphere", P, Nn, w, gt, torch.float64, sqrt=True)
    assert abs(dist - ref_d) < 2e-6 * ref_d
    assert np.abs(gw - ref_g).max() < 5e-5 * np.abs(ref_g).max()
    tiny = (P * 0.05).astype(np.float32)       # radius^2 ~ 2e-4 < 1e-3
    st, params, dist, gw, _, _ = run_harness(harness, tiny, Nn, w, "sphere", tiny[::2])
    ref_p, ref_d, ref_g = oracle("sphere", tiny, Nn, w, tiny[::2], torch.float64)
    assert abs(params[3] - np.sqrt(1e-3)) < 1e-12 and abs(ref_p[3] - np.sqrt(1e-3)) < 1e-12
    assert np.abs(gw - ref_g).max() < 5e-5 * np.abs(ref_g).max()
