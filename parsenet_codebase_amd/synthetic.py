"""Deterministic synthetic inputs with the statistics of the reference's data pipeline.

No dataset ships with the reference, so benchmarks, smoke tests and
fixtures use shapes assembled from analytic patches — planes, spheres, cylinders, cones,
open bicubic patches and closed (u-periodic) tubes — with exact unit normals, segment labels
and the primitive-type ids of readme_data.md:42-47, followed by the reference's
normalisation (src/dataset_segments.py:257-274: centre, normal-direction noise clipped to
+-0.01, PCA-align the minor axis to x, divide by the largest extent).  numpy only; this is
input preparation, not part of the timed path.
"""
import numpy as np

EPS = np.finfo(np.float32).eps

PRIM_PLANE, PRIM_OPEN, PRIM_CONE, PRIM_CYL, PRIM_SPHERE, PRIM_CLOSED = 1, 2, 3, 4, 5, 9


def _frame(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _plane(rng, n):
    R = _frame(rng)
    ext = rng.uniform(0.3, 1.0, 2)
    uv = rng.uniform(-0.5, 0.5, (n, 2)) * ext
    p = uv[:, :1] * R[:, 0] + uv[:, 1:] * R[:, 1] + rng.uniform(-0.5, 0.5, 3)
    return p, np.tile(R[:, 2], (n, 1))


def _sphere(rng, n):
    c, r = rng.uniform(-0.5, 0.5, 3), rng.uniform(0.2, 0.6)
    R = _frame(rng)
    cap = rng.uniform(0.3, 1.0)                         # fraction of the polar range
    z = 1.0 - rng.uniform(0, 1, n) * 2.0 * cap
    phi = rng.uniform(0, 2 * np.pi, n)
    s = np.sqrt(np.clip(1 - z * z, 0, 1))
    d = np.stack([s * np.cos(phi), s * np.sin(phi), z], 1) @ R.T
    return c + r * d, d


def _cylinder(rng, n):
    R, c = _frame(rng), rng.uniform(-0.5, 0.5, 3)
    r, hgt = rng.uniform(0.1, 0.4), rng.uniform(0.4, 1.2)
    phi = rng.uniform(0, rng.uniform(np.pi, 2 * np.pi), n)
    t = rng.uniform(-0.5, 0.5, n) * hgt
    d = np.stack([np.cos(phi), np.sin(phi), np.zeros(n)], 1) @ R.T
    return c + r * d + t[:, None] * R[:, 2], d


def _cone(rng, n):
    R, apex = _frame(rng), rng.uniform(-0.5, 0.5, 3)
    theta = rng.uniform(0.2, 1.0)
    t = np.sqrt(rng.uniform(0.04, 1.0, n)) * rng.uniform(0.5, 1.2)   # distance along the axis
    phi = rng.uniform(0, 2 * np.pi, n)
    radial = np.stack([np.cos(phi), np.sin(phi), np.zeros(n)], 1) @ R.T
    p = apex + t[:, None] * (np.cos(theta) * R[:, 2] + np.sin(theta) * radial) / np.cos(theta)
    nrm = np.cos(theta) * radial - np.sin(theta) * R[:, 2]
    return p, nrm


def _bernstein3(t):
    return np.stack([(1 - t) ** 3, 3 * t * (1 - t) ** 2, 3 * t * t * (1 - t), t ** 3], 1)


def _dbernstein3(t):
    return np.stack([-3 * (1 - t) ** 2, 3 * (1 - t) ** 2 - 6 * t * (1 - t), 6 * t * (1 - t) - 3 * t * t,
                     3 * t * t], 1)


def _open_patch(rng, n):
    g = np.stack(np.meshgrid(np.linspace(-0.5, 0.5, 4), np.linspace(-0.5, 0.5, 4), indexing="ij"), -1)
    ctrl = np.concatenate([g, rng.uniform(-0.25, 0.25, (4, 4, 1))], -1) * rng.uniform(0.5, 1.2)
    ctrl = ctrl @ _frame(rng).T + rng.uniform(-0.4, 0.4, 3)
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    bu, bv, du, dv = _bernstein3(u), _bernstein3(v), _dbernstein3(u), _dbernstein3(v)
    p = np.einsum("ni,nj,ijk->nk", bu, bv, ctrl)
    pu = np.einsum("ni,nj,ijk->nk", du, bv, ctrl)
    pv = np.einsum("ni,nj,ijk->nk", bu, dv, ctrl)
    nrm = np.cross(pu, pv)
    return p, nrm / (np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-12)


def _closed_tube(rng, n):
    R, c = _frame(rng), rng.uniform(-0.5, 0.5, 3)
    a0, a1, w = rng.uniform(0.15, 0.35), rng.uniform(0.02, 0.1), rng.uniform(2.0, 6.0)
    hgt = rng.uniform(0.5, 1.2)
    phi, t = rng.uniform(0, 2 * np.pi, n), rng.uniform(-0.5, 0.5, n) * hgt
    r = a0 + a1 * np.sin(w * t)
    dr = a1 * w * np.cos(w * t)
    radial = np.stack([np.cos(phi), np.sin(phi), np.zeros(n)], 1)
    p = np.concatenate([r[:, None] * radial[:, :2], t[:, None]], 1)
    nrm = np.concatenate([radial[:, :2], -dr[:, None]], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return c + p @ R.T, nrm @ R.T


_MAKERS = [(_plane, PRIM_PLANE), (_sphere, PRIM_SPHERE), (_cylinder, PRIM_CYL), (_cone, PRIM_CONE),
           (_open_patch, PRIM_OPEN), (_closed_tube, PRIM_CLOSED)]


def rotation_a_to_b(A, B):
    """Rotation taking unit vector A to B (src/dataset_segments.py:276-298)."""
    cos = np.dot(A, B)
    sin = np.linalg.norm(np.cross(B, A))
    u = A
    v = B - np.dot(A, B) * A
    v = v / (np.linalg.norm(v) + EPS)
    w = np.cross(B, A)
    w = w / (np.linalg.norm(w) + EPS)
    Fm = np.stack([u, v, w], 1)
    G = np.array([[cos, -sin, 0], [sin, cos, 0], [0, 0, 1]])
    try:
        return Fm @ G @ np.linalg.inv(Fm)
    except np.linalg.LinAlgError:
        return np.eye(3)


def normalize_points(points, normals, rng):
    """src/dataset_segments.py:257-274 (isotropic branch)."""
    points = points - points.mean(0, keepdims=True)
    noise = normals * np.clip(rng.standard_normal((points.shape[0], 1)) * 0.01, -0.01, 0.01)
    points = points + noise
    S, U = np.linalg.eig(points.T @ points)
    Rm = rotation_a_to_b(np.real(U[:, np.argmin(np.real(S))]), np.array([1.0, 0, 0]))
    points = (Rm @ points.T).T
    normals = (Rm @ normals.T).T
    std = points.max(0) - points.min(0)
    points = points / (std.max() + EPS)
    return points.astype(np.float32), normals.astype(np.float32)


def make_shape(shape_id, num_points=10000, min_segments=4, max_segments=12):
    """One synthetic shape: points (N,3), normals (N,3) fp32, labels (N,), primitives (N,) int64."""
    rng = np.random.RandomState(1234 + int(shape_id))
    S = rng.randint(min_segments, max_segments + 1)
    w = rng.uniform(0.5, 1.5, S)
    counts = np.maximum((w / w.sum() * num_points).astype(int), 60)
    counts[np.argmax(counts)] += num_points - counts.sum()
    pts, nrms, labs, prims = [], [], [], []
    for s in range(S):
        maker, prim = _MAKERS[rng.randint(len(_MAKERS))]
        p, n = maker(rng, counts[s])
        pts.append(p)
        nrms.append(n)
        labs.append(np.full(counts[s], s))
        prims.append(np.full(counts[s], prim))
    pts, nrms = np.concatenate(pts), np.concatenate(nrms)
    labs, prims = np.concatenate(labs), np.concatenate(prims)
    order = rng.permutation(num_points)
    pts, nrms = normalize_points(pts[order], nrms[order], rng)
    return pts, nrms, labs[order].astype(np.int64), prims[order].astype(np.int64)


def make_batch(first_id, batch, num_points=10000, **kw):
    """(points (B,N,3), normals (B,N,3), labels (B,N), primitives (B,N)) for shape ids
    first_id .. first_id + batch - 1."""
    items = [make_shape(first_id + i, num_points, **kw) for i in range(batch)]
    return tuple(np.stack([it[j] for it in items], 0) for j in range(4))


def make_batch_ids(ids, num_points=10000, **kw):
    """make_batch for an explicit list of shape ids."""
    items = [make_shape(i, num_points, **kw) for i in ids]
    return tuple(np.stack([it[j] for it in items], 0) for j in range(4))


# shape ids in [0, 3000) whose patches are planes, spheres and cones only: every fit of the fitting
# stage is then well conditioned (no cylinder: its circle fit always takes the reference's noisy
# fp32 ridge branch; no SplineNet: kNN near-ties) — whole-step gradient parity tests use them
ANALYTIC_WELL_POSED_IDS = (68, 160, 172, 200, 382, 395, 436, 486, 633, 821, 984, 1022, 1341, 1360, 1542, 1569)


def make_spline_patches(first_id, batch, num_points=700, grid=20, closed=False):
    """SplineNet inputs (cfg1-3): points (B,N,3) sampled on random smooth bicubic control grids
    (grid x grid x 3, returned as the regression target), canonicalised like
    src/dataset.py:115-131 (centre, PCA-align, scale to the unit box)."""
    P, CP = [], []
    for i in range(batch):
        rng = np.random.RandomState(4321 + first_id + i)
        coarse = rng.uniform(-0.5, 0.5, (5, 5, 3)) * 0.35
        g = np.stack(np.meshgrid(np.linspace(-0.5, 0.5, 5), np.linspace(-0.5, 0.5, 5), indexing="ij"), -1)
        coarse[..., :2] += g
        if closed:
            ang = np.linspace(0, 2 * np.pi, 5, endpoint=False)
            rad = 0.3 + 0.1 * rng.uniform(-1, 1, (5, 5))
            coarse = np.stack([rad * np.cos(ang)[:, None], rad * np.sin(ang)[:, None],
                               np.linspace(-0.5, 0.5, 5)[None, :].repeat(5, 0)], -1)
        # smooth interpolation of the coarse grid to grid x grid control points
        t = np.linspace(0, 1, grid)
        if closed:
            # periodic in u: piecewise-linear wrap-around in u, Bezier-like blend in v
            tu = t * 5.0
            i0 = np.floor(tu).astype(int) % 5
            fr = (tu - np.floor(tu))[:, None, None]
            rows = coarse[i0] * (1 - fr) + coarse[(i0 + 1) % 5] * fr
        else:
            bu = np.stack([np.interp(t, np.linspace(0, 1, 5), coarse[:, j, c]) for j in range(5) for c in range(3)], 1)
            rows = bu.reshape(grid, 5, 3)
        ctrl = np.stack([np.stack([np.interp(t, np.linspace(0, 1, 5), rows[a, :, c]) for c in range(3)], 1)
                         for a in range(grid)], 0)                       # (grid, grid, 3)
        # sample the surface spanned by the control net with bilinear interpolation
        u, v = rng.uniform(0, grid - 1.001, num_points), rng.uniform(0, grid - 1.001, num_points)
        ui, vi = u.astype(int), v.astype(int)
        fu, fv = (u - ui)[:, None], (v - vi)[:, None]
        pts = (ctrl[ui, vi] * (1 - fu) * (1 - fv) + ctrl[ui + 1, vi] * fu * (1 - fv) +
               ctrl[ui, vi + 1] * (1 - fu) * fv + ctrl[ui + 1, vi + 1] * fu * fv)
        mean = pts.mean(0)
        pts, c2 = pts - mean, ctrl - mean
        S, U = np.linalg.eig(pts.T @ pts)
        Rm = rotation_a_to_b(np.real(U[:, np.argmin(np.real(S))]), np.array([1.0, 0, 0]))
        pts, c2 = (Rm @ pts.T).T, (Rm @ c2.reshape(-1, 3).T).T.reshape(grid, grid, 3)
        scale = (pts.max(0) - pts.min(0)).max() + EPS
        P.append((pts / scale).astype(np.float32))
        CP.append((c2 / scale).astype(np.float32))
    return np.stack(P, 0), np.stack(CP, 0)
