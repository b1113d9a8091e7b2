"""Data layer of the segmentation trainers: the generator semantics of
src/dataset_segments.py:14-262 and the augmentation routines of src/augment_utils.py — host-side
numpy like the reference (a 10 000-point shape is 120 kB; the per-step upload is stream-ordered
through pinned memory, `_lib.h2d`), with numpy's RNG consumed in the reference's order so that a
seeded run draws the same shapes, noise and augmentations.

Storage: the reference reads `data/shapes/{train,val,test}_data.h5` with the datasets
`points (M,10000,3)`, `labels (M,10000)`, `normals (M,10000,3)`, `prim (M,10000)`
(dataset_segments.py:38-44).  `load_split` reads that schema from an `.h5` file when h5py is
importable (it is not in the build image) or from an `.npz` with the same four keys; arrays can
also be passed directly."""
import numpy as np

EPS = np.finfo(np.float32).eps


# ---------------------------------------------------------------------------------------
# augmentation (semantics of src/augment_utils.py) as ONE batched affine map per shape
# ---------------------------------------------------------------------------------------
# Four of the reference's five routines are affine maps of a whole shape (x -> x M + t with one
# M / t per shape) and the fifth adds clipped Gaussian noise.  Here a routine only DRAWS its
# parameters — from numpy's global RNG, in exactly the order the reference consumes it, so that a
# seeded run sees the same rotations, shifts and noise — and `Affine.apply` evaluates all shapes
# of the batch as one batched matrix product on whatever device the points live on (SURVEY
# section 8f-4: augmentation on the GPU).
def _axis_rotation(axis, angle):
    """3x3 rotation about a coordinate axis (0 = x, 1 = y, 2 = z), right-handed, float64."""
    c, s = np.cos(angle), np.sin(angle)
    i, j = (axis + 1) % 3, (axis + 2) % 3
    R = np.eye(3)
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


class Affine:
    """Per-shape maps x -> x M + t (+ noise): M (B,3,3), t (B,3), noise (B,N,3) or None."""

    def __init__(self, B):
        self.M = np.tile(np.eye(3), (B, 1, 1))
        self.t = np.zeros((B, 3))
        self.noise = None

    def then(self, M=None, t=None, noise=None):
        """Compose: this map first, then x -> x M + t + noise."""
        if M is not None:
            self.M = self.M @ M
            self.t = np.einsum("bi,bij->bj", self.t, M)
            if self.noise is not None:
                self.noise = np.einsum("bni,bij->bnj", self.noise, M)
        if t is not None:
            self.t = self.t + t
        if noise is not None:
            self.noise = noise if self.noise is None else self.noise + noise
        return self

    def apply(self, points):
        """points (B,N,3): numpy array or torch tensor (any device) -> same kind, float32."""
        import torch
        as_numpy = isinstance(points, np.ndarray)
        P = torch.from_numpy(np.ascontiguousarray(points)) if as_numpy else points
        dt = torch.float64 if P.device.type == "cpu" else torch.float32
        M = torch.from_numpy(self.M).to(P.device, dt)
        t = torch.from_numpy(self.t).to(P.device, dt)
        out = torch.baddbmm(t.unsqueeze(1), P.to(dt), M)
        if self.noise is not None:
            out = out + torch.from_numpy(self.noise).to(P.device, dt)
        out = out.float()
        return out.numpy() if as_numpy else out


def _draw_rotation_y(B, angle=None):
    """rotate_point_cloud / _by_angle (augment_utils.py:7-45): one uniform angle per shape about y."""
    M = np.stack([_axis_rotation(1, np.random.uniform() * 2 * np.pi if angle is None else angle)
                  for _ in range(B)])
    return dict(M=M)


def _draw_perturbation(B, angle_sigma=0.06, angle_clip=0.30):
    """rotate_perturbation_point_cloud (:48-70): three clipped normal angles per shape, x then y then z."""
    M = []
    for _ in range(B):
        ax, ay, az = np.clip(angle_sigma * np.random.randn(3), -angle_clip, angle_clip)
        M.append(_axis_rotation(2, az) @ _axis_rotation(1, ay) @ _axis_rotation(0, ax))
    return dict(M=np.stack(M))


def _draw_jitter(B, N, sigma=0.01, clip=0.05):
    if clip <= 0:
        raise ValueError("clip must be positive")
    return dict(noise=np.clip(sigma * np.random.randn(B, N, 3), -clip, clip))


def _draw_shift(B, shift_range=0.1):
    return dict(t=np.random.uniform(-shift_range, shift_range, (B, 3)))


def _draw_scale(B, scale_low=0.8, scale_high=1.2):
    return dict(M=np.random.uniform(scale_low, scale_high, B)[:, None, None] * np.eye(3)[None])


def _one(draw, batch_data, *a, **k):
    B, N, _ = batch_data.shape
    return Affine(B).then(**draw(B, *([N] if draw is _draw_jitter else []), *a, **k)).apply(batch_data)


def rotate_point_cloud(batch_data):
    return _one(_draw_rotation_y, batch_data)


def rotate_point_cloud_by_angle(batch_data, rotation_angle):
    return _one(_draw_rotation_y, batch_data, rotation_angle)


def rotate_perturbation_point_cloud(batch_data, angle_sigma=0.06, angle_clip=0.30):
    return _one(_draw_perturbation, batch_data, angle_sigma, angle_clip)


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05):
    return _one(_draw_jitter, batch_data, sigma, clip)


def shift_point_cloud(batch_data, shift_range=0.1):
    return _one(_draw_shift, batch_data, shift_range)


def random_scale_point_cloud(batch_data, scale_low=0.8, scale_high=1.2):
    return _one(_draw_scale, batch_data, scale_low, scale_high)


class Augment:
    """augment_utils.py:116-128: perturbation, jitter, shift (0.05), scale, each with probability
    0.3 in this order — composed into ONE affine map + noise, applied in one launch."""

    def augment(self, batch_data):
        B, N, _ = batch_data.shape
        A = Affine(B)
        if np.random.random() > 0.7:
            A.then(**_draw_perturbation(B))
        if np.random.random() > 0.7:
            A.then(**_draw_jitter(B, N))
        if np.random.random() > 0.7:
            A.then(**_draw_shift(B, 0.05))
        if np.random.random() > 0.7:
            A.then(**_draw_scale(B))
        return A.apply(batch_data)


# ---------------------------------------------------------------------------------------
# canonical frame (dataset_segments.py:257-279, 131-147): minor principal axis -> x, unit extent
# ---------------------------------------------------------------------------------------
def pca_numpy(X):
    S, U = np.linalg.eig(X.T @ X)
    return S, U


def rotation_matrix_a_to_b(A, B):
    """Rotation with B = R A for unit vectors, by Rodrigues' formula about A x B.  (Anti)parallel
    vectors give the identity — the reference's construction (a change of basis whose matrix is
    singular then, fitting_utils.py:556-577) falls back to the identity in both cases too."""
    A, B = np.asarray(A, dtype=np.float64), np.asarray(B, dtype=np.float64)
    axis = np.cross(A, B)
    s = np.linalg.norm(axis)
    if s < EPS:
        return np.eye(3, dtype=np.float32)
    k = axis / s
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + s * Kx + (1.0 - np.dot(A, B)) * (Kx @ Kx)


def canonical_frames(points):
    """(B,N,3) float array -> rotations (B,3,3) taking the minor principal axis of every shape to
    +x (LAPACK geev on the host decides the axis' sign, like the reference)."""
    out = []
    for p in points:
        S, U = pca_numpy(np.asarray(p, dtype=np.float32))
        out.append(rotation_matrix_a_to_b(U[:, np.argmin(S)], np.array([1.0, 0.0, 0.0])))
    return np.stack(out)


def _canonicalise_batch(points, normals, anisotropic):
    """Rotate every shape into its canonical frame and divide by its extent (largest extent unless
    anisotropic): one batched product for the whole batch.  Device tensors stay on the device: only
    the 3 x 3 second-moment matrices travel to the host, whose LAPACK ``geev`` decides the axis and
    its sign like the reference's."""
    if not isinstance(points, np.ndarray):
        import torch
        # X^T X per shape (pca_numpy), accumulated in fp64 and symmetrised: the fp32 product of the
        # GPU differs from the host's in the last bits and is not exactly symmetric, which for a
        # nearly isotropic shape can flip geev's minor axis or its sign (another canonical frame)
        Pd = points.double()
        G = torch.bmm(Pd.transpose(1, 2), Pd)
        G = (0.5 * (G + G.transpose(1, 2))).cpu().numpy()
        R = []
        for g in G:
            S, U = np.linalg.eig(g)
            R.append(rotation_matrix_a_to_b(U[:, np.argmin(S)], np.array([1.0, 0.0, 0.0])))
        A = Affine(points.shape[0]).then(M=np.transpose(np.stack(R), (0, 2, 1)))
        pts = A.apply(points)
        nrm = A.apply(normals) if normals is not None else None
        ext = pts.max(1)[0] - pts.min(1)[0]
        pts = pts / (ext[:, None, :] + EPS) if anisotropic else pts / (ext.max(1)[0][:, None, None] + EPS)
        return pts, nrm
    R = canonical_frames(points)
    A = Affine(points.shape[0]).then(M=np.transpose(R, (0, 2, 1)))
    pts = A.apply(points)
    nrm = A.apply(normals) if normals is not None else None
    ext = pts.max(1) - pts.min(1)
    pts = pts / (ext[:, None, :] + EPS) if anisotropic else pts / (ext.max(1)[:, None, None] + EPS)
    return pts.astype(np.float32), nrm


def normalize_points(points, normals, anisotropic=False):
    """dataset_segments.py:262-279 (used by test.py): centre, noise along the normals (one normal
    draw per point), canonical frame, unit extent."""
    points = points - np.mean(points, 0, keepdims=True)
    noise = normals * np.clip(np.random.randn(points.shape[0], 1) * 0.01, a_min=-0.01, a_max=0.01)
    pts, nrm = _canonicalise_batch((points + noise.astype(np.float32))[None], normals[None], anisotropic)
    return pts[0], nrm[0].astype(np.float32)


# ---------------------------------------------------------------------------------------
# storage + generators
# ---------------------------------------------------------------------------------------
def load_split(source, size=None):
    """{"points","labels","normals","prim"} arrays of one split from a dict, an .npz or an .h5."""
    if isinstance(source, dict):
        arrays = source
    elif str(source).endswith(".npz"):
        with np.load(source) as f:
            arrays = {k: f[k] for k in ("points", "labels", "normals", "prim") if k in f.files}
    else:
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - h5py is absent from the build image
            raise ImportError("reading %s needs h5py; convert the split to .npz (same four keys)" % source) from e
        with h5py.File(source, "r") as hf:
            arrays = {k: np.array(hf.get(k)) for k in ("points", "labels", "normals", "prim") if k in hf}
    out = {k: (v[0:size] if size is not None else v) for k, v in arrays.items()}
    if "points" not in out or "labels" not in out:
        raise KeyError("a split needs at least 'points' and 'labels'")
    return out


class Dataset:
    """dataset_segments.Dataset: per-split arrays, points centred per shape at load time, endless
    generators yielding [points, labels, normals | None, primitives | None]."""

    def __init__(self, batch_size, train=None, val=None, test=None, train_size=None, val_size=None,
                 test_size=None, normals=False, primitives=False, device=None):
        """``device``: keep points and normals of every split resident on that device and run the
        per-batch work there (gather of the shuffled batch, augmentation map, normal noise,
        canonical frame and extent): the generators then yield device tensors for points and
        normals.  The random draws stay on numpy's generator in the reference's order, so both
        modes see the same maps; labels and primitive types stay host arrays (the losses and the
        matching read them there)."""
        self.batch_size, self.normals, self.primitives = batch_size, normals, primitives
        self.device = device
        self.augment_routines = [rotate_perturbation_point_cloud, jitter_point_cloud, shift_point_cloud,
                                 random_scale_point_cloud, rotate_point_cloud]   # dataset_segments.py:88-94
        self.splits = {}
        for name, src, size in (("train", train, train_size), ("val", val, val_size), ("test", test, test_size)):
            if src is None:
                continue
            a = load_split(src, size)
            pts = a["points"].astype(np.float32)
            split = {"points": pts - np.expand_dims(np.mean(pts, 1), 1), "labels": a["labels"]}
            if normals:
                split["normals"] = a["normals"].astype(np.float32)
            if primitives:
                split["prim"] = a["prim"]
            if device is not None:
                import torch
                split["points"] = torch.from_numpy(split["points"]).to(device)
                if normals:
                    split["normals"] = torch.from_numpy(split["normals"]).to(device)
            self.splits[name] = split

    def _iterate(self, name, randomize, augment, anisotropic, align_canonical, if_normal_noise):
        s = self.splits[name]
        size, bs = s["points"].shape[0], self.batch_size
        while True:
            order = np.arange(size)
            if randomize:
                np.random.shuffle(order)
            on_device = self.device is not None
            if on_device:
                import torch

                def h2d(a, device):     # pinned + stream-ordered for a GPU (see _lib.h2d)
                    if torch.device(device).type == "cuda":
                        from ._lib import h2d as pinned
                        return pinned(a, device)
                    return torch.from_numpy(np.ascontiguousarray(a)).to(device)
                order_dev = h2d(order, self.device)
                pts_all = nrm_all = None
            else:
                pts_all = s["points"][order]
                nrm_all = s["normals"][order] if self.normals else None
            lab_all = s["labels"][order]
            prm_all = s["prim"][order] if self.primitives else None
            for i in range(size // bs):
                sl = slice(i * bs, (i + 1) * bs)
                if on_device:       # the batch is gathered on the device
                    points = s["points"][order_dev[sl]]
                    normals = s["normals"][order_dev[sl]] if self.normals else None
                else:
                    points = pts_all[sl]
                    normals = nrm_all[sl] if self.normals else None
                if augment:
                    points = self.augment_routines[np.random.choice(np.arange(5))](points)
                if if_normal_noise and self.normals:
                    amp = np.clip(np.random.randn(1, points.shape[1], 1) * 0.01, a_min=-0.01, a_max=0.01)
                    if on_device:
                        points = points + normals * h2d(amp.astype(np.float32), self.device)
                    else:
                        points = points + (normals * amp).astype(np.float32)
                if align_canonical:
                    points, normals = _canonicalise_batch(points, normals, anisotropic)
                yield [points, lab_all[sl], normals, prm_all[sl] if self.primitives else None]

    def get_train(self, randomize=False, augment=False, anisotropic=False, align_canonical=False,
                  if_normal_noise=False):
        return self._iterate("train", randomize, augment, anisotropic, align_canonical, if_normal_noise)

    def get_val(self, randomize=False, anisotropic=False, align_canonical=False, if_normal_noise=False):
        return self._iterate("val", False, False, anisotropic, align_canonical, if_normal_noise)

    def get_test(self, randomize=False, anisotropic=False, align_canonical=False, if_normal_noise=False):
        return self._iterate("test", False, False, anisotropic, align_canonical, if_normal_noise)

    # the reference exposes these as methods too
    normalize_points = staticmethod(normalize_points)
    rotation_matrix_a_to_b = staticmethod(rotation_matrix_a_to_b)
    pca_numpy = staticmethod(pca_numpy)


# ---------------------------------------------------------------------------------------
# SplineNet patches: src/dataset.py:27-260 (DataSetControlPointsPoisson)
# ---------------------------------------------------------------------------------------
class generator_iter:
    """dataset.py:13-24: lets a torch DataLoader pull from a generator."""

    def __init__(self, generator, train_size):
        self.generator, self.train_size = generator, train_size

    def __len__(self):
        return self.train_size

    def __getitem__(self, idx):
        return next(self.generator)


class DataSetControlPointsPoisson:
    """Points (M,P,3) sampled on spline patches + their control grids (M,u,v,3): shuffled once with
    numpy seed 0, split by position (open: 50 000 / 10 000 / rest; closed: 28 000 / 3 000 / rest —
    ``split_at`` overrides the positions for smaller sets), canonicalised per patch on the fly.
    Generators yield [Points, None, controlpoints, scales, RS] like the reference."""

    def __init__(self, path, batch_size, size_u=20, size_v=20, splits={}, closed=False, split_at=None):
        self.path, self.batch_size, self.size_u, self.size_v = path, batch_size, size_u, size_v
        self.train_size, self.val_size, self.test_size = splits["train"], splits["val"], splits["test"]
        if isinstance(path, dict):
            arrays = path
        elif str(path).endswith(".npz"):
            with np.load(path) as f:
                arrays = {"points": f["points"], "controlpoints": f["controlpoints"]}
        else:
            try:
                import h5py
            except ImportError as e:  # pragma: no cover
                raise ImportError("reading %s needs h5py; convert it to .npz (points, controlpoints)" % path) from e
            with h5py.File(path, "r") as hf:
                arrays = {"points": np.array(hf.get(name="points")), "controlpoints": np.array(hf.get(name="controlpoints"))}
        points = arrays["points"].astype(np.float32)
        control_points = arrays["controlpoints"].astype(np.float32)
        np.random.seed(0)
        order = np.arange(points.shape[0])
        np.random.shuffle(order)
        points, control_points = points[order], control_points[order]
        a, b = split_at if split_at is not None else ((28000, 31000) if closed else (50000, 60000))
        self.train_points, self.val_points, self.test_points = points[0:a], points[a:b], points[b:]
        self.train_control_points = control_points[0:a]
        self.val_control_points, self.test_control_points = control_points[a:b], control_points[b:]
        self._augment = Augment()

    rotation_matrix_a_to_b = staticmethod(rotation_matrix_a_to_b)
    pca_numpy = staticmethod(pca_numpy)

    def _batch(self, pts_all, cp_all, batch_id, align_canonical, anisotropic, if_augment, scale_iso=True):
        """One batch, all patches at once: centre on the patch mean, canonical frame (minor axis ->
        x), extent scaling — per axis (anisotropic) or by the largest extent; control grids follow
        their patch through the same map.  dataset.py:96-150 does this patch by patch."""
        sl = slice(batch_id * self.batch_size, (batch_id + 1) * self.batch_size)
        P = pts_all[sl].astype(np.float32)
        CP = cp_all[sl].astype(np.float32).reshape(self.batch_size, self.size_u * self.size_v, 3)
        mean = P.mean(1, keepdims=True)
        P, CP = P - mean, CP - mean
        RS = []
        if align_canonical:
            R = canonical_frames(P)
            A = Affine(self.batch_size).then(M=np.transpose(R, (0, 2, 1)))
            P, CP = A.apply(P), A.apply(CP)
            RS = list(R)
        ext = np.abs(P.max(1) - P.min(1))                                    # (B,3)
        if anisotropic:
            scales = [e.reshape((1, 3)) for e in ext]
            P, CP = P / (ext[:, None, :] + EPS), CP / (ext[:, None, :] + EPS)
        else:
            big = ext.max(1)
            scales = list(big)
            CP = CP / big[:, None, None]
            if scale_iso:   # the reference's test loader leaves the points unscaled here
                P = P / big[:, None, None]
        controlpoints = CP.reshape(self.batch_size, self.size_u, self.size_v, 3)
        if if_augment:
            P = self._augment.augment(P).astype(np.float32)
        return [P, None, controlpoints, scales, RS]

    def load_train_data(self, if_regular_points=False, align_canonical=False, anisotropic=False, if_augment=False):
        while True:
            for batch_id in range(self.train_size // self.batch_size - 1):
                yield self._batch(self.train_points, self.train_control_points, batch_id, align_canonical,
                                  anisotropic, if_augment)

    def load_val_data(self, if_regular_points=False, align_canonical=False, anisotropic=False, if_augment=False):
        while True:
            for batch_id in range(self.val_size // self.batch_size - 1):
                yield self._batch(self.val_points, self.val_control_points, batch_id, align_canonical,
                                  anisotropic, if_augment)

    def load_test_data(self, if_regular_points=False, align_canonical=False, anisotropic=False, if_augment=False):
        for batch_id in range(self.test_size // self.batch_size):
            yield self._batch(self.test_points, self.test_control_points, batch_id, align_canonical, anisotropic,
                              if_augment, scale_iso=False)   # dataset.py:238-239
