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
# src/augment_utils.py
# ---------------------------------------------------------------------------------------
def _rot_y(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotate_point_cloud(batch_data):
    """Random rotation about the up (y) axis, one angle per shape (augment_utils.py:7-26)."""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        out[k, ...] = np.dot(batch_data[k, ...].reshape((-1, 3)), _rot_y(np.random.uniform() * 2 * np.pi))
    return out.astype(np.float32)


def rotate_point_cloud_by_angle(batch_data, rotation_angle):
    out = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        out[k, ...] = np.dot(batch_data[k, ...].reshape((-1, 3)), _rot_y(rotation_angle))
    return out.astype(np.float32)


def rotate_perturbation_point_cloud(batch_data, angle_sigma=0.06, angle_clip=0.30):
    """Small random rotation Rz Ry Rx per shape (augment_utils.py:49-73)."""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        a = np.clip(angle_sigma * np.random.randn(3), -angle_clip, angle_clip)
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        out[k, ...] = np.dot(batch_data[k, ...].reshape((-1, 3)), np.dot(Rz, np.dot(Ry, Rx)))
    return out.astype(np.float32)


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05):
    B, N, C = batch_data.shape
    if clip <= 0:
        raise ValueError("clip must be positive")
    jittered = np.clip(sigma * np.random.randn(B, N, C), -1 * clip, clip)
    jittered += batch_data
    return jittered.astype(np.float32)


def shift_point_cloud(batch_data, shift_range=0.1):
    """One random shift per shape; modifies its argument in place like the reference."""
    B = batch_data.shape[0]
    shifts = np.random.uniform(-shift_range, shift_range, (B, 3))
    for b in range(B):
        batch_data[b, :, :] += shifts[b, :]
    return batch_data.astype(np.float32)


def random_scale_point_cloud(batch_data, scale_low=0.8, scale_high=1.2):
    """One random scale per shape; in place like the reference."""
    B = batch_data.shape[0]
    scales = np.random.uniform(scale_low, scale_high, B)
    for b in range(B):
        batch_data[b, :, :] *= scales[b]
    return batch_data


class Augment:
    """augment_utils.py:122-135: each routine with probability 0.3, in this order."""

    def augment(self, batch_data):
        if np.random.random() > 0.7:
            batch_data = rotate_perturbation_point_cloud(batch_data)
        if np.random.random() > 0.7:
            batch_data = jitter_point_cloud(batch_data)
        if np.random.random() > 0.7:
            batch_data = shift_point_cloud(batch_data, 0.05)
        if np.random.random() > 0.7:
            batch_data = random_scale_point_cloud(batch_data)
        return batch_data


# ---------------------------------------------------------------------------------------
# canonicalisation helpers (dataset_segments.py:262-310)
# ---------------------------------------------------------------------------------------
def pca_numpy(X):
    S, U = np.linalg.eig(X.T @ X)
    return S, U


def rotation_matrix_a_to_b(A, B):
    """Rotation with B = R A for unit vectors (identity when they are parallel)."""
    cos = np.dot(A, B)
    sin = np.linalg.norm(np.cross(B, A))
    u = A
    v = B - np.dot(A, B) * A
    v = v / (np.linalg.norm(v) + EPS)
    w = np.cross(B, A)
    w = w / (np.linalg.norm(w) + EPS)
    F = np.stack([u, v, w], 1)
    G = np.array([[cos, -sin, 0], [sin, cos, 0], [0, 0, 1]])
    try:
        return F @ G @ np.linalg.inv(F)
    except np.linalg.LinAlgError:
        return np.eye(3, dtype=np.float32)


def _canonicalise(points, normals, anisotropic):
    """In place on one shape: minor principal axis -> x, divide by the (largest) extent."""
    S, U = pca_numpy(points)
    R = rotation_matrix_a_to_b(U[:, np.argmin(S)], np.array([1, 0, 0]))
    points[...] = (R @ points.T).T
    if normals is not None:
        normals[...] = (R @ normals.T).T
    std = np.max(points, 0) - np.min(points, 0)
    points[...] = points / (std.reshape((1, 3)) + EPS) if anisotropic else points / (np.max(std) + EPS)


def normalize_points(points, normals, anisotropic=False):
    """dataset_segments.py:262-279 (used by test.py): centre, noise along the normals (one normal
    draw per point), canonicalise."""
    points = points - np.mean(points, 0, keepdims=True)
    noise = normals * np.clip(np.random.randn(points.shape[0], 1) * 0.01, a_min=-0.01, a_max=0.01)
    points = points + noise.astype(np.float32)
    S, U = pca_numpy(points)
    R = rotation_matrix_a_to_b(U[:, np.argmin(S)], np.array([1, 0, 0]))
    points = (R @ points.T).T
    normals = (R @ normals.T).T
    std = np.max(points, 0) - np.min(points, 0)
    points = points / (std.reshape((1, 3)) + EPS) if anisotropic else points / (np.max(std) + EPS)
    return points.astype(np.float32), normals.astype(np.float32)


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
                 test_size=None, normals=False, primitives=False):
        self.batch_size, self.normals, self.primitives = batch_size, normals, primitives
        self.augment_routines = [rotate_perturbation_point_cloud, jitter_point_cloud, shift_point_cloud,
                                 random_scale_point_cloud, rotate_point_cloud]
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
            self.splits[name] = split

    def _iterate(self, name, randomize, augment, anisotropic, align_canonical, if_normal_noise):
        s = self.splits[name]
        size, bs = s["points"].shape[0], self.batch_size
        while True:
            order = np.arange(size)
            if randomize:
                np.random.shuffle(order)
            pts_all, lab_all = s["points"][order], s["labels"][order]
            nrm_all = s["normals"][order] if self.normals else None
            prm_all = s["prim"][order] if self.primitives else None
            for i in range(size // bs):
                sl = slice(i * bs, (i + 1) * bs)
                points = pts_all[sl]
                normals = nrm_all[sl] if self.normals else None
                if augment:
                    points = self.augment_routines[np.random.choice(np.arange(5))](points)
                if if_normal_noise and self.normals:
                    noise = normals * np.clip(np.random.randn(1, points.shape[1], 1) * 0.01, a_min=-0.01, a_max=0.01)
                    points = points + noise.astype(np.float32)
                if align_canonical:
                    if points.base is not None or not points.flags.writeable:
                        points = np.array(points)     # the reference writes into its slices
                    if normals is not None:
                        normals = np.array(normals)
                    for j in range(bs):
                        _canonicalise(points[j], normals[j] if normals is not None else None, anisotropic)
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
        Points, controlpoints, scales, RS = [], [], [], []
        for i in range(self.batch_size):
            points = pts_all[batch_id * self.batch_size + i]
            mean = np.mean(points, 0)
            points = points - mean
            R = None
            if align_canonical:
                S, U = pca_numpy(points)
                R = rotation_matrix_a_to_b(U[:, np.argmin(S)], np.array([1, 0, 0]))
                points = (R @ points.T).T
                RS.append(R)
            if anisotropic:
                std = np.abs(np.max(points, 0) - np.min(points, 0)).reshape((1, 3))
                points = points / (std + EPS)
            else:
                std = np.max(np.max(points, 0) - np.min(points, 0))
                if scale_iso:   # the reference's test loader leaves the points unscaled here
                    points = points / std
            scales.append(std)
            Points.append(points)
            cp = cp_all[batch_id * self.batch_size + i] - mean.reshape((1, 1, 3))
            if align_canonical:
                cp = np.reshape((R @ cp.reshape((self.size_u * self.size_v, 3)).T).T, (self.size_u, self.size_v, 3))
            cp = cp / (std.reshape((1, 1, 3)) + EPS) if anisotropic else cp / std
            controlpoints.append(cp)
        controlpoints, Points = np.stack(controlpoints, 0), np.stack(Points, 0)
        if if_augment:
            Points = self._augment.augment(Points).astype(np.float32)
        return [Points, None, controlpoints, scales, RS]

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
