"""Synthetic workload generator (SURVEY.md §8d "Synthetic inputs").

No datasets, detections or checkpoints exist in the build environment, so the parity tests and
``bench.py`` run on seeded synthetic ROIs that have the shapes, dtypes and value ranges of the
reference's test-time batch (``batch_data_test``, engine_utils.py:213-241) and of the network
outputs consumed by the evaluator (gdrn_evaluator.py:461-573).

The renderer is injected (``render_fn``) so that tests may use the CPU oracle and the benchmark
the HIP rasteriser; this module itself depends on NumPy only.
"""
from __future__ import annotations

import numpy as np

# ref/ycbv.py:83-89 camera; 640x480
YCBV_K = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]], np.float32)
LMO_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], np.float32)
IM_W, IM_H = 640, 480
DATASET_NUM_CLASSES = {"lmo": 8, "ycbv": 21, "tless": 30, "icbin": 2, "hb": 16, "itodd": 28, "tudl": 3}


def icosphere(subdiv: int = 4):
    """Unit icosphere: 10*4^s+2 vertices, 20*4^s faces (2562 / 5120 at subdivision 4)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    verts = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    faces = [tuple(x) for x in f]
    for _ in range(subdiv):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.asarray(verts, np.float64), np.asarray(faces, np.int32)


def make_models(num_classes: int, rng: np.random.Generator, subdiv: int = 4):
    """Per class an ellipsoid with semi-axes extent/2, extent_c ~ U(0.05, 0.25) m, vertex order shuffled."""
    sv, sf = icosphere(subdiv)
    verts, faces, extents = [], [], []
    for _ in range(num_classes):
        ext = rng.uniform(0.05, 0.25, 3)
        perm = rng.permutation(len(sv))
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(sv))
        v = (sv * (ext / 2.0))[perm].astype(np.float32)
        f = inv[sf].astype(np.int32)
        verts.append(v)
        faces.append(f)
        # extent as the data loader computes it: size of the vertex bbox (data_loader.py _get_extents)
        extents.append((v.max(0) - v.min(0)).astype(np.float32))
    return verts, faces, np.stack(extents).astype(np.float32)


def random_rotation(rng: np.random.Generator) -> np.ndarray:
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_detections(b: int, num_classes: int, extents: np.ndarray, rng: np.random.Generator, K=YCBV_K,
                    out_res: int = 64, dzi_pad_scale: float = 1.5):
    """Detections + ground-truth poses.  Box/scale rules follow data_loader.py:754-769."""
    f32 = np.float32
    centers = np.stack([rng.uniform(80, 560, b), rng.uniform(80, 400, b)], 1)
    bw, bh = rng.uniform(40, 200, b), rng.uniform(40, 200, b)
    scale = np.minimum(np.maximum(bw, bh) * dzi_pad_scale, max(IM_H, IM_W)) * 1.0
    cls = rng.integers(0, num_classes, b)
    score = rng.uniform(0.3, 1.0, b)
    R = np.stack([random_rotation(rng) for _ in range(b)])
    # depth such that the projected object roughly fills the detection box
    tz = np.clip(K[0, 0] * extents[cls].max(1) / np.maximum(bw, bh), 0.3, 3.0)
    off = rng.uniform(-5, 5, (b, 2))
    tx = (centers[:, 0] + off[:, 0] - K[0, 2]) * tz / K[0, 0]
    ty = (centers[:, 1] + off[:, 1] - K[1, 2]) * tz / K[1, 1]
    t = np.stack([tx, ty, tz], 1)
    return dict(
        roi_cls=cls.astype(np.int64), score=score.astype(f32), roi_center=centers.astype(f32),
        roi_wh=np.stack([bw, bh], 1).astype(f32), scale=scale.astype(f32),
        resize_ratio=(out_res / scale).astype(f32), roi_cam=np.repeat(K[None], b, 0).astype(f32),
        roi_extent=extents[cls].astype(f32), R_gt=R.astype(f32), t_gt=t.astype(f32),
        im_W=np.full(b, IM_W, f32), im_H=np.full(b, IM_H, f32),
    )


def zoom_K_np(K, centers, scales, out_res):
    """camera_geometry.py:6-21 as called at engine_utils.py:260-264 (float32)."""
    K = np.asarray(K, np.float32)
    s = np.asarray(scales, np.float32).reshape(-1, 1)
    crop_xy = np.asarray(centers, np.float32) - s / np.float32(2)
    r = np.float32(out_res) / s
    out = K.copy()
    out[:, [0, 1], 2] = K[:, [0, 1], 2] - crop_xy
    out[:, [0, 1]] = out[:, [0, 1]] * r.reshape(-1, 1, 1)
    return out


def coord2d_roi(centers, scales, out_res=64, im_w=IM_W, im_h=IM_H):
    """Analytic roi_coord_2d: get_2d_coord_np (data_utils.py:304-323, linspace(0,1,endpoint=False)) sampled by the
    ROI affine at the output pixel centres — the value cv2.warpAffine interpolates in the image interior
    (data_loader.py:792-797).  f32[b,2,out,out]."""
    b = len(scales)
    j = np.arange(out_res, dtype=np.float64)
    out = np.zeros((b, 2, out_res, out_res), np.float32)
    for i in range(b):
        a = scales[i] / out_res
        xs = centers[i, 0] + (j - out_res * 0.5) * a
        ys = centers[i, 1] + (j - out_res * 0.5) * a
        out[i, 0] = np.clip(xs / im_w, 0, None)[None, :]
        out[i, 1] = np.clip(ys / im_h, 0, None)[:, None]
    return out


def box3(x: np.ndarray) -> np.ndarray:
    p = np.pad(x, 1, mode="edge")
    return sum(p[i:i + x.shape[0], j:j + x.shape[1]] for i in range(3) for j in range(3)) / 9.0


def make_map_inputs(det: dict, verts, faces, render_fn, rng: np.random.Generator, out_res: int = 64,
                    tz_sigma: float = 0.03):
    """Map-space inputs for the post-processing path (independent of network weights).

    render_fn(obj, K[b,3,3] f32, R[b,3,3] f32, t[b,3] f32, res) -> (depth f32[b,res,res], xyz f32[b,res,res,3])
    """
    f32 = np.float32
    b = len(det["scale"])
    K_crop = zoom_K_np(det["roi_cam"], det["roi_center"], det["scale"], out_res)
    depth, xyz = render_fn(det["roi_cls"].astype(np.int32), K_crop, det["R_gt"], det["t_gt"], out_res)
    vis = depth > 0
    ext = det["roi_extent"]
    xyz_n = xyz / ext[:, None, None, :] + 0.5 + rng.normal(0, 0.01, xyz.shape)
    xyz_n = np.where(vis[..., None], xyz_n, 0.0).astype(f32)
    coor = [np.ascontiguousarray(xyz_n[..., c][:, None]) for c in range(3)]
    mask = np.stack([box3(v.astype(np.float64)) for v in vis]) + rng.normal(0, 0.05, vis.shape)
    a = rng.uniform(0.5, 3.0, (b, 1, 1))
    c = rng.uniform(-1.0, 1.0, (b, 1, 1))
    mask_raw = (mask * a + c).astype(f32)[:, None]
    big = np.repeat(np.repeat(depth, 4, axis=1), 4, axis=2).astype(np.float64)
    noisy = big + rng.normal(0, 0.002, big.shape)
    noisy = np.where(big > 0, noisy, 0.0)
    drop = rng.uniform(0, 1, big.shape) < 0.05
    roi_depth = np.where(drop, 0.0, noisy).astype(f32)[:, None]
    t_init = det["t_gt"].copy()
    t_init[:, 2] += rng.normal(0, tz_sigma, b).astype(f32)
    return dict(coor_x=coor[0], coor_y=coor[1], coor_z=coor[2], mask=mask_raw, roi_depth=roi_depth,
                K_crop=K_crop.astype(f32), t_init=t_init.astype(f32),
                roi_coord_2d=coord2d_roi(det["roi_center"], det["scale"], out_res))


# ---- platform-independent seeded values (network parity fixtures) -------------------------------------------------------
def seeded_uniform(name: str, shape, seed: int = 0) -> np.ndarray:
    """f32 array of ``shape`` with values in [-1, 1) that depend only on (name, seed, flat index): splitmix64 of a counter
    keyed by sha256(name) — no library RNG stream involved, so the recipe that wrote a fixture in the authoring container
    and the test that rebuilds its inputs on the GPU box get bit-identical tensors."""
    import hashlib
    key = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little")
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(key)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)   # 24 bits: exact in f32
    return u.reshape(shape)


def seeded_param(name: str, shape, seed: int = 0) -> np.ndarray:
    """Seeded value of one network parameter/buffer, scaled by its role so that activations stay O(1) through the whole
    graph (the reference's std=0.001 initialisers would make every map ~1e-20 and hide mistakes):
    >= 2-D weights U(-a, a) with a = sqrt(6 / fan_in) (variance 2 / fan_in); 1-D ``weight`` (norm scales) 1 + 0.2 u;
    ConvNeXt layer scale ``gamma`` 0.4 + 0.2 u; ``running_var`` 1 + 0.2 u; everything else 1-D (biases, means) 0.1 u."""
    shape = tuple(int(s) for s in shape)
    u = seeded_uniform(name, shape, seed)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (u * np.float32(np.sqrt(6.0 / fan_in))).astype(np.float32)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "gamma":
        return (np.float32(0.4) + np.float32(0.2) * u).astype(np.float32)
    if leaf in ("weight", "running_var"):
        return (np.float32(1.0) + np.float32(0.2) * u).astype(np.float32)
    return (np.float32(0.1) * u).astype(np.float32)


def seeded_state_dict(named_shapes, seed: int = 0, alias=None) -> dict:
    """{key: torch tensor} for ``named_shapes`` = iterable of (key, shape).  ``alias(key)`` maps a key to the name its
    value is derived from (the reference's ConvModule registers its norm twice: ``norm.*`` and ``gn.*`` are one tensor)."""
    import torch
    out = {}
    for k, shp in named_shapes:
        out[k] = torch.from_numpy(seeded_param(alias(k) if alias else k, shp, seed))
    return out
