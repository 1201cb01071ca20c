"""Shared by tests/golden/make_golden_net.py (authoring container, runs the reference's modules) and the tests that replay
its fixtures: the ROI batch of the network parity case and the seeded parameters, both platform-independent."""
import json
import os

import numpy as np
import torch

from gdrnpp_bop2022_amd import synthetic as S

SEED = 20220925
B = 4
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def norm_alias(key):
    """The reference's ConvModule registers one norm module under two names (conv_module.py:175-182): ``norm.*`` and
    ``gn.*`` are the same tensor, so both get the values derived from the ``gn`` name."""
    if key.startswith("geo_head_net.") and ".norm." in key:
        return key.replace(".norm.", ".gn.")
    return key


def net_image(b=B, seed=SEED):
    return (S.seeded_uniform("roi_img", (b, 3, 256, 256), seed) + np.float32(1)) * np.float32(0.5)


def net_detections(num_classes, b=B, seed=SEED):
    """Authoring-side only (NumPy Generator stream): the arrays are stored in the fixture."""
    rng = np.random.default_rng(seed + num_classes)
    ext = rng.uniform(0.05, 0.25, (num_classes, 3)).astype(np.float32)
    det = S.make_detections(b, num_classes, ext, rng)
    det["roi_cls"] = np.resize(np.array([0, num_classes - 1, 7 % num_classes, 13 % num_classes], np.int64), b)
    det["roi_extent"] = ext[det["roi_cls"]]
    return det


def net_detections_b128(num_classes, b=128, seed=SEED):
    """The ROI batch of the benchmark-size fixtures (net_golden_<ds>_b128.npz): every class occurs, roi_cls = i mod C."""
    rng = np.random.default_rng(seed + 1000 + num_classes)
    ext = rng.uniform(0.05, 0.25, (num_classes, 3)).astype(np.float32)
    det = S.make_detections(b, num_classes, ext, rng)
    det["roi_cls"] = (np.arange(b) % num_classes).astype(np.int64)
    det["roi_extent"] = ext[det["roi_cls"]]
    return det


def net_detections_large(num_classes, b, seed=SEED):
    """The ROI batches of the iteration-size fixtures (net_golden_tless_b1024.npz, net_golden_ycbv_b512.npz): their own generator
    stream, every class present (roi_cls = i mod C, unsorted)."""
    rng = np.random.default_rng(seed + 5000 + b + num_classes)
    ext = rng.uniform(0.05, 0.25, (num_classes, 3)).astype(np.float32)
    det = S.make_detections(b, num_classes, ext, rng)
    det["roi_cls"] = (np.arange(b) % num_classes).astype(np.int64)
    det["roi_extent"] = ext[det["roi_cls"]]
    return det


def cfg_name(ds):
    """Fixture name -> named config: "ycbvso" is a single-object config (class-agnostic head, configs/gdrn/ycbvSO/...)."""
    return f"{ds[:-2]}_convnext_so" if ds.endswith("so") else f"{ds}_convnext_a6"


def load_fixture(ds):
    z = np.load(os.path.join(GOLDEN, f"net_golden_{ds}.npz"))
    fx = {k: z[k] for k in z.files}
    fx["cfg"] = json.loads(str(fx.pop("cfg_json")))
    fx["head_keys"] = [(k, tuple(s)) for k, s in json.loads(str(fx.pop("head_keys")))]
    return fx


def load_f64_fixture(ds):
    """net_golden_<ds>_b128_f64.npz: the reference's module evaluated in fp64 on the 128-ROI batch (make_golden_net.py
    record_b128_f64) + the per-ROI distance of its own fp32 forward (the b128 fixture) from that."""
    name = f"net_golden_{ds}_f64.npz" if ds == "lmo_resnet34" else f"net_golden_{ds}_b128_f64.npz"
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def seeded_reference_state_dict(model, fx):
    """The state_dict the reference model held when the fixture was recorded: geo head + Patch-PnP entries by the
    REFERENCE's key/shape manifest (incl. its duplicate ``norm.*`` keys), backbone entries by this model's own keys."""
    if "backbone_keys" in fx:       # ResNet fixtures also carry the reference-side backbone manifest (BatchNorm buffers)
        named = [(k, tuple(s)) for k, s in json.loads(str(fx["backbone_keys"]))]
    else:
        named = [(k, tuple(v.shape)) for k, v in model.state_dict().items() if k.startswith("backbone.")]
    named += list(fx["head_keys"])
    return S.seeded_state_dict(named, SEED, alias=norm_alias)


def forward_kwargs(fx, device):
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    coord2d = S.coord2d_roi(fx["roi_center"], fx["scale"])
    return dict(roi_classes=T(fx["roi_cls"]), roi_cams=T(fx["roi_cam"]), roi_whs=T(fx["roi_wh"]),
                roi_centers=T(fx["roi_center"]), resize_ratios=T(fx["resize_ratio"]), roi_coord_2d=T(coord2d),
                roi_extents=T(fx["roi_extent"]))


# ---- the reference's rot6d -> R_allo -> R_ego in float64, vectorised (rot_reps.py:34-55, core/utils/utils.py:31-62) --------------
def ego_rot_from_rot6d_f64(d6, trans):
    """R_ego f64[b,3,3] of allocentric 6-D rotations ``d6`` f64[b,6] and translations ``trans`` f64[b,3]: Gram-Schmidt with the
    columns (x, y, z), then the rotation about ``z_cam x t`` by ``acos(t_z / |t|)`` applied from the left."""
    d6, trans = np.asarray(d6, np.float64), np.asarray(trans, np.float64)
    x = d6[:, 0:3] / np.linalg.norm(d6[:, 0:3], axis=1, keepdims=True)
    z = np.cross(x, d6[:, 3:6])
    z = z / np.linalg.norm(z, axis=1, keepdims=True)
    y = np.cross(z, x)
    R_allo = np.stack((x, y, z), axis=-1)
    ray = trans / np.linalg.norm(trans, axis=1, keepdims=True)
    angle = np.arccos(np.clip(ray[:, 2], -1.0, 1.0))
    axis = np.cross(np.array([0.0, 0.0, 1.0])[None], ray)
    n = np.linalg.norm(axis, axis=1, keepdims=True)
    axis = axis / np.where(n > 0, n, 1.0)
    c, s = np.cos(angle)[:, None, None], np.sin(angle)[:, None, None]
    ax, ay, az = axis[:, 0], axis[:, 1], axis[:, 2]
    zero = np.zeros_like(ax)
    Kx = np.stack([np.stack([zero, -az, ay], -1), np.stack([az, zero, -ax], -1), np.stack([-ay, ax, zero], -1)], 1)
    outer = axis[:, :, None] * axis[:, None, :]
    Rd = c * np.eye(3)[None] + s * Kx + (1 - c) * outer
    Rd = np.where((angle > 0)[:, None, None], Rd, np.eye(3)[None])
    return Rd @ R_allo


def pose_from_net_outputs_f64(pred_rot_, pred_t_, fx):
    """The reference's pose function downstream of the network, in float64: ``pose_from_predictions_test`` for the convnext_a6
    configs (allo_rot6d, centroid_z with Z_TYPE REL; pose_from_pred_centroid_z.py:56-154): centroid = (dx, dy) * (bw, bh) + centre,
    z = z_rel * resize_ratio, t = (z (cx - px) / fx, z (cy - py) / fy, z), R_ego = allo->ego(R(6-D), t).  ``fx``: the fixture's
    per-ROI arrays.  -> (R_ego f64[b,3,3], t f64[b,3])."""
    pt = np.asarray(pred_t_, np.float64)
    K, ctr, wh = fx["roi_cam"].astype(np.float64), fx["roi_center"].astype(np.float64), fx["roi_wh"].astype(np.float64)
    rr = fx["resize_ratio"].astype(np.float64).reshape(-1)
    cx, cy = pt[:, 0] * wh[:, 0] + ctr[:, 0], pt[:, 1] * wh[:, 1] + ctr[:, 1]
    z = pt[:, 2] * rr
    t = np.stack([z * (cx - K[:, 0, 2]) / K[:, 0, 0], z * (cy - K[:, 1, 2]) / K[:, 1, 1], z], 1)
    return ego_rot_from_rot6d_f64(pred_rot_, t), t


def ego_rot_sensitivity(pred_rot_, pred_t_, fx, h=1e-6):
    """|dR_ego / dx_j| (max over the nine entries of R) for the nine NETWORK OUTPUTS x = (6-D rotation, centroid dx, dy, z_rel), by
    central differences of ``pose_from_net_outputs_f64`` in float64: f64[b,9].  What the reference's own pose function does to an
    error in the network's outputs — large where the 6-D vectors are short / nearly parallel (Gram-Schmidt) or the predicted
    centroid lies near the principal point in front of or behind the camera (acos and the axis normalisation of
    allocentric_to_egocentric)."""
    x = np.concatenate([np.asarray(pred_rot_, np.float64), np.asarray(pred_t_, np.float64)], 1)
    out = np.zeros((x.shape[0], 9))
    scale = np.maximum(np.abs(x), 1e-3)
    for j in range(9):
        d = np.zeros_like(x)
        d[:, j] = h * scale[:, j]
        Rp, _ = pose_from_net_outputs_f64((x + d)[:, :6], (x + d)[:, 6:], fx)
        Rm, _ = pose_from_net_outputs_f64((x - d)[:, :6], (x - d)[:, 6:], fx)
        out[:, j] = np.abs((Rp - Rm) / (2 * d[:, j])[:, None, None]).reshape(x.shape[0], -1).max(1)
    return out
