"""core/csrc/uncertainty_pnp/un_pnp_utils.py:11-158 with the same call signatures.

The reference initialises with ``cv2.solvePnP(..., SOLVEPNP_EPNP)`` on the 4 highest-weight points
(:27-44) — OpenCV is a third-party dependency outside the tree and is not installed here.  When ``cv2`` is
importable it is used exactly like the reference; otherwise the same four points go through the device EPnP
(``gdrnpp_epnp_batched``, OpenCV's published epnp.cpp restated; float32 points like the device path everywhere).
``init_rt`` (6-vector, angle-axis + t) overrides both."""
import numpy as np

from ._ext import ffi, lib

try:  # pragma: no cover - cv2 is absent in the build container
    import cv2
except Exception:  # noqa: BLE001
    cv2 = None


def _rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0] (angle-axis -> matrix), float64."""
    r = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def _rodrigues_inv(R):
    """cv2.Rodrigues(R)[0] (matrix -> angle-axis), float64."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-12:
        return 0.5 * v
    if np.pi - th < 1e-6:      # near pi: axis from the diagonal of (R + I) / 2
        a = np.sqrt(np.maximum((np.diag(R) + 1.0) * 0.5, 0.0))
        k = int(np.argmax(a))
        a = np.where(np.arange(3) == k, a, np.copysign(a, (R[k] + R[:, k])))
        return th * a / np.linalg.norm(a)
    return th * v / (2.0 * np.sin(th))


def _init_pose(points_3d, points_2d, camera_matrix, idxs, init_rt):
    if init_rt is not None:
        return np.ascontiguousarray(np.asarray(init_rt, np.float64).reshape(6, 1))
    if cv2 is None:
        import torch

        from .... import hip_lib

        T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()  # noqa: E731
        R, t, status = hip_lib.epnp_batched(T(points_2d[idxs][None]), T(points_3d[idxs][None]), T(camera_matrix.reshape(1, 9)))
        if int(status.item()) != 1:
            raise RuntimeError("uncertainty_pnp: EPnP on the four best-weighted points is degenerate; pass init_rt")
        return np.ascontiguousarray(np.concatenate([_rodrigues_inv(R[0].cpu().numpy().astype(np.float64)),
                                                    t[0].cpu().numpy().astype(np.float64)]).reshape(6, 1))
    dist_coeffs = np.zeros(shape=[8, 1], dtype=np.float64)
    _, R_exp, t = cv2.solvePnP(np.expand_dims(points_3d[idxs, :], 0), np.expand_dims(points_2d[idxs, :], 0),
                               camera_matrix, dist_coeffs, None, None, False, flags=cv2.SOLVEPNP_EPNP)
    return np.ascontiguousarray(np.concatenate([R_exp, t], 0), np.float64)


def _solve(points_2d, points_3d, weights_2d, camera_matrix, init_rt):
    pn = points_2d.shape[0]
    points_2d = np.ascontiguousarray(points_2d, np.float64)
    points_3d = np.ascontiguousarray(points_3d, np.float64)
    weights_2d = np.ascontiguousarray(weights_2d, np.float64)
    camera_matrix = np.ascontiguousarray(camera_matrix, np.float64)
    result_rt = np.empty([6], np.float64)
    lib.uncertainty_pnp(ffi.cast("double*", points_2d.ctypes.data), ffi.cast("double*", points_3d.ctypes.data),
                        ffi.cast("double*", weights_2d.ctypes.data), ffi.cast("double*", camera_matrix.ctypes.data),
                        ffi.cast("double*", init_rt.ctypes.data), ffi.cast("double*", result_rt.ctypes.data), pn)
    if not np.isfinite(result_rt).all():
        raise RuntimeError("uncertainty_pnp failed on the device (see stderr)")
    R = _rodrigues(result_rt[:3])
    return np.concatenate([R, result_rt[3:, None]], axis=-1)


def uncertainty_pnp(points_2d, weights_2d, points_3d, camera_matrix, init_rt=None):
    """points_2d [pn,2], weights_2d [pn,3]=(wxx,wxy,wyy), points_3d [pn,3], camera_matrix [3,3] -> Rt [3,4]."""
    pn = points_2d.shape[0]
    assert points_3d.shape[0] == pn and pn >= 4
    points_3d = points_3d.astype(np.float64)
    points_2d = points_2d.astype(np.float64)
    weights_2d = weights_2d.astype(np.float64)
    camera_matrix = camera_matrix.astype(np.float64)
    idxs = np.argsort(weights_2d[:, 0] + weights_2d[:, 1])[-4:]
    init = _init_pose(points_3d, points_2d, camera_matrix, idxs, init_rt)
    if pn == 4:
        return np.concatenate([_rodrigues(init[:3]), init[3:].reshape(3, 1)], axis=-1)
    return _solve(points_2d, points_3d, weights_2d, camera_matrix, init)


def uncertainty_pnp_v2(points_2d, covars, points_3d, camera_matrix, type="single", init_rt=None):
    """un_pnp_utils.py:81-158: isotropic weights 1/max-eigenvalue of each 2x2 covariance."""
    pn = points_2d.shape[0]
    assert points_3d.shape[0] == pn and pn >= 4 and covars.shape[0] == pn
    points_3d = points_3d.astype(np.float64)
    points_2d = points_2d.astype(np.float64)
    camera_matrix = camera_matrix.astype(np.float64)
    weights = []
    for pi in range(pn):
        if covars[pi, 0, 0] < 1e-5:
            weights.append(0.0)
        else:
            weights.append(1.0 / np.max(np.linalg.eigvals(covars[pi])))
    weights = np.asarray(weights, np.float64)
    idxs = np.argsort(weights)[-4:]
    init = _init_pose(points_3d, points_2d, camera_matrix, idxs, init_rt)
    if pn == 4:
        return np.concatenate([_rodrigues(init[:3]), init[3:].reshape(3, 1)], axis=-1)
    w3 = np.concatenate([weights[:, None], np.zeros([pn, 1]), weights[:, None]], 1)
    return _solve(points_2d, points_3d, w3, camera_matrix, init)
