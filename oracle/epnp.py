"""ORACLE — test infrastructure.  ``cv2.solvePnPRansac(..., flags=SOLVEPNP_EPNP)`` restated in NumPy.

Callers in the reference: lib/pysixd/misc.py:153-208 (``pnp_v2``: reprojErr 3, 100 iterations),
core/gdrn_modeling/engine/gdrn_evaluator.py:373-459 (``process_pnp_ransac``) and :313-330 (``net_ransac_pnp``: 20 iterations),
core/csrc/uncertainty_pnp/un_pnp_utils.py:27-44 (EPnP on the four best-weighted points as the LM initialiser).

The arithmetic lives in OpenCV (calib3d), a third-party dependency that is NOT under /root/reference and not installed here
(unpinned version; pulled in by mmcv).  PARITY UNPINNED: this file restates the published algorithms —

* EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) the way calib3d/src/epnp.cpp runs it: control points from the PCA of the
  model points, barycentric coordinates, the 2n x 12 system M, the four smallest eigenvectors of MtM, the three beta
  approximations (N = 1..3 unknown betas from the 6 x 10 distance system) each polished by five Gauss-Newton steps, camera
  points -> sign fix -> Horn/Arun absolute orientation, the candidate with the smallest mean reprojection error wins;
* the RANSAC driver of calib3d/src/ptsetreg.cpp (``RANSACPointSetRegistrator::run``) with ``PnPRansacCallback``
  (solvepnp.cpp): 5-point minimal sets drawn with cv::RNG(2^64 - 1) (``getSubset``: redraw on duplicates), squared
  reprojection error in float32 against reprojErr^2, best = strictly more inliers, adaptive iteration count
  (``RANSACUpdateNumIters``, confidence 0.99), final EPnP on all inliers of the best model —

* exactly four correspondences: solvePnPRansac switches its kernel to SOLVEPNP_P3P and solves once with all four
  (solvepnp.cpp: ``npoints == 4 -> model_points = 4, ransac_kernel_method = SOLVEPNP_P3P; model_points == npoints -> solvePnP``).
  calib3d's p3p.cpp (Gao, Hou, Tang, Chang, PAMI 2003) finds the up-to-four poses consistent with the FIRST THREE points and
  orders them by the reprojection error of the fourth; the first is returned.  ``p3p_4points`` restates that contract with
  Grunert's elimination (Haralick et al., "Review and analysis of solutions of the three point perspective pose estimation
  problem", IJCV 1994): the P3P solution set does not depend on the elimination used, so both pick the same pose —

and is anchored on closed-form properties instead of golden vectors: exact correspondences recover the pose they were
projected with; with outliers the inlier set is the set of uncontaminated points.  The dense linear algebra uses LAPACK
(np.linalg), an implementation independent of the device code.
"""
from __future__ import annotations

import numpy as np

PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


# ---- cv::RNG (core/include/opencv2/core/operations.hpp): multiply-with-carry -------------------------------------------------
class CvRNG:
    COEFF = 4164903690

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * self.COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def get_subset(rng_next, count, model_points=5, max_attempts=1000):
    """ptsetreg.cpp getSubset: model_points distinct indices, each ``next() % count``, redrawn while it repeats."""
    idx = []
    iters = 0
    i = 0
    while i < model_points and iters < max_attempts:
        cand = int(rng_next() % count)
        iters += 1            # OpenCV counts an attempt per full subset; a per-draw cap is unreachable for count >= 5 anyway
        if cand in idx:
            continue
        idx.append(cand)
        i += 1
    return idx if len(idx) == model_points else None


# ---- EPnP ------------------------------------------------------------------------------------------------------------
def _control_points(pw):
    c0 = pw.mean(0)
    d = pw - c0
    dc, uc = np.linalg.eigh(d.T @ d)               # ascending; epnp.cpp takes the SVD (descending) — order is immaterial
    cws = [c0]
    for i in (2, 1, 0):
        axis = uc[:, i]
        # the sign of a principal axis is the eigen-solver's choice and moves the control point to the mirrored position
        # (the pose then differs at the noise level): fixed here — the largest-magnitude component is positive
        if axis[int(np.argmax(np.abs(axis)))] < 0:
            axis = -axis
        cws.append(c0 + np.sqrt(max(dc[i], 0.0) / len(pw)) * axis)
    return np.array(cws)


def _alphas(pw, cws):
    cc = (cws[1:] - cws[0]).T                      # columns c_j - c_0
    a = np.linalg.solve(cc, (pw - cws[0]).T).T
    return np.concatenate([1.0 - a.sum(1, keepdims=True), a], 1)


def _L_rho(v, cws):
    L = np.zeros((6, 10))
    rho = np.zeros(6)
    for r, (a, b) in enumerate(PAIRS):
        dv = [v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for i in range(4)]
        L[r] = [dv[0] @ dv[0], 2 * dv[0] @ dv[1], dv[1] @ dv[1], 2 * dv[0] @ dv[2], 2 * dv[1] @ dv[2], dv[2] @ dv[2],
                2 * dv[0] @ dv[3], 2 * dv[1] @ dv[3], 2 * dv[2] @ dv[3], dv[3] @ dv[3]]
        rho[r] = ((cws[a] - cws[b]) ** 2).sum()
    return L, rho


def _gauss_newton(L, rho, betas, iters=5):
    b = np.array(betas, np.float64)
    for _ in range(iters):
        A = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                      L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                      L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                      L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], 1)
        res = rho - (L[:, 0] * b[0] ** 2 + L[:, 1] * b[0] * b[1] + L[:, 2] * b[1] ** 2 + L[:, 3] * b[0] * b[2]
                     + L[:, 4] * b[1] * b[2] + L[:, 5] * b[2] ** 2 + L[:, 6] * b[0] * b[3] + L[:, 7] * b[1] * b[3]
                     + L[:, 8] * b[2] * b[3] + L[:, 9] * b[3] ** 2)
        q, r = np.linalg.qr(A)                     # epnp.cpp qr_solve: Householder QR, no rank truncation
        if np.any(np.diag(r) == 0):
            return None
        b = b + np.linalg.solve(r, q.T @ res)
    return b


def _betas_approx(L, rho):
    out = []
    b4 = np.linalg.lstsq(L[:, [0, 1, 3, 6]], rho, rcond=None)[0]        # N = 1: B11 B12 B13 B14
    if b4[0] < 0:
        s = np.sqrt(-b4[0]); out.append([s, -b4[1] / s, -b4[2] / s, -b4[3] / s])
    else:
        s = np.sqrt(b4[0]); out.append([s, b4[1] / s, b4[2] / s, b4[3] / s])
    b3 = np.linalg.lstsq(L[:, [0, 1, 2]], rho, rcond=None)[0]           # N = 2: B11 B12 B22
    if b3[0] < 0:
        be = [np.sqrt(-b3[0]), np.sqrt(-b3[2]) if b3[2] < 0 else 0.0]
    else:
        be = [np.sqrt(b3[0]), np.sqrt(b3[2]) if b3[2] > 0 else 0.0]
    if b3[1] < 0:
        be[0] = -be[0]
    out.append([be[0], be[1], 0.0, 0.0])
    b5 = np.linalg.lstsq(L[:, [0, 1, 2, 3, 4]], rho, rcond=None)[0]     # N = 3: B11 B12 B22 B13 B23
    if b5[0] < 0:
        be = [np.sqrt(-b5[0]), np.sqrt(-b5[2]) if b5[2] < 0 else 0.0]
    else:
        be = [np.sqrt(b5[0]), np.sqrt(b5[2]) if b5[2] > 0 else 0.0]
    if b5[1] < 0:
        be[0] = -be[0]
    out.append([be[0], be[1], b5[3] / be[0], 0.0])
    return out


def _pose_from_betas(v, betas, alphas, pw, uv, fu, fv, uc, vc):
    ccs = sum(betas[i] * v[i].reshape(4, 3) for i in range(4))
    pcs = alphas @ ccs
    if pcs[0, 2] < 0:
        ccs, pcs = -ccs, -pcs
    pc0, pw0 = pcs.mean(0), pw.mean(0)
    abt = (pcs - pc0).T @ (pw - pw0)
    U, _, Vt = np.linalg.svd(abt)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R[2] = -R[2]
    t = pc0 - R @ pw0
    cam = pw @ R.T + t
    inv = 1.0 / cam[:, 2]
    err = np.sqrt((uc + fu * cam[:, 0] * inv - uv[:, 0]) ** 2 + (vc + fv * cam[:, 1] * inv - uv[:, 1]) ** 2).mean()
    return R, t, err


def canonical_null_basis(w, vec, rel_tol=1e-9):
    """Four or five points leave MtM an exactly degenerate null space (12 - 2n dimensions): ANY orthonormal basis of it is a
    valid set of "smallest eigenvectors", the three beta approximations start from whichever the eigen-solver returns and
    five Gauss-Newton steps do not erase the choice (OpenCV's result depends on its SVD there).  To make the restatement
    well defined, the degenerate block is replaced by the basis obtained by projecting e_0, e_1, ... onto the subspace and
    orthonormalising in that order (unique for a given subspace, signs included).  Non-degenerate eigenvectors are kept."""
    d = int((w[:4] <= rel_tol * w[-1]).sum())
    if d < 2:
        return vec
    d = int((w <= rel_tol * w[-1]).sum())
    Q = vec[:, :d]
    basis = []
    for k in range(12):
        c = Q @ Q[k]                               # projection of e_k onto the subspace
        for b in basis:
            c = c - (b @ c) * b
        nrm = np.linalg.norm(c)
        if nrm > 1e-6:
            basis.append(c / nrm)
            if len(basis) == d:
                break
    out = vec.copy()
    out[:, :d] = np.array(basis).T
    return out


def epnp(pw, uv, K):
    """pw f64[n,3], uv f64[n,2] (n >= 4), K 3x3 -> (R, t) of the best of the three beta approximations, or None when no
    candidate is finite (degenerate configuration)."""
    pw, uv = np.asarray(pw, np.float64), np.asarray(uv, np.float64)
    fu, fv, uc, vc = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    cws = _control_points(pw)
    al = _alphas(pw, cws)
    n = len(pw)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu
        M[0::2, 3 * j + 2] = al[:, j] * (uc - uv[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv
        M[1::2, 3 * j + 2] = al[:, j] * (vc - uv[:, 1])
    w, vec = np.linalg.eigh(M.T @ M)               # ascending: columns 0..3 are the null-space candidates
    vec = canonical_null_basis(w, vec)
    v = [vec[:, i] for i in range(4)]              # epnp.cpp: ut[11], ut[10], ut[9], ut[8]
    L, rho = _L_rho(v, cws)
    best = None
    for b0 in _betas_approx(L, rho):
        if not np.all(np.isfinite(b0)):
            continue
        betas = _gauss_newton(L, rho, b0)
        if betas is None:
            continue
        R, t, err = _pose_from_betas(v, betas, al, pw, uv, fu, fv, uc, vc)
        if np.isfinite(err) and (best is None or err < best[2]):
            best = (R, t, err)
    if best is None:
        return None
    return best[0], best[1]


# ---- RANSAC ----------------------------------------------------------------------------------------------------------
def reproj_err2_f32(pw32, uv32, K, R, t):
    """PnPRansacCallback::computeError: projectPoints in double from float32 points, projections stored as float32,
    squared distance in float32."""
    cam = pw32.astype(np.float64) @ np.asarray(R, np.float64).T + np.asarray(t, np.float64)
    u = (K[0][0] * cam[:, 0] / cam[:, 2] + K[0][2]).astype(np.float32)
    v = (K[1][1] * cam[:, 1] / cam[:, 2] + K[1][2]).astype(np.float32)
    dx, dy = uv32[:, 0] - u, uv32[:, 1] - v
    return dx * dx + dy * dy


def update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))


def p3p_4points(pw, uv, K):
    """cv2.solvePnP(flags=SOLVEPNP_P3P) on exactly four correspondences -> (R, t) or None.

    Unit bearings j_i of the first three image points, model distances a = |P2 P3|, b = |P1 P3|, c = |P1 P2|, cosines
    alpha = (j2, j3), beta = (j1, j3), gamma = (j1, j2).  With camera distances s2 = u s1, s3 = v s1 the three cosine laws give
        u = N(v) / D(v),  N = (q - 1) v^2 - 2 q cos(beta) v + 1 + q,  D = 2 (cos(gamma) - v cos(alpha)),  q = (a^2 - c^2) / b^2
    and, substituted into  u^2 - 2 cos(gamma) u + 1 = (c^2 / b^2)(1 + v^2 - 2 v cos(beta)),  the quartic
        N^2 - 2 cos(gamma) N D + D^2 (1 - (c^2 / b^2)(1 + v^2 - 2 v cos(beta))) = 0   in v.
    Every real root v > 0 with u > 0 (u re-derived from the two cosine laws it has to satisfy, see below) gives s1 = c / sqrt(1 + u^2 - 2 u cos(gamma)) and the camera points s_i j_i; the rigid motion
    model -> camera follows by absolute orientation (Arun).  The candidate with the smallest reprojection error of the FOURTH
    point (normalised image coordinates, like p3p.cpp) wins."""
    pw = np.asarray(pw, np.float64).reshape(4, 3)
    uv = np.asarray(uv, np.float64).reshape(4, 2)
    K = np.asarray(K, np.float64).reshape(3, 3)
    xn = np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0], (uv[:, 1] - K[1, 2]) / K[1, 1], np.ones(4)], 1)
    j = xn[:3] / np.linalg.norm(xn[:3], axis=1, keepdims=True)
    a2, b2, c2 = np.sum((pw[1] - pw[2]) ** 2), np.sum((pw[0] - pw[2]) ** 2), np.sum((pw[0] - pw[1]) ** 2)
    if min(a2, b2, c2) <= 0.0:
        return None
    ca, cb, cg = j[1] @ j[2], j[0] @ j[2], j[0] @ j[1]
    q = (a2 - c2) / b2
    N = np.array([q - 1.0, -2.0 * q * cb, 1.0 + q])                     # highest power first (np.poly convention)
    D = np.array([-2.0 * ca, 2.0 * cg])
    Q = np.array([-c2 / b2, 2.0 * (c2 / b2) * cb, 1.0 - c2 / b2])        # 1 - (c^2 / b^2)(1 + v^2 - 2 v cos beta)
    quartic = np.polyadd(np.polysub(np.polymul(N, N), 2.0 * cg * np.polymul(N, D)), np.polymul(np.polymul(D, D), Q))
    if not np.all(np.isfinite(quartic)) or abs(quartic[0]) < 1e-300:
        return None
    best = None
    for v in np.roots(quartic):
        if abs(v.imag) > 1e-7 * max(1.0, abs(v.real)) or v.real <= 0.0:
            continue
        v = v.real
        # u from the quadratic  u^2 - 2 cos(gamma) u + 1 = (c^2 / b^2) f,  f = 1 + v^2 - 2 v cos(beta)  (N / D is 0 / 0 when the
        # object subtends a small angle: cos(gamma) ~ v cos(alpha)); of its two roots the one that also satisfies
        # u^2 - 2 v cos(alpha) u + v^2 = (a^2 / b^2) f
        f = 1.0 + v * v - 2.0 * v * cb
        disc = max(cg * cg - 1.0 + (c2 / b2) * f, 0.0)
        cand = [cg + np.sqrt(disc), cg - np.sqrt(disc)]
        u = min(cand, key=lambda x: abs(x * x - 2.0 * v * ca * x + v * v - (a2 / b2) * f))
        w = 1.0 + u * u - 2.0 * u * cg
        if u <= 0.0 or w <= 0.0:
            continue
        s1 = np.sqrt(c2 / w)
        pc = np.stack([s1 * j[0], u * s1 * j[1], v * s1 * j[2]])
        # absolute orientation of the three pairs (Arun): pc = R pw + t
        mw, mc = pw[:3].mean(0), pc.mean(0)
        H = (pc - mc).T @ (pw[:3] - mw)
        U, _, Vt = np.linalg.svd(H)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            U[:, 2] = -U[:, 2]
            R = U @ Vt
        t = mc - R @ mw
        p4 = R @ pw[3] + t
        err = (p4[0] / p4[2] - xn[3, 0]) ** 2 + (p4[1] / p4[2] - xn[3, 1]) ** 2
        if np.isfinite(err) and (best is None or err < best[0]):
            best = (err, R, t)
    return None if best is None else (best[1], best[2])


def solve_pnp_ransac_epnp(pw, uv, K, reproj_err=3.0, iters=100, confidence=0.99, rng_next=None):
    """-> (ok, R, t, inlier_mask bool[n]).  ``rng_next`` yields 32-bit words (default: cv::RNG seeded like OpenCV)."""
    pw32, uv32 = np.asarray(pw, np.float32), np.asarray(uv, np.float32)      # solvePnPRansac converts to CV_32F
    K = np.asarray(K, np.float64)
    n = len(pw32)
    if n < 4:
        return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
    if n == 4:      # model_points = 4, kernel P3P, model_points == npoints: one solve with all four, every point an inlier
        sol = p3p_4points(pw32, uv32, K)
        if sol is None:
            return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
        return True, sol[0], sol[1], np.ones(n, bool)
    model_points = 5
    if n == model_points:
        sol = epnp(pw32, uv32, K)
        if sol is None:
            return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
        return True, sol[0], sol[1], np.ones(n, bool)
    if rng_next is None:
        rng_next = CvRNG().next
    thr2 = np.float32(reproj_err) * np.float32(reproj_err)
    best_mask, best_count, niters = None, 0, iters
    it = 0
    while it < niters:
        idx = get_subset(rng_next, n, model_points)
        if idx is None:
            if it == 0:
                return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
            break
        sol = epnp(pw32[idx], uv32[idx], K)           # runKernel: no model -> next iteration
        it += 1
        if sol is None:
            continue
        mask = reproj_err2_f32(pw32, uv32, K, sol[0], sol[1]) <= thr2
        good = int(mask.sum())
        if good > max(best_count, model_points - 1):
            best_mask, best_count = mask, good
            niters = update_num_iters(confidence, (n - good) / n, model_points, niters)
    if best_mask is None:
        return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
    sol = epnp(pw32[best_mask], uv32[best_mask], K)
    if sol is None:
        return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
    return True, sol[0], sol[1], best_mask
