"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py
It compiles the reference's own sources where they lie (oracle.build_ref) and records their
outputs on seeded inputs:
  fps_golden.npz   core/csrc/fps/src/farthest_point_sampling.cpp (both entry points)
  nnd_golden.npz   core/csrc/torch_nndistance/src/nnd_cpu.cpp (forward + backward)
  upnp_golden.npz  uncertainty_pnp.cpp:16-34 cost evaluated through the vendored ceres/jet.h +
                   ceres/rotation.h, and the optimum found by the vendored ceres::TinySolver
  flow_golden.npz  core/csrc/flow/src/flow_cpu.cpp flow_kernel<float> (one image per call: the file's pointer bump
                   makes batches > 1 meaningless)
  ransac_golden.npz  the four kernels of core/csrc/ransac_voting/src/ransac_voting_kernel.cu, extracted verbatim and
                   compiled for the host behind a grid emulator (oracle/ref_shims/ransac_ref_shim.cpp)
The fixtures travel to the GPU box; /root/reference does not.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

f32p = ctypes.POINTER(ctypes.c_float)
f64p = ctypes.POINTER(ctypes.c_double)
i32p = ctypes.POINTER(ctypes.c_int)


def P(a, t):
    return a.ctypes.data_as(t)


def make_flow_case(rng, h, w, motion):
    """A smooth surface seen from two nearby poses: depth_src analytic, depth_tgt = z-buffer splat of the warped source
    points (+ 1 mm noise, 5 % holes), so a large share of the pixels passes the reference's 3 mm consistency test."""
    K = np.array([[572.4114 * w / 640, 0, 325.2611 * w / 640], [0, 573.57043 * h / 480, 242.04899 * h / 480], [0, 0, 1]])
    yy, xx = np.mgrid[0:h, 0:w]
    ds = (0.8 + 0.1 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.float32)
    ds[rng.random((h, w)) < 0.05] = 0.0
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    ang = motion * 0.5
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    t = rng.standard_normal(3) * motion * 0.2
    KT = (K @ np.concatenate([R, t[:, None]], 1)).astype(np.float32)[None]
    Kinv = np.linalg.inv(K).astype(np.float32)[None]
    pts = (np.linalg.inv(K) @ np.stack([xx.ravel(), yy.ravel(), np.ones(h * w)])) * ds.ravel()
    pr = K @ (R @ pts + t[:, None])
    dt = np.zeros((h, w), np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = np.rint(pr[0] / pr[2]).astype(int), np.rint(pr[1] / pr[2]).astype(int)
    ok = (ds.ravel() > 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
    for i in np.argsort(-pr[2]):   # far first, near overwrites
        if ok[i]:
            dt[v[i], u[i]] = pr[2, i]
    dt += (rng.standard_normal((h, w)) * 1e-3).astype(np.float32) * (dt > 0)
    return np.ascontiguousarray(ds[None, None]), np.ascontiguousarray(dt[None, None]), np.ascontiguousarray(KT), np.ascontiguousarray(Kinv)


def main():
    libs = oracle.build_ref(force=True)
    assert set(libs) == {"fps", "nnd", "upnp", "flow", "ransac"}, libs
    rng = np.random.default_rng(20220925)

    # ---- FPS -------------------------------------------------------------------------
    ref = ctypes.CDLL(libs["fps"])
    out = {}
    for k, (pn, sn) in enumerate([(1000, 8), (2562, 64), (13000, 128), (40, 8), (6, 10)]):
        pts = rng.standard_normal((pn, 3)).astype(np.float32) * rng.uniform(0.02, 0.2, 3).astype(np.float32)
        if k == 3:  # duplicated points + exact ties
            pts[20:] = pts[:20]
        ic = np.zeros(sn, np.int32)
        ref.farthest_point_sampling_init_center(P(pts, f32p), P(ic, i32p), pn, sn)
        rd = np.zeros(sn, np.int32)
        ref.farthest_point_sampling(P(pts, f32p), P(rd, i32p), pn, sn)  # rd[0] = its time-seeded start
        out[f"pts{k}"] = pts
        out[f"init_center{k}"] = ic
        out[f"random{k}"] = rd
    np.savez_compressed(os.path.join(HERE, "fps_golden.npz"), **out)

    # ---- NN distance -----------------------------------------------------------------
    ref = ctypes.CDLL(libs["nnd"])
    out = {}
    for k, (b, n, m) in enumerate([(2, 300, 450), (1, 65, 1030), (3, 7, 5)]):
        x1 = rng.standard_normal((b, n, 3)).astype(np.float32)
        x2 = rng.standard_normal((b, m, 3)).astype(np.float32)
        if k == 2:
            x2[:, 3] = x2[:, 1]  # duplicate target -> first minimum must win
            x1[:, 0] = x2[:, 1]
        d1, d2 = np.zeros((b, n), np.float32), np.zeros((b, m), np.float32)
        i1, i2 = np.zeros((b, n), np.int32), np.zeros((b, m), np.int32)
        ref.ref_nnd_forward(P(x1, f32p), P(x2, f32p), P(d1, f32p), P(d2, f32p), P(i1, i32p), P(i2, i32p), b, n, m)
        gd1 = rng.standard_normal((b, n)).astype(np.float32)
        gd2 = rng.standard_normal((b, m)).astype(np.float32)
        g1, g2 = np.zeros_like(x1), np.zeros_like(x2)
        ref.ref_nnd_backward(P(x1, f32p), P(x2, f32p), P(g1, f32p), P(g2, f32p), P(gd1, f32p), P(gd2, f32p),
                             P(i1, i32p), P(i2, i32p), b, n, m)
        out.update({f"x1_{k}": x1, f"x2_{k}": x2, f"d1_{k}": d1, f"d2_{k}": d2, f"i1_{k}": i1, f"i2_{k}": i2,
                    f"gd1_{k}": gd1, f"gd2_{k}": gd2, f"g1_{k}": g1, f"g2_{k}": g2})
    np.savez_compressed(os.path.join(HERE, "nnd_golden.npz"), **out)

    # ---- uncertainty-PnP -------------------------------------------------------------
    ref = ctypes.CDLL(libs["upnp"])
    K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1.0])
    poses, p2s, p3s, ws, rs, Js = [], [], [], [], [], []
    for trial in range(32):
        pose = rng.uniform(-1, 1, 6)
        pose[5] += 3
        if trial == 0:
            pose[:3] = 0.0          # small-angle branch of AngleAxisRotatePoint
        if trial == 1:
            pose[:3] = 1e-9
        p2, p3, w = rng.uniform(0, 480, 2), rng.uniform(-0.2, 0.2, 3), rng.uniform(-1, 2, 3)
        r, J = np.zeros(2), np.zeros((2, 6))
        ref.ref_upnp_residual(P(pose, f64p), P(p2, f64p), P(p3, f64p), P(w, f64p), P(K, f64p), P(r, f64p), P(J, f64p))
        for lst, v in zip((poses, p2s, p3s, ws, rs, Js), (pose, p2, p3, w, r, J)):
            lst.append(v)
    out = dict(K=K, pose=np.stack(poses), p2=np.stack(p2s), p3=np.stack(p3s), w=np.stack(ws), r=np.stack(rs),
               J=np.stack(Js))

    def rodrigues(w):
        th = np.linalg.norm(w)
        if th < 1e-12:
            return np.eye(3)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx

    for k, pn in enumerate([9, 64, 1024]):
        rt = np.concatenate([rng.uniform(-1, 1, 3), [0.05, -0.03, 1.0]])
        p3 = rng.uniform(-0.1, 0.1, (pn, 3))
        Xc = (rodrigues(rt[:3]) @ p3.T).T + rt[3:]
        p2 = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[4] * Xc[:, 1] / Xc[:, 2] + K[5]], 1)
        p2 = p2 + rng.normal(0, 1, (pn, 2))
        w = np.stack([rng.uniform(0.5, 2, pn), rng.uniform(-0.2, 0.2, pn), rng.uniform(0.5, 2, pn)], 1)
        init = rt + rng.uniform(0, 0.1, 6)  # perturbation of uncertainty_pnp.cpp:120-128
        res = np.zeros(6)
        ref.ref_upnp_tinysolver(P(p2, f64p), P(p3, f64p), P(w, f64p), P(K, f64p), P(init, f64p), P(res, f64p), pn, 200)
        out.update({f"lm_p2_{k}": p2, f"lm_p3_{k}": p3, f"lm_w_{k}": w, f"lm_init_{k}": init, f"lm_opt_{k}": res,
                    f"lm_gt_{k}": rt})
    np.savez_compressed(os.path.join(HERE, "upnp_golden.npz"), **out)

    # ---- depth-to-flow ------------------------------------------------------------------
    ref = ctypes.CDLL(libs["flow"])
    out = {}
    for k, (h, w, motion) in enumerate([(48, 64, 0.01), (60, 80, 0.05), (33, 47, 0.3)]):
        ds, dt, KT, Kinv = make_flow_case(rng, h, w, motion)
        flow = np.full((1, 2, h, w), np.nan, np.float32)
        valid = np.full((1, 1, h, w), np.nan, np.float32)
        ref.ref_flow_forward(P(ds, f32p), P(dt, f32p), P(KT, f32p), P(Kinv, f32p), P(flow, f32p), P(valid, f32p), 1, h, w)
        assert np.isfinite(flow).all() and 0.05 < valid.mean() < 0.999, valid.mean()
        out.update({f"ds{k}": ds, f"dt{k}": dt, f"KT{k}": KT, f"Kinv{k}": Kinv, f"flow{k}": flow, f"valid{k}": valid})
    np.savez_compressed(os.path.join(HERE, "flow_golden.npz"), **out)

    # ---- RANSAC voting kernels -------------------------------------------------------------
    ref = ctypes.CDLL(libs["ransac"])
    u8p = ctypes.POINTER(ctypes.c_ubyte)
    out = {}
    for k, (tn, vn, hn) in enumerate([(500, 9, 128), (37, 3, 16), (1500, 9, 96)]):
        coords = np.stack([rng.uniform(0, 64, tn), rng.uniform(0, 64, tn)], 1).astype(np.float32)
        kpts = rng.uniform(-20, 84, (vn, 2))
        d = kpts[None] - coords[:, None].astype(np.float64)
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        direct = (d + rng.normal(0, 0.05, d.shape)).astype(np.float32)
        direct[rng.integers(0, tn, 3)] = 0.0                                    # zero-norm directions
        idxs = rng.integers(0, tn, (hn, vn, 2)).astype(np.int32)
        idxs[0, :, 1] = idxs[0, :, 0]                                           # degenerate pairs: kernel returns early
        out.update({f"direct{k}": direct, f"coords{k}": coords, f"idxs{k}": idxs})
        for vp in (0, 1):
            hypo = np.zeros((hn, vn, 3 if vp else 2), np.float32)
            ref.ref_generate_hypothesis(P(direct, f32p), P(coords, f32p), P(idxs, i32p), P(hypo, f32p), tn, vn, hn, vp)
            inl = np.zeros((hn, vn, tn), np.uint8)
            ref.ref_voting_for_hypothesis(P(direct, f32p), P(coords, f32p), P(hypo, f32p), inl.ctypes.data_as(u8p), tn, vn, hn,
                                          ctypes.c_float(0.99), vp)
            assert np.isfinite(hypo).all() and 0 < inl.mean() < 0.9, (k, vp, inl.mean())
            out.update({f"hypo{k}_{vp}": hypo, f"inl{k}_{vp}": np.packbits(inl, axis=-1)})
    np.savez_compressed(os.path.join(HERE, "ransac_golden.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
