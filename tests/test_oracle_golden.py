"""The CPU oracle against fixtures recorded from the reference's own compiled sources
(tests/golden/make_golden.py) — and, where oracle/_ref exists, against those libraries live."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from oracle import postproc as P

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int)


def test_fps_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "fps_golden.npz"))
    k = 0
    while f"pts{k}" in g:
        pts, ic, rd = g[f"pts{k}"], g[f"init_center{k}"], g[f"random{k}"]
        assert np.array_equal(P.fps(pts, len(ic), init_center=True), ic)
        # the reference's time-seeded start index is its first output; replaying it pins the rest
        assert np.array_equal(P.fps(pts, len(rd), init_center=False, start=int(rd[0])), rd)
        k += 1
    assert k == 5


def test_fps_oracle_matches_reference_live():
    ref = oracle.ref_lib("fps")
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(1)
    for pn, sn in [(3000, 32), (17, 17), (257, 300)]:
        pts = rng.standard_normal((pn, 3)).astype(np.float32)
        idx = np.zeros(sn, np.int32)
        ref.farthest_point_sampling_init_center(pts.ctypes.data_as(f32p), idx.ctypes.data_as(i32p), pn, sn)
        assert np.array_equal(P.fps(pts, sn, True), idx)


def test_nnd_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "nnd_golden.npz"))
    for k in range(3):
        d1, d2, i1, i2 = P.nnd_forward(g[f"x1_{k}"], g[f"x2_{k}"])
        assert np.array_equal(i1, g[f"i1_{k}"]) and np.array_equal(i2, g[f"i2_{k}"])
        assert np.array_equal(d1, g[f"d1_{k}"]) and np.array_equal(d2, g[f"d2_{k}"])
        g1, g2 = P.nnd_backward(g[f"x1_{k}"], g[f"x2_{k}"], g[f"gd1_{k}"], g[f"gd2_{k}"], i1, i2)
        assert np.array_equal(g1, g[f"g1_{k}"]) and np.array_equal(g2, g[f"g2_{k}"])


def test_upnp_cost_and_jacobian_match_ceres_jets(golden_dir):
    g = np.load(os.path.join(golden_dir, "upnp_golden.npz"))
    for i in range(len(g["pose"])):
        r, J = P.upnp_residual(g["pose"][i], g["p2"][i], g["p3"][i], g["w"][i], g["K"])
        np.testing.assert_allclose(r, g["r"][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(J, g["J"][i], rtol=1e-11, atol=1e-11)


def test_upnp_minimiser_reaches_tinysolver_optimum(golden_dir):
    g = np.load(os.path.join(golden_dir, "upnp_golden.npz"))
    for k in range(3):
        out, info = P.uncertainty_pnp(g[f"lm_p2_{k}"], g[f"lm_p3_{k}"], g[f"lm_w_{k}"], g["K"], g[f"lm_init_{k}"],
                                      return_info=True)
        assert info[1] in (0, 1, 2), info  # converged
        # Ceres' function tolerance (1e-6 relative cost) stops slightly before the tight optimum
        np.testing.assert_allclose(out, g[f"lm_opt_{k}"], atol=1e-4)


def test_upnp_exact_data_recovers_pose():
    rng = np.random.default_rng(3)
    K = np.array([400.0, 0, 128, 0, 400, 128, 0, 0, 1])
    rt = np.array([0.3, -0.2, 0.5, 0.1, -0.05, 1.2])
    p3 = rng.uniform(0, 1, (8, 3)) - 0.5
    th = np.linalg.norm(rt[:3])
    k = rt[:3] / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    X = p3 @ R.T + rt[3:]
    p2 = np.stack([400 * X[:, 0] / X[:, 2] + 128, 400 * X[:, 1] / X[:, 2] + 128], 1)
    w = np.tile([1.0, 0.0, 1.0], (8, 1))
    out = P.uncertainty_pnp(p2, p3, w, K, rt + rng.uniform(0, 0.1, 6))  # the reference's own demo (cpp:98-156)
    np.testing.assert_allclose(out, rt, atol=1e-6)


def test_flow_oracle_matches_reference_golden(golden_dir):
    """oracle/flow_oracle.c == the reference's flow_cpu.cpp (compiled unmodified, one image per call), bit for bit."""
    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    k = 0
    while f"ds{k}" in g:
        flow, valid = P.flow_forward(g[f"ds{k}"], g[f"dt{k}"], g[f"KT{k}"], g[f"Kinv{k}"])
        assert np.array_equal(flow, g[f"flow{k}"]) and np.array_equal(valid, g[f"valid{k}"])
        assert 0.05 < valid.mean() < 0.999
        k += 1
    assert k == 3


def test_flow_oracle_batches_are_per_image(golden_dir):
    """Batch semantics of the CUDA kernel (Kinv/KT indexed per image): a stacked batch equals the images one by one."""
    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    ds = np.concatenate([g["ds0"], g["ds0"][:, :, ::-1].copy()])
    dt = np.concatenate([g["dt0"], g["dt0"][:, :, ::-1].copy()])
    KT = np.concatenate([g["KT0"], g["KT0"] * np.float32(1.0)])
    KT[1, :, 3] += np.float32(0.01)
    Kinv = np.concatenate([g["Kinv0"], g["Kinv0"]])
    fb, vb = P.flow_forward(ds, dt, KT, Kinv)
    for i in range(2):
        f1, v1 = P.flow_forward(ds[i:i + 1], dt[i:i + 1], KT[i:i + 1], Kinv[i:i + 1])
        assert np.array_equal(fb[i:i + 1], f1) and np.array_equal(vb[i:i + 1], v1)


def test_ransac_voting_oracle_matches_reference_kernels(golden_dir):
    """oracle/ransac_voting_oracle.c == the reference's own CUDA kernels (ransac_voting_kernel.cu, bodies extracted
    verbatim and compiled for the host): hypotheses bit for bit (degenerate pairs untouched = 0), inlier flags identical."""
    g = np.load(os.path.join(golden_dir, "ransac_golden.npz"))
    k = 0
    while f"direct{k}" in g:
        direct, coords, idxs = g[f"direct{k}"], g[f"coords{k}"], g[f"idxs{k}"]
        tn = direct.shape[0]
        for vp in (0, 1):
            hypo = P.generate_hypothesis(direct, coords, idxs, vanishing_point=bool(vp))
            assert np.array_equal(hypo, g[f"hypo{k}_{vp}"])
            inl = P.voting_for_hypothesis(direct, coords, hypo, 0.99, vanishing_point=bool(vp))
            assert np.array_equal(inl, np.unpackbits(g[f"inl{k}_{vp}"], axis=-1)[..., :tn])
        k += 1
    assert k == 3


def test_postproc_oracle_matches_reference_python_functions(golden_dir):
    """oracle/postproc.py against the reference's OWN Python functions executed from their source text
    (tests/golden/make_golden_pyref.py): get_out_mask / get_out_coor, the 2D-3D correspondence selection (indices, order
    and values bit for bit), get_K_crop_resize, rot6d_to_mat_batch, pose_from_predictions_test."""
    g = np.load(os.path.join(golden_dir, "pyref_golden.npz"))
    mask_prob = P.get_out_mask(g["raw_mask"])
    assert np.array_equal(mask_prob, g["mask_prob"])
    xyz = P.get_out_coor(*g["coor"])
    assert np.array_equal(xyz, g["xyz"])
    off = 0
    for i, n in enumerate(g["corr_counts"]):
        ip, mp, _ = P.get_img_model_points_with_coords2d(mask_prob[i, 0], xyz[i].transpose(1, 2, 0), g["coord2d"][i], 480, 640,
                                                         g["extents"][i], mask_thr=0.5)
        assert len(ip) == n and np.array_equal(ip, g["corr_img"][off:off + n]) and np.array_equal(mp, g["corr_mdl"][off:off + n])
        off += n
    crop_xy = g["centers"] - g["scales"][:, None] / 2
    assert np.array_equal(P.get_K_crop_resize(g["K"], crop_xy, (64 / g["scales"])[:, None].astype(np.float32)), g["K_crop"])
    assert np.array_equal(P.zoom_K(g["K"], g["centers"], g["scales"], 64), g["K_crop"])
    np.testing.assert_allclose(P.rot6d_to_mat_batch(g["d6"]), g["R_allo"], rtol=0, atol=2e-7)   # torch vs NumPy reductions
    R_ego, trans = P.pose_from_predictions_test(g["R_allo"], g["pred_centroids"], g["pred_z"], g["K"], g["centers"],
                                                g["resize_ratio"], g["roi_whs"])
    assert np.array_equal(trans, g["trans"])
    np.testing.assert_allclose(R_ego, g["R_ego"], rtol=0, atol=1e-6)   # transforms3d stood in by scipy when recorded


def test_depth_refine_oracle_matches_reference_process_depth_refine(golden_dir):
    """oracle.postproc.depth_refine_roi against the reference's OWN process_depth_refine (gdrn_evaluator.py:461-573 executed
    from source with a stand-in `self`; renderer and cv2.resize served by the oracle's rasteriser / restated INTER_LINEAR):
    the refined translation agrees to 1e-9 m.  (Not bit for bit: torch's CPU `norm` accumulates x*x, fma(y,y,.), fma(z,z,.)
    in float32 on this host, the oracle and the HIP kernel use the uncontracted chain — a 1-ulp difference in ~2 % of the
    q-map weights, 1e-10 relative in t; SURVEY §8a lists this reduction among the order-dependent, tolerance-checked ones.)"""
    g = np.load(os.path.join(golden_dir, "pyref_golden.npz"))
    mask = P.get_out_mask(g["rf_mask"])
    n = len(g["rf_roi_cls"])
    for i in range(n):
        o = int(g["rf_roi_cls"][i])
        xyz = np.concatenate([g["rf_coor_x"][i], g["rf_coor_y"][i], g["rf_coor_z"][i]], 0).transpose(1, 2, 0)
        t = P.depth_refine_roi(xyz, mask[i, 0], g["rf_roi_depth"][i, 0], g["rf_K_crop"][i], g["rf_R"][i], g["rf_t_init"][i],
                               g["rf_verts"][o], g["rf_faces"][o], iters=2, threshold=0.8)
        np.testing.assert_allclose(t, g["rf_t_refined"][i], rtol=0, atol=1e-9)
        # and the refinement does what it is for: closer to the ground-truth depth than the initial estimate
        assert abs(t[2] - g["rf_t_gt"][i, 2]) < abs(g["rf_t_init"][i, 2] - g["rf_t_gt"][i, 2]) + 1e-3


def test_ransac_voting_layer_oracle_matches_reference_python_layer(golden_dir):
    """oracle.postproc.ransac_voting_layer against the reference's own ransac_voting_layer (ransac_voting_gpu.py:7-104 executed
    from source, its extension served by the reference kernels compiled for the host), replaying the recorded index draw
    (the reference draws ONE index set per image and reuses it in every round): voted keypoints within float32 LSQ noise,
    the too-few-pixels image gives zeros."""
    g = np.load(os.path.join(golden_dir, "pyref_golden.npz"))
    mask, vertex, win, idxs = g["rv_mask"], g["rv_vertex"], g["rv_win"], g["rv_idxs"]
    for i in range(2):
        got, n_iter = P.ransac_voting_layer(mask[i], vertex[i], [idxs[i]] * 8, inlier_thresh=0.99, max_iter=5)
        np.testing.assert_allclose(got, win[i], rtol=0, atol=2e-3)
        assert np.abs(got - g["rv_kpts"][i]).max() < 1.0
    z = P.ransac_voting_layer(mask[2], vertex[2], [idxs[0]], inlier_thresh=0.99, max_iter=5)
    assert not np.asarray(z if not isinstance(z, tuple) else z[0]).any() and not win[2].any()
