"""Host-side / oracle logic that needs no GPU: synthetic generator, decode + correspondence
restatement properties, depth-refine restatement recovering a known depth error."""
import os

import numpy as np

from gdrnpp_bop2022_amd import synthetic as S
from oracle import postproc as P


def _render_fn(verts, faces):
    def fn(obj, K, R, t, res):
        ds, xs = [], []
        for i in range(len(obj)):
            d, x = P.render_depth(verts[obj[i]], faces[obj[i]], K[i], R[i], t[i].astype(np.float64), res,
                                  want_xyz=True)
            ds.append(d)
            xs.append(x)
        return np.stack(ds), np.stack(xs)
    return fn


def make_case(b=6, seed=0, subdiv=3, num_classes=4):
    rng = np.random.default_rng(seed)
    verts, faces, ext = S.make_models(num_classes, rng, subdiv)
    det = S.make_detections(b, num_classes, ext, rng)
    maps = S.make_map_inputs(det, verts, faces, _render_fn(verts, faces), rng)
    return verts, faces, det, maps


def test_icosphere_counts():
    v, f = S.icosphere(4)
    assert v.shape == (2562, 3) and f.shape == (5120, 3)
    v, f = S.icosphere(3)
    assert v.shape == (642, 3) and f.shape == (1280, 3)


def test_get_out_mask_l1_range_and_constant_map():
    m = np.random.default_rng(0).standard_normal((3, 1, 8, 8)).astype(np.float32)
    o = P.get_out_mask(m)
    assert o.min() == 0 and o.max() == 1
    c = P.get_out_mask(np.ones((1, 1, 4, 4), np.float32))
    assert np.isnan(c).all()  # no epsilon in engine_utils.py:325


def test_correspondences_are_row_major_and_thresholded():
    verts, faces, det, maps = make_case()
    mask = P.get_out_mask(maps["mask"])
    for i in range(len(det["scale"])):
        xyz = np.concatenate([maps["coor_x"][i], maps["coor_y"][i], maps["coor_z"][i]], 0).transpose(1, 2, 0)
        c2 = maps["roi_coord_2d"][i].transpose(1, 2, 0)
        ip, mp, sel = P.get_img_model_points_with_coords2d(mask[i, 0], xyz, c2, 480, 640, det["roi_extent"][i])
        idx = np.flatnonzero(sel.reshape(-1))
        assert np.all(np.diff(idx) > 0) and len(idx) == len(ip) == len(mp)
        assert np.all(np.abs(mp) > 1e-4 * det["roi_extent"][i] * 0.999)
        assert len(idx) > 4


def test_depth_refine_restatement_removes_depth_error():
    verts, faces, det, maps = make_case(b=8, seed=5)
    mask = P.get_out_mask(maps["mask"])
    err0, err1 = [], []
    for i in range(8):
        o = int(det["roi_cls"][i])
        xyz = np.concatenate([maps["coor_x"][i], maps["coor_y"][i], maps["coor_z"][i]], 0).transpose(1, 2, 0)
        t = P.depth_refine_roi(xyz, mask[i, 0], maps["roi_depth"][i, 0], maps["K_crop"][i], det["R_gt"][i],
                               maps["t_init"][i], verts[o], faces[o])
        err0.append(abs(maps["t_init"][i][2] - det["t_gt"][i][2]))
        err1.append(abs(t[2] - det["t_gt"][i][2]))
    assert np.median(err1) < 0.25 * np.median(err0)
    assert np.median(err1) < 3e-3


def test_allocentric_to_egocentric_identity_on_axis():
    R = S.random_rotation(np.random.default_rng(1)).astype(np.float32)
    pose = np.hstack([R, np.array([[0], [0], [1.0]], np.float32)])
    assert np.array_equal(P.allocentric_to_egocentric_mat(pose)[:3, :3], R)
    pose[:, 3] = [0.2, -0.1, 0.9]
    Re = P.allocentric_to_egocentric_mat(pose)[:3, :3].astype(np.float64)
    np.testing.assert_allclose(Re @ Re.T, np.eye(3), atol=1e-6)


def test_zoom_K_maps_roi_to_output_square():
    det = S.make_detections(4, 3, np.full((3, 3), 0.1, np.float32), np.random.default_rng(2))
    Kc = P.zoom_K(det["roi_cam"], det["roi_center"], det["scale"], 64)
    # the ROI centre must land on (32, 32)
    for i in range(4):
        K = det["roi_cam"][i].astype(np.float64)
        c = det["roi_center"][i]
        ray = np.array([(c[0] - K[0, 2]) / K[0, 0], (c[1] - K[1, 2]) / K[1, 1], 1.0])
        uv = Kc[i].astype(np.float64) @ ray
        np.testing.assert_allclose(uv[:2] / uv[2], [32, 32], atol=1e-3)
    np.testing.assert_array_equal(Kc, S.zoom_K_np(det["roi_cam"], det["roi_center"], det["scale"], 64))


def test_warp_affine_restatement_properties():
    """Self-consistency of the cv2.warpAffine restatement (cv2 itself is unavailable: parity unpinned)."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    dep = rng.uniform(0.3, 2, (120, 160)).astype(np.float32)
    # scale == out size -> pure integer translation: bilinear == nearest == a slice
    a, b, c = P.crop_resize_roi(img, dep, (80.0, 60.0), 64.0, input_res=64, out_res=16)
    assert np.array_equal((a * 255).round().astype(np.uint8).transpose(1, 2, 0), img[28:92, 48:112])
    assert np.array_equal(b[0], dep[28:92, 48:112])
    # interior bilinear of the linear ramp coord_2d reproduces the ramp (to float32 rounding)
    ys, xs = np.mgrid[0:16, 0:16]
    np.testing.assert_allclose(c[0], (80 + (xs - 8) * 4.0) / 160.0, atol=2e-7)
    np.testing.assert_allclose(c[1], (60 + (ys - 8) * 4.0) / 120.0, atol=2e-7)
    # constant image stays constant inside, border value 0 outside
    const = np.full((120, 160, 3), 200, np.uint8)
    a, _, _ = P.crop_resize_roi(const, None, (0.0, 0.0), 100.0, input_res=64, out_res=16)
    v = (a * 255).round().astype(np.uint8)
    assert set(np.unique(v[:, 40:, 40:])) == {200} and set(np.unique(v[:, :30, :30])) == {0}
    M = P.get_affine_transform((80.0, 60.0), 64.0, 64)
    np.testing.assert_allclose(M, [[1, 0, -48], [0, 1, -28]], atol=1e-12)


def test_axangle2mat_restatement_against_scipy():
    """transforms3d.axangles.axangle2mat is restated from its published algorithm; scipy's Rotation is an
    independent implementation of the same map."""
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(4)
    for _ in range(20):
        axis = rng.standard_normal(3)
        ang = rng.uniform(0, np.pi)
        ref = Rotation.from_rotvec(axis / np.linalg.norm(axis) * ang).as_matrix()
        np.testing.assert_allclose(P.axangle2mat(axis, ang), ref, atol=1e-12)


def test_upnp_oracle_against_scipy_least_squares():
    """Independent check of the restated LM: MINPACK (scipy) minimises the same residuals to the same optimum."""
    from scipy.optimize import least_squares

    rng = np.random.default_rng(8)
    K = np.array([572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1.0])
    pn = 40
    rt = np.array([0.5, -0.4, 0.3, 0.02, 0.01, 0.8])
    p3 = rng.uniform(-0.08, 0.08, (pn, 3))
    th = np.linalg.norm(rt[:3]); k = rt[:3] / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    X = p3 @ (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).T + rt[3:]
    p2 = np.stack([K[0] * X[:, 0] / X[:, 2] + K[2], K[4] * X[:, 1] / X[:, 2] + K[5]], 1) + rng.normal(0, 0.7, (pn, 2))
    w = np.stack([rng.uniform(0.5, 2, pn), rng.uniform(-0.3, 0.3, pn), rng.uniform(0.5, 2, pn)], 1)
    init = rt + rng.uniform(-0.05, 0.05, 6)

    def res(x):
        return np.concatenate([P.upnp_residual(x, p2[i], p3[i], w[i], K)[0] for i in range(pn)])

    sol = least_squares(res, init, method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14).x
    out = P.uncertainty_pnp(p2, p3, w, K, init)
    np.testing.assert_allclose(out, sol, atol=1e-4)


def test_load_ply_ascii_and_binary(tmp_path):
    from gdrnpp_bop2022_amd.lib.pysixd.inout import load_ply

    v, f = S.icosphere(1)
    v = (v * 37.5).astype(np.float32)
    a = tmp_path / "a.ply"
    with open(a, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face %d\n"
                 "property list uchar int vertex_indices\nend_header\n" % (len(v), len(f)))
        for p in v:
            fh.write("%r %r %r 10 20 30\n" % (float(p[0]), float(p[1]), float(p[2])))
        for t in f:
            fh.write("3 %d %d %d\n" % tuple(t))
    b = tmp_path / "b.ply"
    with open(b, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                  "property float z\nelement face %d\nproperty list uchar uint vertex_index\nend_header\n"
                  % (len(v), len(f) // 2 + 1)).encode())
        fh.write(v.astype("<f4").tobytes())
        quads = 0
        for t in f[: len(f) // 2]:
            fh.write(b"\x03" + np.asarray(t, "<u4").tobytes())
        fh.write(b"\x04" + np.asarray([0, 1, 2, 3], "<u4").tobytes())  # one quad -> 2 triangles
    ma = load_ply(str(a), vertex_scale=0.001)
    mb = load_ply(str(b), vertex_scale=0.001)
    np.testing.assert_allclose(ma["pts"], v.astype(np.float64) * 0.001, rtol=1e-7)
    np.testing.assert_allclose(mb["pts"], v.astype(np.float64) * 0.001, rtol=1e-7)
    assert np.array_equal(ma["faces"], f) and ma["colors"].shape == (len(v), 3)
    assert len(mb["faces"]) == len(f) // 2 + 2 and np.array_equal(mb["faces"][-2:], [[0, 1, 2], [0, 2, 3]])


def test_yolox_postprocess_oracle_matches_plain_numpy_nms():
    """oracle/nms_oracle.c against an independent NumPy statement of the same rules on a small case (per-class NMS by
    looping over classes instead of the coordinate-offset trick: identical keeps when boxes stay well inside the offset)."""
    import numpy as np
    from oracle import postproc as P
    rng = np.random.default_rng(5)
    a, c = 400, 6
    det = np.zeros((1, a, 5 + c), np.float32)
    det[0, :, 0:2] = rng.uniform(50, 500, (a, 2)); det[0, :, 2:4] = rng.uniform(20, 150, (a, 2))
    det[0, :, 4] = rng.uniform(0, 1, a); det[0, :, 5:] = rng.uniform(0, 1, (a, c))
    out = P.yolox_postprocess(det, c, 0.25, 0.45, False)[0]
    x = det[0]
    boxes = np.stack([x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2, x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2], 1).astype(np.float32)
    cls = x[:, 5:].argmax(1); cc = x[:, 5:].max(1); score = (x[:, 4] * cc).astype(np.float32)
    idx = np.where(score >= np.float32(0.25))[0]
    order = idx[np.lexsort((idx, -score[idx]))]
    keep = []
    for i in order:
        ok = True
        for k in keep:
            if cls[k] != cls[i]:
                continue
            xx1, yy1 = max(boxes[k, 0], boxes[i, 0]), max(boxes[k, 1], boxes[i, 1])
            xx2, yy2 = min(boxes[k, 2], boxes[i, 2]), min(boxes[k, 3], boxes[i, 3])
            inter = max(xx2 - xx1, 0) * max(yy2 - yy1, 0)
            ak = (boxes[k, 2] - boxes[k, 0]) * (boxes[k, 3] - boxes[k, 1]); ai = (boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1])
            if inter / (ak + ai - inter) > 0.45 + 1e-4:
                ok = False
                break
            assert not (abs(inter / (ak + ai - inter) - 0.45) < 1e-4), "test case too close to the threshold"
        if ok:
            keep.append(i)
    assert len(out) == len(keep) and np.array_equal(out[:, 6].astype(int), cls[keep])
    assert np.allclose(out[:, :4], boxes[keep], atol=0) and np.array_equal(out[:, 4], x[keep, 4])


def test_paste_mask_oracle_matches_torch_grid_sample():
    """oracle/mask_rle_oracle.c against detectron2's `_do_paste_mask` restated with torch's own F.grid_sample (CPU):
    identical binary masks (no pixel within 1e-6 of the threshold in these cases), run lengths sum to H*W and decode back."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import postproc as P
    from gdrnpp_bop2022_amd.lib.utils import mask_utils as M
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:64, 0:64]
    H, W = 120, 160
    for k, box in enumerate([(30.3, 20.7, 110.9, 90.2), (-20.5, 40.0, 60.0, 130.5), (100.0, 5.0, 170.0, 60.0), (70.2, 50.1, 75.9, 58.7)]):
        m = (np.clip(1.4 - np.hypot(yy - 31.5 + 3 * k, xx - 31.5) / (12.0 + 3 * k), 0, 1) * 0.9 + 0.1 * rng.random((64, 64))).astype(np.float32)
        counts, binary = P.paste_mask_rle(m, box, H, W, 0.5, True)
        x0, y0, x1, y1 = [torch.tensor([[v]], dtype=torch.float32) for v in box]
        img_y = (torch.arange(0, H, dtype=torch.float32) + 0.5 - y0) / (y1 - y0) * 2 - 1
        img_x = (torch.arange(0, W, dtype=torch.float32) + 0.5 - x0) / (x1 - x0) * 2 - 1
        grid = torch.stack([img_x[:, None, :].expand(1, H, W), img_y[:, :, None].expand(1, H, W)], 3)
        val = F.grid_sample(torch.from_numpy(m)[None, None], grid, align_corners=False)[0, 0].numpy()
        assert (np.abs(val - 0.5) < 1e-6).sum() == 0
        assert np.array_equal(binary, (val >= 0.5).astype(np.uint8))
        assert sum(counts) == H * W and binary.sum() > 0
        rle = M.rle_from_counts(counts, H, W)
        assert np.array_equal(M.rle_to_binary_mask(rle), binary)


def test_rle_counts_equal_the_references_own_encoder():
    """The run lengths against the reference's `binary_mask_to_rle(mask, compressed=False)` (/root/reference/lib/utils/
    mask_utils.py:96-109, executed from source by tests/golden/make_golden_pyref.py: `rle_*` keys of pyref_golden.npz): the four pasted
    instance masks of the test above, an empty, a full, a first-pixel, a last-pixel and a random mask.  Pins the ENCODER half of the
    SAVE_RESULTS_ONLY writer to the reference; the paste half (detectron2) stays restated."""
    import numpy as np
    from oracle import postproc as P
    from gdrnpp_bop2022_amd.lib.utils import mask_utils as M
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "pyref_golden.npz"))
    masks, lens = fx["rle_masks"], fx["rle_counts_len"]
    want = np.split(fx["rle_counts"], np.cumsum(lens)[:-1])
    H, W = masks.shape[1:]
    for k in range(4):                                   # paste + encode: same binary mask, the reference's run lengths
        counts, binary = P.paste_mask_rle(fx["rle_soft"][k], fx["rle_boxes"][k], H, W, 0.5, True)
        assert np.array_equal(binary, masks[k]) and counts == want[k].tolist(), k
    for k in range(len(masks)):                          # encode alone: the mask pasted onto itself (box = the image: identity sampling)
        counts = P.paste_mask_rle(masks[k].astype(np.float32), (0.0, 0.0, float(W), float(H)), H, W, 0.5)
        assert counts == want[k].tolist(), k
        assert np.array_equal(M.rle_to_binary_mask(M.rle_from_counts(counts, H, W)), masks[k])
        assert np.array_equal(M.rle_to_binary_mask(M.rle_from_counts(counts, H, W, compressed=False)), masks[k])
    assert want[4].tolist() == [H * W] and want[5].tolist() == [0, H * W] and want[6][0] == 0


def test_coco_rle_string_codec():
    from gdrnpp_bop2022_amd.lib.utils import mask_utils as M
    assert M.rle_counts_to_string([307200]) == "PP\\9"          # pycocotools' encoding of an empty 480x640 mask
    for counts in ([10, 3, 7], [0, 5, 100000, 2, 3, 40, 1], [5, 1, 5, 1, 5, 1, 700, 33, 2], [1, 1, 1, 1, 1, 1, 1, 1, 1, 1]):
        assert M.rle_string_to_counts(M.rle_counts_to_string(counts)) == counts
