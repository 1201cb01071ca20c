"""Reference-anchored and independent checks of the oracles whose third-party arithmetic (cv2, detectron2, torchvision) is
not installed here — SURVEY.md §8 rows a1, a1b, f3.  They do not make those oracles bit-pinned (only the third-party
libraries themselves could); they pin what CAN be pinned: the reference's own matrix construction, the sampling geometry
against independent implementations, and closed-form answers."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import postproc as P


def test_affine_matrix_matches_reference_get_affine_transform(golden_dir):
    """oracle get_affine_transform == the reference's get_affine_transform / get_dir / get_3rd_point executed from source
    (tests/golden/make_golden_crop.py; cv2.getAffineTransform served by a float64 LU solve of OpenCV's 6x6 system)."""
    g = np.load(os.path.join(golden_dir, "crop_golden.npz"))
    for res, key in ((256, "M256"), (64, "M64")):
        for i in range(len(g["scales"])):
            M = P.get_affine_transform(g["centers"][i], float(g["scales"][i]), res)
            np.testing.assert_allclose(M, g[key][i], rtol=0, atol=1e-9)
    # the three warpAffine calls of read_data_test: same matrix for image and depth at 256, its own matrix at 64, dsize (w, h)
    np.testing.assert_allclose(g["call_M"][0], g["M256"][5], atol=0)
    np.testing.assert_allclose(g["call_M"][1], g["M256"][5], atol=0)
    np.testing.assert_allclose(g["call_M"][2], g["M64"][5], atol=0)
    assert g["call_flags"].tolist() == [1, 0, 1]   # INTER_LINEAR, INTER_NEAREST, INTER_LINEAR


def _grid_sample_affine(img_chw, M, out, mode):
    """dst(x, y) = src(M^-1 [x, y, 1]) with pixel centres at integer coordinates, zero border — ATen's grid_sample."""
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)
    ys, xs = np.mgrid[0:out, 0:out].astype(np.float64)
    sx = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
    sy = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    h, w = img_chw.shape[1:]
    grid = np.stack([2 * sx / (w - 1) - 1, 2 * sy / (h - 1) - 1], -1)[None]
    t = torch.from_numpy(np.ascontiguousarray(img_chw, np.float64))[None]
    o = F.grid_sample(t, torch.from_numpy(grid), mode=mode, padding_mode="zeros", align_corners=True)
    return o[0].numpy(), sx, sy


def _smooth_image(rng, h, w, c, amp):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127.5 + amp * np.sin(xx / (9.0 + k) + rng.uniform(0, 6)) * np.cos(yy / (7.0 + 2 * k) + rng.uniform(0, 6))
                    for k in range(c)], -1)
    return img


@pytest.mark.parametrize("center,scale", [((300.0, 220.0), 180.0), ((90.5, 400.25), 97.3), ((600.0, 40.0), 250.0),
                                          ((320.0, 240.0), 640.0)])
def test_warp_u8_bilinear_against_float_bilinear(center, scale):
    """The 8-bit path of the cv2.warpAffine restatement (coordinates in 1/32 px, 15-bit weights) against an exact float
    bilinear of the same source pixels (F.grid_sample): they may differ only by the coordinate quantisation (<= 1/32 px per
    axis times the image gradient) plus the final rounding.  A half-pixel convention error would be 16x larger."""
    rng = np.random.default_rng(3)
    img = np.clip(np.rint(_smooth_image(rng, 480, 640, 3, 110.0)), 0, 255).astype(np.uint8)
    M = P.get_affine_transform(center, scale, 256)
    got = P.warp_affine(img, M, 256).astype(np.float64).transpose(2, 0, 1)
    want, sx, sy = _grid_sample_affine(img.astype(np.float64).transpose(2, 0, 1), M, 256, "bilinear")
    inside = (sx >= 0) & (sx <= 639) & (sy >= 0) & (sy <= 479)
    gy, gx = np.gradient(img.astype(np.float64), axis=(0, 1))
    gmax = max(np.abs(gx).max(), np.abs(gy).max())             # levels per source pixel
    tol = 0.5 + 2 * gmax / 32.0 + 0.05
    err = np.abs(got - want)[:, inside]
    assert err.max() <= tol, (err.max(), tol)
    assert err.mean() < 0.45                                     # ~ the rounding error of an exact bilinear
    # a half-pixel shift of the sampling grid would NOT pass
    shifted, _, _ = _grid_sample_affine(img.astype(np.float64).transpose(2, 0, 1), M + np.array([[0, 0, 0.5 * M[0, 0]], [0, 0, 0]]),
                                        256, "bilinear")
    assert np.abs(got - shifted)[:, inside].mean() > 2 * err.mean()
    # outside the source image the border value is 0 (pixels whose four neighbours all lie outside)
    far = (sx < -1) | (sx > 640) | (sy < -1) | (sy > 480)
    assert not got[:, far].any()


def test_warp_float_bilinear_and_nearest_against_grid_sample():
    """Float path (roi_coord_2d: 2-channel ramp, INTER_LINEAR) and depth path (INTER_NEAREST) against grid_sample."""
    rng = np.random.default_rng(4)
    c2d = P.get_2d_coord_np(640, 480)
    M64 = P.get_affine_transform((311.3, 207.9), 143.7, 64)
    got = P.warp_affine(c2d, M64, 64).transpose(2, 0, 1).astype(np.float64)
    want, sx, sy = _grid_sample_affine(c2d.transpose(2, 0, 1), M64, 64, "bilinear")
    # ramp slope 1/640 (1/480) per source pixel, coordinate quantisation 1/32 px
    assert np.abs(got[0] - want[0]).max() <= 1.0 / 640 / 32 + 1e-6
    assert np.abs(got[1] - want[1]).max() <= 1.0 / 480 / 32 + 1e-6
    dep = rng.uniform(0.3, 2.0, (480, 640)).astype(np.float32)
    M = P.get_affine_transform((311.3, 207.9), 143.7, 256)
    gotn = P.warp_affine(dep, M, 256, nearest=True)
    wantn, sx, sy = _grid_sample_affine(dep[None].astype(np.float64), M, 256, "nearest")
    frac = np.minimum(np.abs(sx - np.floor(sx) - 0.5), np.abs(sy - np.floor(sy) - 0.5))
    clear = frac > 1.0 / 1024 + 1e-9     # away from rounding ties: cv2 rounds the 10-bit fixed-point coordinate
    assert np.array_equal(gotn[clear], wantn[0][clear].astype(np.float32))
    assert clear.mean() > 0.99


def _roi_align_independent(x, rois, out, sampling_ratio=0):
    """ROIAlign (aligned=True) written from its definition with torch ops: every bin is the mean of a regular grid of
    bilinear samples; sample (y, x) outside [-1, H] x [-1, W] contributes 0, coordinates are clamped to the image."""
    b, c, h, w = x.shape
    res = []
    xt = torch.from_numpy(x).double()
    for r in rois:
        bi = int(r[0])
        x1, y1, x2, y2 = [float(v) - 0.5 for v in r[1:]]
        bw, bh = (x2 - x1) / out, (y2 - y1) / out
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil((y2 - y1) / out))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil((x2 - x1) / out))
        acc = torch.zeros(c, out, out, dtype=torch.float64)
        for iy in range(gh):
            ys = y1 + (torch.arange(out, dtype=torch.float64) + (iy + 0.5) / gh) * bh
            for ix in range(gw):
                xs = x1 + (torch.arange(out, dtype=torch.float64) + (ix + 0.5) / gw) * bw
                valid = ((ys >= -1) & (ys <= h))[:, None] & ((xs >= -1) & (xs <= w))[None, :]
                yc, xc = ys.clamp(0, h - 1), xs.clamp(0, w - 1)
                y0, x0 = yc.floor().long().clamp(max=h - 1), xc.floor().long().clamp(max=w - 1)
                y1i, x1i = (y0 + 1).clamp(max=h - 1), (x0 + 1).clamp(max=w - 1)
                ly, lx = (yc - y0)[:, None], (xc - x0)[None, :]
                img = xt[bi]
                v = (img[:, y0][:, :, x0] * (1 - ly) * (1 - lx) + img[:, y0][:, :, x1i] * (1 - ly) * lx
                     + img[:, y1i][:, :, x0] * ly * (1 - lx) + img[:, y1i][:, :, x1i] * ly * lx)
                acc += v * valid
        res.append(acc / (gh * gw))
    return torch.stack(res).numpy()


def test_roi_align_oracle_against_independent_formulation():
    """oracle/roi_align_oracle.c (detectron2 ROIAlign restated) against the definition written with torch tensor ops, on
    boxes inside, across and beyond the image border, adaptive and fixed sampling ratios."""
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 3, 37, 45)).astype(np.float32)
    rois = np.array([[0, 5.3, 4.1, 30.7, 28.9], [1, 10.0, 8.0, 26.0, 24.0], [0, -6.0, -3.5, 12.2, 9.9],
                     [1, 30.0, 20.0, 60.0, 50.0], [0, 0.0, 0.0, 45.0, 37.0], [1, 7.25, 3.5, 9.0, 5.0]], np.float32)
    for sr in (0, 2):
        got = P.roi_align(x, rois, 8, 1.0, sr, True)
        want = _roi_align_independent(x, rois, 8, sr)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def _preds(boxes, scores, classes, num_classes):
    """YOLOX head rows (cx, cy, w, h, obj, class scores) for given corner boxes."""
    a = len(boxes)
    p = np.zeros((1, a, 5 + num_classes), np.float32)
    b = np.asarray(boxes, np.float32)
    p[0, :, 0] = (b[:, 0] + b[:, 2]) / 2
    p[0, :, 1] = (b[:, 1] + b[:, 3]) / 2
    p[0, :, 2] = b[:, 2] - b[:, 0]
    p[0, :, 3] = b[:, 3] - b[:, 1]
    p[0, :, 4] = 1.0
    for i, (s, c) in enumerate(zip(scores, classes)):
        p[0, i, 5 + c] = s
    return p


def test_nms_oracle_closed_form_cases():
    """torchvision.ops.nms semantics restated in oracle/nms_oracle.c, on cases whose answer follows from the definition:
    IoU([0,0,10,10],[1,1,11,11]) = 81/119 > 0.45 suppresses the lower score; IoU exactly 0.5 at threshold 0.5 does NOT
    suppress (the test is iou > thr); boxes of different classes never suppress each other (batched_nms) unless
    class_agnostic; ties keep the earlier box; the output is ordered by descending score."""
    boxes = [[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]]
    out = P.yolox_postprocess(_preds(boxes, [0.9, 0.8, 0.95, 0.85], [0, 0, 0, 1], 2), 2, 0.1, 0.45)[0]
    assert out.shape == (3, 7)
    np.testing.assert_allclose(out[:, :4], [[20, 20, 30, 30], [0, 0, 10, 10], [0, 0, 10, 10]], atol=1e-5)
    assert out[:, 6].tolist() == [0.0, 0.0, 1.0]
    np.testing.assert_allclose(out[:, 5], [0.95, 0.9, 0.85], atol=1e-6)
    out = P.yolox_postprocess(_preds(boxes, [0.9, 0.8, 0.95, 0.85], [0, 0, 0, 1], 2), 2, 0.1, 0.45, class_agnostic=True)[0]
    assert out.shape == (2, 7) and out[:, 5].tolist() == pytest.approx([0.95, 0.9])
    # IoU == threshold: A = [0,0,2,1] (area 2), B = [0,0,1,1] inside A: inter 1, union 2 -> 0.5
    out = P.yolox_postprocess(_preds([[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 1), 1, 0.1, 0.5)[0]
    assert out.shape[0] == 2
    out = P.yolox_postprocess(_preds([[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], [0, 0], 1), 1, 0.1, 0.4999)[0]
    assert out.shape[0] == 1
    # chain A-B-C: B is suppressed by A, so C (overlapping only B) survives
    out = P.yolox_postprocess(_preds([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [0.9, 0.8, 0.7], [0, 0, 0], 1), 1, 0.1, 0.4)[0]
    np.testing.assert_allclose(out[:, 0], [0, 8], atol=1e-5)
    # score threshold: obj * class_conf >= conf_thre
    assert P.yolox_postprocess(_preds([[0, 0, 10, 10]], [0.5], [0], 1), 1, 0.7, 0.45)[0] is None


def test_roi_pool_oracle_against_an_independent_formulation_and_closed_forms():
    """RoIPool (batch_crop_resize(interpolation="nearest"), core/utils/zoom_utils.py:92-93): torchvision is absent, so the
    restatement (oracle/roi_align_oracle.c, parity unpinned) is held against an independent NumPy formulation of the published
    definition and against closed forms: a pixel-aligned box with one pixel per bin returns the pixels themselves; a box outside
    the image returns zeros; with the box equal to the image and an evenly dividing output it equals max pooling."""
    import math

    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, 3, 17, 23)).astype(np.float32)
    rois = np.array([[0, 2, 3, 9, 10], [1, -4.4, -2.6, 6.5, 7.49], [0, 30, 30, 40, 40], [1, 0.49, 0.51, 21.7, 15.2],
                     [0, 5, 5, 5, 5], [1, 10.5, 2.5, 3.5, 1.5]], np.float32)
    out = P.roi_pool(x, rois, (4, 5))

    def rnd(v):                                    # std::round: half away from zero
        return int(math.floor(abs(float(v)) + 0.5) * (1 if v >= 0 else -1))

    for n, r in enumerate(rois):
        sw, sh, ew, eh = rnd(r[1]), rnd(r[2]), rnd(r[3]), rnd(r[4])
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = np.float32(rh) / np.float32(4), np.float32(rw) / np.float32(5)
        for ph in range(4):
            for pw in range(5):
                h0 = min(max(int(np.floor(np.float32(ph) * bh)) + sh, 0), 17)
                h1 = min(max(int(np.ceil(np.float32(ph + 1) * bh)) + sh, 0), 17)
                w0 = min(max(int(np.floor(np.float32(pw) * bw)) + sw, 0), 23)
                w1 = min(max(int(np.ceil(np.float32(pw + 1) * bw)) + sw, 0), 23)
                want = x[int(r[0]), :, h0:h1, w0:w1].reshape(3, -1).max(1) if (h1 > h0 and w1 > w0) else np.zeros(3, np.float32)
                assert np.array_equal(out[n, :, ph, pw], want), (n, ph, pw)
    assert not out[2].any()                                                        # box outside the image
    px = P.roi_pool(x, np.array([[0, 4, 6, 8, 9]], np.float32), (4, 5))            # 5 x 4 pixels, one per bin
    assert np.array_equal(px[0], x[0, :, 6:10, 4:9])
    whole = P.roi_pool(x[:, :, :16, :20], np.array([[1, 0, 0, 19, 15]], np.float32), (4, 5))
    assert np.array_equal(whole[0], F.max_pool2d(torch.from_numpy(x[1:2, :, :16, :20]), 4)[0].numpy())


def test_nms_oracle_equals_the_reference_postprocess_executed_from_source(golden_dir):
    """yolox_golden.npz = the reference's ``postprocess`` (det/yolox/utils/boxes.py:34-74) run from its source text with only the
    torchvision NMS primitive served by a stand-in (tests/golden/make_golden_yolox.py): the oracle reproduces counts, rows and keep
    order bit for bit — corner conversion, class argmax, the obj * class >= thr mask, the 7-column layout and the None of an empty
    image are therefore the reference's own; what stays a restatement is the NMS primitive."""
    z = np.load(os.path.join(golden_dir, "yolox_golden.npz"))
    total = 0
    for name in "abcd":
        c, conf, thr, agn = z[name + "_args"]
        outs = P.yolox_postprocess(z[name + "_det"], int(c), float(conf), float(thr), bool(agn))
        assert [0 if o is None else len(o) for o in outs] == z[name + "_count"].tolist(), name
        cat = np.concatenate([np.zeros((0, 7), np.float32)] + [o for o in outs if o is not None])
        assert np.array_equal(cat, z[name + "_out"]), name
        total += len(cat)
    assert total > 400 and z["d_count"].sum() == 0


def _readdata_case(golden_dir):
    import sys
    sys.path.insert(0, golden_dir)
    z = np.load(os.path.join(golden_dir, "readdata_golden.npz"))
    rng = np.random.default_rng(20220925 + 71)                 # tests/golden/make_golden_readdata.case()
    image = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    depth = (rng.integers(300, 2000, (480, 640)).astype(np.uint16) / 1000.0).astype(np.float32)
    return z, image, depth


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_roi_plumbing_equals_the_reference_read_data_test(golden_dir):
    """readdata_golden.npz = the reference's own ``read_data_test`` (data_loader.py:647-818) executed from source on one seeded image
    with six detections (tests/golden/make_golden_readdata.py).  engine.rois_from_detections reproduces its per-ROI scalars exactly —
    XYWH -> XYXY, centre, scale = min(max(w, h) * 1.5, 640), roi_wh clamped to >= 1, resize_ratio — and the oracle's crop chain
    (fed those scalars) reproduces every crop BYTE FOR BYTE (SHA-256), i.e. which array is warped with which interpolation to which
    size, the fp64 normalisation and the float32 casts are the reference's."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    z, image, depth = _readdata_case(golden_dir)
    b = z["boxes_xywh"]
    xyxy = np.stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], 1)
    r = engine.rois_from_detections(xyxy, 480, 640, 1.5, 64)
    assert np.array_equal(r["bbox_center"].astype(np.float32), z["abs_bbox_center"]) and np.array_equal(r["scale"], z["abs_scale"])
    assert np.array_equal(r["roi_wh"], z["abs_roi_wh"]) and np.array_equal(r["resize_ratio"], z["abs_resize_ratio"])
    assert z["abs_scale"].tolist()[2] == 640.0 and z["abs_roi_wh"][3, 0] == 1.0            # the clamps are exercised
    assert str(z["abs_scale_dtype"]) == "float64" and str(z["abs_bbox_center_dtype"]) == "float32" and str(z["abs_roi_cls_dtype"]) == "int64"
    for i in range(len(b)):
        img, dep, c2d = P.crop_resize_roi(image, depth, r["bbox_center"][i], float(r["scale"][i]))
        assert _sha(img) == str(z["abs_roi_img_sha256"][i]) and _sha(dep) == str(z["abs_roi_depth_sha256"][i]), i
        assert _sha(c2d) == str(z["abs_roi_coord_2d_sha256"][i]), i
        rel = ((r["bbox_center"][i].reshape(2, 1, 1) - c2d * np.array([640, 480]).reshape(2, 1, 1)) / r["scale"][i]).astype(np.float32)
        assert _sha(rel) == str(z["rel_roi_coord_2d_rel_sha256"][i]), i
