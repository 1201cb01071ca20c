"""Golden vectors for the ROI crop from the reference's own Python (authoring container only).

core/utils/data_utils.py builds the warp matrix in plain NumPy — get_affine_transform (:136-184) with get_dir (:199-206) and
get_3rd_point (:194-196) — and hands three point pairs to cv2.getAffineTransform.  cv2 is not installed here, so that ONE call
is served by a float64 LU solve of the same 6x6 system OpenCV sets up (imgproc/imgwarp.cpp, getAffineTransform: rows
[x y 1 0 0 0] / [0 0 0 x y 1], solved with DECOMP_LU in double); everything else — float32 point construction, direction
vector, third point, argument order of crop_resize_by_warp_affine (:115-133) — is the reference's text, cut with `ast` and
executed unmodified.  The interpolation itself (cv2.warpAffine) cannot be pinned this way: crop_golden.npz holds the matrices
and, for the call plumbing, the (matrix, dsize, flags) triples the reference passes to cv2.warpAffine.
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def cut(path, name):
    src = open(os.path.join(REF, path)).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            lines = src.splitlines()[node.lineno - 1:node.end_lineno]
            indent = len(lines[0]) - len(lines[0].lstrip())
            return "\n".join(l[indent:] for l in lines) + "\n"
    raise KeyError(name)


def get_affine_transform_lu(src, dst):
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        A[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def main():
    calls = []

    def warp_affine(img, trans, dsize, flags=None):
        calls.append((np.array(trans, np.float64), tuple(int(v) for v in dsize), flags))
        return np.zeros((dsize[1], dsize[0]) + img.shape[2:], img.dtype)

    cv2 = types.SimpleNamespace(getAffineTransform=get_affine_transform_lu, warpAffine=warp_affine, INTER_LINEAR=1,
                                INTER_NEAREST=0)
    ns = dict(np=np, cv2=cv2)
    path = "core/utils/data_utils.py"
    for name in ("get_dir", "get_3rd_point", "get_affine_transform", "crop_resize_by_warp_affine"):
        exec(compile(cut(path, name), os.path.join(REF, path), "exec"), ns)
    rng = np.random.default_rng(20220925 + 11)
    n = 24
    centers = np.stack([rng.uniform(20, 620, n), rng.uniform(20, 460, n)], 1).astype(np.float32)
    scales = rng.uniform(30, 640, n).astype(np.float32)
    scales[:3] = [64.0, 256.0, 640.0]
    centers[0] = [80.0, 60.0]
    Ms = {}
    for res in (256, 64):
        Ms[res] = np.stack([ns["get_affine_transform"](centers[i], float(scales[i]), 0, res) for i in range(n)])
    # call plumbing of read_data_test's three crops (data_loader.py:773-797): image bilinear 256, depth nearest 256, coord bilinear 64
    img = np.zeros((480, 640, 3), np.uint8)
    ns["crop_resize_by_warp_affine"](img, centers[5], float(scales[5]), 256, interpolation=cv2.INTER_LINEAR)
    ns["crop_resize_by_warp_affine"](img[:, :, :1].astype(np.float32), centers[5], float(scales[5]), 256, interpolation=cv2.INTER_NEAREST)
    ns["crop_resize_by_warp_affine"](np.zeros((480, 640, 2), np.float32), centers[5], float(scales[5]), 64, interpolation=cv2.INTER_LINEAR)
    assert [c[1] for c in calls] == [(256, 256), (256, 256), (64, 64)] and [c[2] for c in calls] == [1, 0, 1]
    out = dict(centers=centers, scales=scales, M256=Ms[256], M64=Ms[64], call_M=np.stack([c[0] for c in calls]),
               call_dsize=np.array([c[1] for c in calls]), call_flags=np.array([c[2] for c in calls]))
    np.savez_compressed(os.path.join(HERE, "crop_golden.npz"), **out)
    print("crop_golden.npz", os.path.getsize(os.path.join(HERE, "crop_golden.npz")), "bytes;  M256[0] =", Ms[256][0].tolist())


if __name__ == "__main__":
    sys.exit(main())
