"""core/csrc/fps/fps_utils.py:6-21, verbatim interface: ``farthest_point_sampling(pts, sn, init_center=False)``
-> the sampled points ``pts[idxs]`` (f32[sn,3])."""
import numpy as np

from ._ext import ffi, lib


def farthest_point_sampling(pts, sn, init_center=False):
    pn, _ = pts.shape
    assert pts.shape[1] == 3
    pts = np.ascontiguousarray(pts, np.float32)
    idxs = np.ascontiguousarray(np.zeros([sn], np.int32))
    pts_ptr = ffi.cast("float*", pts.ctypes.data)
    idxs_ptr = ffi.cast("int*", idxs.ctypes.data)
    if init_center:
        lib.farthest_point_sampling_init_center(pts_ptr, idxs_ptr, pn, sn)
    else:
        lib.farthest_point_sampling(pts_ptr, idxs_ptr, pn, sn)
    if (idxs < 0).any():
        raise RuntimeError("farthest_point_sampling failed on the device (see stderr)")
    return pts[idxs]


def farthest_point_sampling_idx(pts, sn, init_center=False):
    """Same call, returning the indices (what tools/ycbv/ycbv_1_compute_fps.py effectively stores)."""
    pts = np.ascontiguousarray(pts, np.float32)
    idxs = np.zeros([sn], np.int32)
    fn = lib.farthest_point_sampling_init_center if init_center else lib.farthest_point_sampling
    fn(ffi.cast("float*", pts.ctypes.data), ffi.cast("int*", idxs.ctypes.data), pts.shape[0], sn)
    return idxs
