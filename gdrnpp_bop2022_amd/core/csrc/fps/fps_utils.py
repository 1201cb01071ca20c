"""Call surface of the reference's ``core/csrc/fps/fps_utils.py`` (``farthest_point_sampling(pts, sn, init_center)``
returning the sampled points), routed through the host-pointer symbols of ``libgdrnpp_hip.so`` that keep the
reference's C prototypes (core/csrc/fps/src/ext.h:1-14).  ``farthest_point_sampling_idx`` additionally exposes the
index list, which is what the offline tools persist (tools/ycbv/ycbv_1_compute_fps.py:29-37)."""
import numpy as np

from ._ext import ffi, lib


def farthest_point_sampling_idx(pts, sn, init_center=False):
    """Indices i32[sn] of the farthest-point sample of ``pts`` f32[pn,3]."""
    cloud = np.ascontiguousarray(pts, dtype=np.float32)
    if cloud.ndim != 2 or cloud.shape[1] != 3:
        raise ValueError(f"pts must be [pn, 3], got {cloud.shape}")
    picked = np.full((int(sn),), -1, dtype=np.int32)
    entry = lib.farthest_point_sampling_init_center if init_center else lib.farthest_point_sampling
    entry(ffi.cast("float*", cloud.ctypes.data), ffi.cast("int*", picked.ctypes.data), cloud.shape[0], int(sn))
    if picked.min() < 0:  # the C symbol has no status channel: -1 marks a device failure
        raise RuntimeError("farthest_point_sampling failed on the device (see stderr)")
    return picked


def farthest_point_sampling(pts, sn, init_center=False):
    return np.ascontiguousarray(pts, dtype=np.float32)[farthest_point_sampling_idx(pts, sn, init_center)]
