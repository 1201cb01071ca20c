"""Replaces the cffi-generated ``core.csrc.fps._ext`` (core/csrc/fps/_ext.c:586-713): same ``ffi``/``lib`` pair,
``lib.farthest_point_sampling[_init_center](float*, int*, int, int)`` now runs the HIP kernel."""
from .._cffi_like import make

ffi, lib = make(["farthest_point_sampling", "farthest_point_sampling_init_center"])
