"""Replaces the cffi-generated ``core.csrc.uncertainty_pnp._ext``: ``lib.uncertainty_pnp(double*…, int)``
(core/csrc/uncertainty_pnp/src/ext.h:1-9) now runs the HIP LM kernel."""
from .._cffi_like import make

ffi, lib = make(["uncertainty_pnp"])
