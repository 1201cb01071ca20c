"""ORACLE — test infrastructure only.

CPU restatements of the GDRNPP inference hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``gdrnpp_bop2022_amd``) never does.

* ``oracle/*.c``        line-cited C restatements, built into ``oracle/liboracle.so``
* ``oracle/postproc.py`` NumPy restatements of the evaluator / engine_utils math
* ``oracle/_ref/``      the reference's OWN sources compiled where they lie under
                        ``/root/reference`` (only in the authoring container; outputs are
                        git-ignored but travel to the GPU box with the snapshot)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF_DIR = os.path.join(_HERE, "_ref")
REFERENCE_ROOT = "/root/reference"

_C_SOURCES = ["fps_oracle.c", "nnd_oracle.c", "ransac_voting_oracle.c", "upnp_oracle.c", "raster_oracle.c",
              "warp_oracle.c", "roi_align_oracle.c", "flow_oracle.c", "nms_oracle.c", "mask_rle_oracle.c"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str:
    """Compile the C restatements (plain gcc, -O2, no FMA contraction — like the reference's
    x86-64 build flags, core/csrc/fps/setup.py:5-7)."""
    srcs = [os.path.join(_HERE, s) for s in _C_SOURCES if os.path.exists(os.path.join(_HERE, s))]
    if not force and _newer(_LIB, srcs):
        return _LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", _LIB] + srcs + ["-lm"]
    subprocess.run(cmd, check=True)
    return _LIB


def build_ref(force: bool = False) -> dict:
    """Compile the reference's own CPU sources, unmodified, from /root/reference into
    oracle/_ref/.  No-op (returns what already exists) when the reference tree is absent,
    e.g. on the GPU box."""
    os.makedirs(_REF_DIR, exist_ok=True)
    out = {}
    fps_so = os.path.join(_REF_DIR, "libfps_ref.so")
    fps_src = os.path.join(REFERENCE_ROOT, "core/csrc/fps/src/farthest_point_sampling.cpp")
    if os.path.exists(fps_src) and (force or not os.path.exists(fps_so)):
        # flags from core/csrc/fps/setup.py:5-7 (without -fopenmp: the source has no pragmas)
        subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++11", "-o", fps_so, fps_src], check=True)
    if os.path.exists(fps_so):
        out["fps"] = fps_so

    nnd_so = os.path.join(_REF_DIR, "libnnd_ref.so")
    nnd_src = os.path.join(REFERENCE_ROOT, "core/csrc/torch_nndistance/src/nnd_cpu.cpp")
    shim = os.path.join(_HERE, "ref_shims", "nnd_ref_shim.cpp")
    if os.path.exists(nnd_src) and os.path.exists(shim) and (force or not os.path.exists(nnd_so)):
        # nnsearch() is plain C++; the torch entry points around it are not needed.  The shim
        # supplies a minimal <torch/torch.h> stand-in so the file compiles without libtorch.
        subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++14", "-I", os.path.join(_HERE, "ref_shims"),
                        "-o", nnd_so, shim, "-DNND_SRC=\"%s\"" % nnd_src], check=True)
    if os.path.exists(nnd_so):
        out["nnd"] = nnd_so

    upnp_so = os.path.join(_REF_DIR, "libupnp_ref.so")
    upnp_drv = os.path.join(_HERE, "ref_shims", "upnp_ceres_jet_driver.cpp")
    ceres_inc = os.path.join(REFERENCE_ROOT, "core/csrc/uncertainty_pnp/include")
    if os.path.isdir(ceres_inc) and os.path.exists(upnp_drv) and (force or not os.path.exists(upnp_so)):
        r = subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++14", "-I", ceres_inc,
                            "-I", os.path.join(ceres_inc, "eigen3"), "-o", upnp_so, upnp_drv],
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write("[oracle] ceres-header driver did not build (kept unpinned):\n" + r.stderr[-2000:])
    if os.path.exists(upnp_so):
        out["upnp"] = upnp_so

    flow_so = os.path.join(_REF_DIR, "libflow_ref.so")
    flow_src = os.path.join(REFERENCE_ROOT, "core/csrc/flow/src/flow_cpu.cpp")
    flow_shim = os.path.join(_HERE, "ref_shims", "flow_ref_shim.cpp")
    if os.path.exists(flow_src) and os.path.exists(flow_shim) and (force or not os.path.exists(flow_so)):
        # flow_kernel<scalar_t>() is plain C++; ref_shims/torch/extension.h stands in for libtorch's header
        subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++14", "-I", os.path.join(_HERE, "ref_shims"),
                        "-o", flow_so, flow_shim, "-DFLOW_SRC=\"%s\"" % flow_src], check=True)
    if os.path.exists(flow_so):
        out["flow"] = flow_so

    rv_so = os.path.join(_REF_DIR, "libransac_ref.so")
    rv_src = os.path.join(REFERENCE_ROOT, "core/csrc/ransac_voting/src/ransac_voting_kernel.cu")
    rv_shim = os.path.join(_HERE, "ref_shims", "ransac_ref_shim.cpp")
    if os.path.exists(rv_src) and os.path.exists(rv_shim) and (force or not os.path.exists(rv_so)):
        # no nvcc: the four kernels (plain C per thread, no shared memory / atomics / barriers) are extracted verbatim
        # into a generated include under _ref/ and compiled for the host behind a grid emulator
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:   # the extracted text is a build intermediate: only the .so is kept
            inc = os.path.join(tmp, "ransac_kernels_extracted.inc")
            subprocess.run([sys.executable, os.path.join(_HERE, "ref_shims", "extract_cuda_kernels.py"), rv_src, inc], check=True)
            subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++14", "-o", rv_so, rv_shim,
                            "-DRANSAC_KERNELS_INC=\"%s\"" % inc], check=True)
    if os.path.exists(rv_so):
        out["ransac"] = rv_so
    return out


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def ref_lib(name: str):
    """ctypes handle on a compiled reference library, or None if it was never built."""
    path = {"fps": "libfps_ref.so", "nnd": "libnnd_ref.so", "upnp": "libupnp_ref.so", "flow": "libflow_ref.so", "ransac": "libransac_ref.so"}[name]
    path = os.path.join(_REF_DIR, path)
    if not os.path.exists(path):
        build_ref()
    return ctypes.CDLL(path) if os.path.exists(path) else None
