"""The packed-fp32 / MFMA hazard of MI355X as a function (tests/test_gpu_stream_guard.py, bench.py's ``pk_hazard_probe`` key):
the RAW probe of tools/probe/pk_probe2.hip — every op_sel / op_sel_hi form of v_pk_add_f32 / v_pk_mul_f32 checked against the
scalar instruction in the same lane — launched on one stream while a companion runs on another.  Diagnostic code, not part of
the product: the product library holds none of these instruction forms (tools/check_isa_hazards.py refuses to link them)."""
import collections
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE_SO = os.path.join(HERE, "probe", "libpk_probe.so")
MAXLOG = 16384


def load_probe(build: bool = True):
    """The probe library (built on demand with hipcc: tools/probe/build.sh), or None when it cannot be had."""
    if not os.path.exists(PROBE_SO) and build:
        try:
            subprocess.run(["bash", os.path.join(HERE, "probe", "build.sh")], check=True, capture_output=True, timeout=600)
        except Exception:
            return None
    if not os.path.exists(PROBE_SO):
        return None
    lib = ctypes.CDLL(PROBE_SO)
    lib.pk_probe2_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    lib.pk_probe2_launch.restype = ctypes.c_int
    return lib


def _forms(mask16):
    return [f"op_sel:[{k >> 3 & 1},{k >> 2 & 1}] op_sel_hi:[{k >> 1 & 1},{k & 1}]" for k in range(16) if mask16 >> k & 1]


def raw_probe_beside(lib, launch_companion, device="cuda", reps: int = 3, probe_launches: int = 4, iters: int = 100) -> dict:
    """``launch_companion()`` enqueues the companion work on the CURRENT stream (called inside a side stream's context); the probe
    runs on a second stream at the same time.  -> {"wrong_results": loop trips with a wrong packed result, "lanes": [lo, hi],
    "lane_quarters": {...}, "v_pk_add_f32": [forms], "v_pk_mul_f32": [forms]}."""
    dev = torch.device(device)
    nt = 1024 * 1024
    log = torch.zeros(8 + 8 * MAXLOG, dtype=torch.int32, device=dev)
    sink = torch.empty(nt, dtype=torch.float32, device=dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    for _ in range(reps):
        if launch_companion is not None:
            with torch.cuda.stream(sb):
                launch_companion()
        with torch.cuda.stream(sa):
            for _ in range(probe_launches):
                rc = lib.pk_probe2_launch(log.data_ptr(), MAXLOG, iters, sink.data_ptr(), nt, sa.cuda_stream)
                if rc != 0:
                    raise RuntimeError(f"pk_probe2_launch: hipError {rc}")
        torch.cuda.synchronize(dev)
    L = log.cpu().numpy().view(np.uint32)
    cnt = int(L[0])
    out = {"wrong_results": cnt, "lanes": None, "lane_quarters": None, "v_pk_add_f32": [], "v_pk_mul_f32": []}
    if cnt:
        E = L[8:8 + 8 * min(cnt, MAXLOG)].reshape(-1, 8)
        lanes = sorted(set(E[:, 2].tolist()))
        m = int(np.bitwise_or.reduce(E[:, 3]))
        out.update(lanes=[int(lanes[0]), int(lanes[-1])], lane_quarters={int(k): int(v) for k, v in collections.Counter((E[:, 2] // 16).tolist()).items()},
                   v_pk_add_f32=_forms(m & 0xffff), v_pk_mul_f32=_forms(m >> 16))
    return out


def product_gemm_companion(device="cuda", launches: int = 4):
    """-> a ``launch_companion`` that runs the product's three-product implicit-GEMM convolution (the head's 3x3 shape at 64
    ROIs: v_mfma_f32_32x32x16_f16 on every SIMD) through hip_layers, as the steps do."""
    from gdrnpp_bop2022_amd import hip_lib
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    torch.manual_seed(3)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(device)
    x = torch.randn(64, 256, 32, 32, device=device).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        n0 = hip_lib.x3_launch_count()
        hip_layers.conv2d(conv, x)
        assert hip_lib.x3_launch_count() > n0, "the companion must be the three-product (f16 MFMA) GEMM"
    torch.cuda.synchronize()

    def launch():
        with torch.no_grad():
            for _ in range(launches):
                hip_layers.conv2d(conv, x)
    return launch


if __name__ == "__main__":
    import json
    import sys

    sys.path.insert(0, os.path.dirname(HERE))
    lib_ = load_probe()
    assert lib_ is not None, "probe library could not be built"
    print(json.dumps({"alone": raw_probe_beside(lib_, None), "beside_product_gemm": raw_probe_beside(lib_, product_gemm_companion())}))
