"""gdrnpp_roi_align on the ops microbenchmark's workload: 128 ROIs of a [16,3,480,640] batch -> 3 x 256 x 256 and 3 x 64 x 64, adaptive
sampling grid.  (The launch-shape variants of profiles/r06_roi_align.md were A/B builds of round 6 — rows per thread, store kind, LDS
staging: tools/probe/roi_align_lds_experiment.hip.txt — the library keeps the one that won.)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdrnpp_bop2022_amd import hip_lib, synthetic as S
hip_lib.load()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(20220925)
b = 128
verts, faces, ext = S.make_models(21, np.random.default_rng(1), 2)
det = S.make_detections(b, 21, ext, rng)
x = torch.rand(16, 3, 480, 640, device=dev)
rois = torch.from_numpy(np.concatenate([rng.integers(0, 16, (b, 1)), det["roi_center"] - det["scale"][:, None] / 2, det["roi_center"] + det["scale"][:, None] / 2], 1).astype(np.float32)).to(dev)
def gpu_time(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
ref = None
names = {0: "4 rows per thread, non-temporal stores (product)"}
for out_res in (256, 64):
    for v in range(1):
        y = hip_lib.roi_align(x, rois, out_res)
        torch.cuda.synchronize()
        if v == 0: ref = y.clone()
        t = gpu_time(lambda: hip_lib.roi_align(x, rois, out_res))
        nb = b * 3 * out_res * out_res * 4
        print(json.dumps({"out": out_res, "variant": v, "shape": names[v], "us": t * 1e6, "GBs": nb / t / 1e9, "frac_of_8TBs": nb / t / 8e12, "bit_equal": bool(torch.equal(y, ref))}), flush=True)
