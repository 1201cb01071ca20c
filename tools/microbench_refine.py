"""Micro-benchmark of gdrnpp_depth_refine: prologue vs per-iteration cost, batch scaling (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gdrnpp_bop2022_amd import hip_lib, synthetic as S

dev = "cuda"
rng = np.random.default_rng(0)
verts, faces, ext = S.make_models(21, rng, int(os.environ.get("SUBDIV", "4")))
meshes = hip_lib.MeshSet(verts, faces)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

def render_fn(obj, K, R, t, res):
    d, x = hip_lib.render_depth(meshes, T(obj), T(K), T(R), T(t), res, want_xyz=True)
    return d.cpu().numpy(), x.cpu().numpy()

for b in (128, 256, 1024):
    det = S.make_detections(b, 21, ext, rng)
    maps = S.make_map_inputs(det, verts, faces, render_fn, rng)
    args = (meshes, T(det["roi_cls"].astype(np.int32)), T(maps["coor_x"]), T(maps["coor_y"]), T(maps["coor_z"]),
            T(maps["mask"]), T(maps["roi_depth"]), T(maps["K_crop"]), T(det["R_gt"]), T(maps["t_init"]))
    out = torch.empty((b, 3), dtype=torch.float64, device=dev)
    for iters in (0, 1, 2):
        for _ in range(5):
            hip_lib.depth_refine(*args, iters=iters, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            hip_lib.depth_refine(*args, iters=iters, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"b={b} iters={iters}: {us:.1f} us/launch  ({us / b:.2f} us/ROI)")

import ctypes
buf = (ctypes.c_longlong * 16)()
hip_lib.depth_refine(*args, iters=2, out=out); torch.cuda.synchronize()
hip_lib.load().gdrnpp_debug_refine_profile(ctypes.cast(buf, ctypes.c_void_p))
st = list(buf)
names = ["prologue", "stage0", "raster0", "reduce0", "median0", "update0", "stage1", "raster1", "reduce1", "median1", "update1"]
print("workgroup-0 phase cycles (s_memtime @100MHz? raw units):", {n: st[i + 1] - st[i] for i, n in enumerate(names)})
