"""Workload for the PMC passes that fill roofline.traffic (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
  1. calibration: known-size streaming reads with 4-byte and 16-byte lanes over a buffer larger than the
     256 MiB Infinity Cache (MI355X_MICROARCH.md §HBM: FETCH_SIZE halves wide streams on gfx950; other
     widths must be calibrated),
  2. the refine kernel at the bench workload (128 ROIs, 2562/5120 meshes), maps produced by a real forward.
Parse with tools/pmc_parse.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gdrnpp_bop2022_amd import hip_lib, synthetic as S

dev = "cuda"
lib = hip_lib.load()
n = 512 * 1024 * 1024 // 4  # 512 MiB of floats
buf = torch.rand(n, device=dev)
out = torch.zeros(4096, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.gdrnpp_debug_stream_read(buf.data_ptr(), n, 4, out.data_ptr(), 4096, st)
    lib.gdrnpp_debug_stream_read(buf.data_ptr(), n, 16, out.data_ptr(), 4096, st)
torch.cuda.synchronize()
del buf

rng = np.random.default_rng(0)
b = 128
verts, faces, ext = S.make_models(21, np.random.default_rng(20220925), 4)
meshes = hip_lib.MeshSet(verts, faces)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
det = S.make_detections(b, 21, ext, rng)

def render_fn(obj, K, R, t, res):
    d, x = hip_lib.render_depth(meshes, T(obj), T(K), T(R), T(t), res, want_xyz=True)
    return d.cpu().numpy(), x.cpu().numpy()

maps = S.make_map_inputs(det, verts, faces, render_fn, rng)
args = (meshes, T(det["roi_cls"].astype(np.int32)), T(maps["coor_x"]), T(maps["coor_y"]), T(maps["coor_z"]),
        T(maps["mask"]), T(maps["roi_depth"]), T(maps["K_crop"]), T(det["R_gt"]), T(maps["t_init"]))
flush = torch.empty(600 * 1024 * 1024 // 4, device=dev)
for _ in range(5):
    flush.sum()               # read-only sweep: evicts the inputs from L2 / Infinity Cache without leaving dirty lines
    hip_lib.depth_refine(*args)
torch.cuda.synchronize()
print("done")
