"""gdrnpp_refine_to_records at the reference's own batch sizes (one image = 3-30 ROIs): one workgroup per ROI vs 2 / 4 workgroups per
ROI (gdrnpp_refine_to_records_split), us per launch, bit-equality of the records.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gdrnpp_bop2022_amd import hip_lib, synthetic as S

dev = "cuda"
rng = np.random.default_rng(0)
verts, faces, ext = S.make_models(21, rng, 4)
meshes = hip_lib.MeshSet(verts, faces)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def render_fn(obj, K, R, t, res):
    d, x = hip_lib.render_depth(meshes, T(obj), T(K), T(R), T(t), res, want_xyz=True)
    return d.cpu().numpy(), x.cpu().numpy()


def us(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()          # launch-to-launch time without the Python wrapper: only meaningful for split = 1 (capture)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("| ROIs | split 1 (us) | split 2 (us) | split 4 (us) | library's choice | records bit-equal |")
print("|---|---|---|---|---|---|")
for b in (4, 8, 16, 32, 64, 128):
    det = S.make_detections(b, 21, ext, rng)
    maps = S.make_map_inputs(det, verts, faces, render_fn, rng)
    obj = T(det["roi_cls"].astype(np.int32))
    args = (T(maps["coor_x"]), T(maps["coor_y"]), T(maps["coor_z"]), T(maps["mask"]), T(maps["roi_depth"]), T(det["roi_cam"]).reshape(b, 9),
            T(det["roi_center"]), T(det["scale"]), T(det["R_gt"]).reshape(b, 9), T(maps["t_init"]), T(det["score"]), None)
    auto = hip_lib.load().gdrnpp_refine_split_factor(meshes.c, b)
    want = hip_lib.refine_to_records(meshes, obj, *args, split=1)
    row, same = [], True
    for split in (1, 2, 4):
        if b * split > 256:
            row.append("-")
            continue
        same &= bool(torch.equal(hip_lib.refine_to_records(meshes, obj, *args, split=split), want))
        row.append(f"{us(lambda: hip_lib.refine_to_records(meshes, obj, *args, split=split)):.1f}")
    print(f"| {b} | {row[0]} | {row[1]} | {row[2]} | {auto} | {same} |")
print("status word:", hip_lib.refine_split_status())
