"""Per-op micro-benchmark of the hand-written kernels at the sizes of SURVEY.md §8(d), each with its algorithmic
bytes / pair counts and the CPU oracle ("port", 1 thread) timed beside it on a bounded sample.
Run on the GPU box:  python tools/microbench_ops.py > gpurun_out/ops.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gdrnpp_bop2022_amd import hip_lib, synthetic as S
from oracle import postproc as P

dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rng = np.random.default_rng(0)
res = {}


def gpu_time(fn, n=30, warm=5):
    """Seconds per call.  The calls are captured into one HIP graph and replayed, so that launches of a few microseconds are
    not measured at the rate the Python wrapper can issue them; ops whose wrapper synchronises with the host (capture fails)
    fall back to a plain event-timed loop."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        gpu_time.mode = "graph"
        return e0.elapsed_time(e1) / n * 1e-3
    except Exception:
        torch.cuda.synchronize()
    gpu_time.mode = "loop"
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def cpu_time(fn, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


# ---- FPS: 21 model clouds (2562 pts) x 8+... and one big cloud
pts = (rng.standard_normal((21, 2562, 3)) * 0.05).astype(np.float32)
d = T(pts)
t = gpu_time(lambda: hip_lib.fps(d, 64, True))
tc = cpu_time(lambda: P.fps(pts[0], 64, True), 3)
res["fps_21x2562_sn64"] = dict(gpu_s=t, bytes=21 * 2562 * 12 + 21 * 64 * 4, dist_evals=21 * 2562 * 64,
                               gpu_Gdist_s=21 * 2562 * 64 / t / 1e9, cpu_port_s_per_cloud=tc, cpu_cores=1,
                               gpu_clouds_per_s=21 / t, cpu_clouds_per_s=1 / tc)
big = (rng.standard_normal((1, 100000, 3)) * 0.05).astype(np.float32)
db = T(big)
t = gpu_time(lambda: hip_lib.fps(db, 256, True), n=5, warm=1)
tc = cpu_time(lambda: P.fps(big[0], 256, True))
res["fps_1x100000_sn256"] = dict(gpu_s=t, dist_evals=100000 * 256, gpu_Gdist_s=100000 * 256 / t / 1e9,
                                 cpu_port_s=tc, cpu_cores=1)

# ---- NN distance: reference smoke size b=10, n=1000, m=1500 and a larger one
for (b, n, m) in [(10, 1000, 1500), (32, 4096, 4096)]:
    x1 = rng.uniform(0, 1, (b, n, 3)).astype(np.float32); x2 = rng.uniform(0, 1, (b, m, 3)).astype(np.float32)
    a1, a2 = T(x1), T(x2)
    d1 = torch.zeros(b, n, device=dev); d2 = torch.zeros(b, m, device=dev)
    i1 = torch.zeros(b, n, dtype=torch.int32, device=dev); i2 = torch.zeros(b, m, dtype=torch.int32, device=dev)
    t = gpu_time(lambda: hip_lib.nnd_forward(a1, a2, d1, d2, i1, i2))
    tc = cpu_time(lambda: P.nnd_forward(x1[:2], x2[:2])) * b / 2
    pairs = 2 * b * n * m
    res[f"nnd_b{b}_n{n}_m{m}"] = dict(gpu_s=t, pairs=pairs, gpu_Gpairs_s=pairs / t / 1e9,
                                     bytes=12 * (n + m) * b + 8 * (n + m) * b, cpu_port_s=tc, cpu_cores=1,
                                     cpu_Gpairs_s=pairs / tc / 1e9)

# ---- RANSAC voting: tn=4096 (full 64x64 mask), vn=9, hn=128: fused count vs flags
tn, vn, hn = 4096, 9, 128
coords = np.stack(np.meshgrid(np.arange(64), np.arange(64)), -1).reshape(-1, 2).astype(np.float32)
direct = rng.standard_normal((tn, vn, 2)).astype(np.float32)
direct /= np.linalg.norm(direct, axis=-1, keepdims=True)
idxs = rng.integers(0, tn, (hn, vn, 2)).astype(np.int32)
dd, dc, di = T(direct), T(coords), T(idxs)
from gdrnpp_bop2022_amd.core.csrc.ransac_voting import ransac_voting as rv
hyp = rv.generate_hypothesis(dd, dc, di)
t_gen = gpu_time(lambda: rv.generate_hypothesis(dd, dc, di))
t_cnt = gpu_time(lambda: rv.vote_count(dd, dc, hyp, 0.999))
inl = torch.zeros((hn, vn, tn), dtype=torch.uint8, device=dev)
t_flag = gpu_time(lambda: rv.voting_for_hypothesis(dd, dc, hyp, inl, 0.999))
tc = cpu_time(lambda: P.voting_for_hypothesis(direct, coords, hyp.cpu().numpy(), 0.999))
pairs = hn * vn * tn
res["ransac_voting_tn4096_vn9_hn128"] = dict(
    generate_s=t_gen, vote_count_s=t_cnt, voting_flags_s=t_flag, pairs=pairs, vote_count_Gpairs_s=pairs / t_cnt / 1e9,
    bytes_count=tn * vn * 8 + tn * 8 + hn * vn * 8 + hn * vn * 4, bytes_flags=tn * vn * 8 + tn * 8 + hn * vn * 8 + pairs,
    cpu_port_s=tc, cpu_cores=1, cpu_Gpairs_s=pairs / tc / 1e9)

# ---- uncertainty-PnP: 128 problems, pn = 9 and 4096
for pn in (9, 4096):
    b = 128
    K = np.tile(np.array([572.4, 0, 325.3, 0, 573.6, 242.0, 0, 0, 1.0]), (b, 1))
    rt = np.concatenate([rng.uniform(-1, 1, (b, 3)), np.tile([0.05, -0.03, 1.0], (b, 1))], 1)
    p3 = rng.uniform(-0.1, 0.1, (b, pn, 3))
    p2 = np.zeros((b, pn, 2))
    for i in range(b):
        th = np.linalg.norm(rt[i, :3]); k = rt[i, :3] / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        X = p3[i] @ R.T + rt[i, 3:]
        p2[i] = np.stack([572.4 * X[:, 0] / X[:, 2] + 325.3, 573.6 * X[:, 1] / X[:, 2] + 242.0], 1)
    p2 += rng.normal(0, 1, p2.shape)
    w = np.stack([rng.uniform(0.5, 2, (b, pn)), rng.uniform(-0.2, 0.2, (b, pn)), rng.uniform(0.5, 2, (b, pn))], 2)
    init = rt + rng.uniform(0, 0.1, (b, 6))
    a = [T(v) for v in (p2, p3, w, K, init)]
    t = gpu_time(lambda: hip_lib.uncertainty_pnp_batched(*a), n=10, warm=2)
    nb = 8 if pn == 4096 else b
    tc = cpu_time(lambda: P.uncertainty_pnp_batched(p2[:nb], p3[:nb], w[:nb], K[:nb], init[:nb])) / nb
    res[f"upnp_b128_pn{pn}"] = dict(gpu_s=t, gpu_problems_per_s=b / t, bytes=b * (64 * pn + 120 + 48),
                                    cpu_port_s_per_problem=tc, cpu_problems_per_s=1 / tc, cpu_cores=1)

# ---- decode + correspondences, crop-resize, roi_align, render at 128 ROIs
b = 128
verts, faces, ext = S.make_models(21, np.random.default_rng(20220925), 4)
meshes = hip_lib.MeshSet(verts, faces)
det = S.make_detections(b, 21, ext, rng)

def render_fn(obj, K, R, t, r):
    dd_, xx_ = hip_lib.render_depth(meshes, T(obj), T(K), T(R), T(t), r, want_xyz=True)
    return dd_.cpu().numpy(), xx_.cpu().numpy()

maps = S.make_map_inputs(det, verts, faces, render_fn, rng)
m = {k: T(v) for k, v in maps.items()}
extd, imwh = T(det["roi_extent"]), T(np.stack([det["im_W"], det["im_H"]], 1))
t = gpu_time(lambda: hip_lib.decode_correspondences(m["coor_x"], m["coor_y"], m["coor_z"], m["mask"], m["roi_coord_2d"], extd, imwh))
cnt = hip_lib.decode_correspondences(m["coor_x"], m["coor_y"], m["coor_z"], m["mask"], m["roi_coord_2d"], extd, imwh)[0]
nsel = int(cnt.sum())
omask = P.get_out_mask(maps["mask"][:16])
def cpu_dec():
    for i in range(16):
        xyz = np.concatenate([maps["coor_x"][i], maps["coor_y"][i], maps["coor_z"][i]], 0).transpose(1, 2, 0)
        P.get_img_model_points_with_coords2d(omask[i, 0], xyz, maps["roi_coord_2d"][i].transpose(1, 2, 0), 480, 640, det["roi_extent"][i])
tc = cpu_time(cpu_dec) / 16
byts = b * (98304 + 48 + 16384) + nsel * 24 + b * 4
res["decode_correspondences_b128"] = dict(gpu_s=t, bytes=byts, gpu_GBs=byts / t / 1e9, selected=nsel,
                                          cpu_port_s_per_roi=tc, cpu_rois_per_s=1 / tc, gpu_rois_per_s=b / t, cpu_cores=1)

obj, Kc, Rg, tg = T(det["roi_cls"].astype(np.int32)), m["K_crop"], T(det["R_gt"]), T(det["t_gt"])
t = gpu_time(lambda: hip_lib.render_depth(meshes, obj, Kc, Rg, tg, 64))
tc = cpu_time(lambda: [P.render_depth(verts[int(det["roi_cls"][i])], faces[int(det["roi_cls"][i])], maps["K_crop"][i], det["R_gt"][i], det["t_gt"][i].astype(np.float64), 64) for i in range(16)]) / 16
res["render_depth_b128_64x64_5120F"] = dict(gpu_s=t, gpu_renders_per_s=b / t, bytes=b * (12 * 2562 + 12 * 5120 + 4 * 4096),
                                           cpu_port_s_per_render=tc, cpu_renders_per_s=1 / tc, cpu_cores=1)

images = torch.randint(0, 256, (16, 480, 640, 3), dtype=torch.uint8, device=dev)
depths = torch.rand(16, 480, 640, device=dev)
ctr = T(det["roi_center"].astype(np.float64)); scl = T(det["scale"].astype(np.float64))
imi = T(rng.integers(0, 16, b).astype(np.int32))
t = gpu_time(lambda: hip_lib.crop_resize_roi(images, depths, imi, ctr, scl))
img0 = images[0].cpu().numpy(); dep0 = depths[0].cpu().numpy()
tc = cpu_time(lambda: [P.crop_resize_roi(img0, dep0, det["roi_center"][i].astype(np.float64), float(det["scale"][i])) for i in range(4)]) / 4
wbytes = b * (3 * 256 * 256 * 4 + 256 * 256 * 4 + 2 * 64 * 64 * 4)
rbytes = int(sum(7 * float(s) ** 2 for s in det["scale"]))
res["crop_resize_roi_b128"] = dict(gpu_s=t, bytes=wbytes + rbytes, gpu_GBs=(wbytes + rbytes) / t / 1e9, gpu_rois_per_s=b / t,
                                  cpu_port_s_per_roi=tc, cpu_rois_per_s=1 / tc, cpu_cores=1)

x = torch.rand(16, 3, 480, 640, device=dev)
rois = T(np.concatenate([rng.integers(0, 16, (b, 1)), det["roi_center"] - det["scale"][:, None] / 2, det["roi_center"] + det["scale"][:, None] / 2], 1).astype(np.float32))
t = gpu_time(lambda: hip_lib.roi_align(x, rois, 256))
res["roi_align_b128_3x256x256"] = dict(gpu_s=t, bytes=b * 3 * 256 * 256 * 4, gpu_GBs=b * 3 * 256 * 256 * 4 / t / 1e9, gpu_rois_per_s=b / t)

# ---- the reference's OWN compiled CPU sources beside the GPU numbers where they exist (oracle/_ref, kind "reference")
import ctypes
from oracle import ref_lib
f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
_fps = ref_lib("fps")
if _fps is not None:
    p0 = np.ascontiguousarray(pts[0]); idx = np.zeros(64, np.int32)
    tc = cpu_time(lambda: _fps.farthest_point_sampling_init_center(p0.ctypes.data_as(f32p), idx.ctypes.data_as(i32p), len(p0), 64), 20)
    res["fps_21x2562_sn64"].update(cpu_reference_s_per_cloud=tc, cpu_reference="core/csrc/fps/src/farthest_point_sampling.cpp compiled (oracle/_ref), 1 thread")
_nnd = ref_lib("nnd")
if _nnd is not None:
    bb, nn, mm = 2, 1000, 1500
    x1 = rng.uniform(0, 1, (bb, nn, 3)).astype(np.float32); x2 = rng.uniform(0, 1, (bb, mm, 3)).astype(np.float32)
    d1, d2 = np.zeros((bb, nn), np.float32), np.zeros((bb, mm), np.float32)
    i1, i2 = np.zeros((bb, nn), np.int32), np.zeros((bb, mm), np.int32)
    tc = cpu_time(lambda: _nnd.ref_nnd_forward(x1.ctypes.data_as(f32p), x2.ctypes.data_as(f32p), d1.ctypes.data_as(f32p), d2.ctypes.data_as(f32p),
                                               i1.ctypes.data_as(i32p), i2.ctypes.data_as(i32p), bb, nn, mm), 3)
    for k_ in ("nnd_b10_n1000_m1500", "nnd_b32_n4096_m4096"):
        res[k_].update(cpu_reference_Gpairs_s=2 * bb * nn * mm / tc / 1e9, cpu_reference="core/csrc/torch_nndistance/src/nnd_cpu.cpp compiled (oracle/_ref), 1 thread")

# ---- what bounds each op, and how far it is from that bound (MI355X_MICROARCH.md: HBM 8 TB/s, fp32 vector 157.3 TFLOP/s)
HBM, VALU32 = 8000.0, 157.3


def roof(bound, achieved, peak, unit, why):
    return dict(bound=bound, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak, why=why)


def hbm(key, why, t_key="gpu_s", b_key="bytes"):
    e = res[key]
    e["roofline"] = roof("hbm", e[b_key] / e[t_key] / 1e9, HBM, "GB/s", why)


e = res["fps_21x2562_sn64"]
e["roofline"] = roof("latency", e["bytes"] / e["gpu_s"] / 1e9, HBM, "GB/s",
                     "64 DEPENDENT rounds per cloud (distance update + argmax over 2562 points resident in LDS, wave DPP argmax + one barrier); one "
                     "workgroup per cloud = 21 of 256 CUs; %.2f us per round; the cloud is read from HBM once (0.65 MB)" % (e["gpu_s"] / 64 * 1e6))
e = res["fps_1x100000_sn256"]
e["roofline"] = roof("latency", 100000 * 12 * 256 / e["gpu_s"] / 1e9, HBM, "GB/s",
                     "cloud beyond the LDS form (> 12 288 points): every round re-reads it (L2-resident, 1.2 MB) on ONE workgroup; %.1f us per round" % (e["gpu_s"] / 256 * 1e6))
for k_ in ("nnd_b10_n1000_m1500", "nnd_b32_n4096_m4096"):
    e = res[k_]
    e["roofline"] = roof("valu", e["pairs"] * 8 / e["gpu_s"] / 1e12, VALU32, "TFLOP/s",
                         "all-pairs: 8 fp32 operations per pair (3 sub, 3 fma, compare + select with the index), operands from LDS tiles; HBM traffic is "
                         "negligible (%.1f GB/s)" % (e["bytes"] / e["gpu_s"] / 1e9))
e = res["ransac_voting_tn4096_vn9_hn128"]
e["roofline"] = roof("valu", e["pairs"] * 10 / e["vote_count_s"] / 1e12, VALU32, "TFLOP/s",
                     "vote_count: 10 fp32 operations per (hypothesis, point) pair incl. the normalisation + compare, ballot + popcount per wave; "
                     "the flag form writes 1 B per pair instead (%.0f GB/s)" % (e["bytes_flags"] / e["voting_flags_s"] / 1e9))
for k_, why in (("upnp_b128_pn9", "one wave per problem, fp64 Levenberg-Marquardt: ~10 dependent iterations of (residual + Jacobian over 9 points, 6x6 Cholesky), "
                                   "launch + latency bound (128 waves on 256 CUs)"),
                ("upnp_b128_pn4096", "one wave per problem, fp64 LM: per iteration 4096 residual / Jacobian rows per problem (64 per lane) + wave reductions of "
                                     "the 27 normal-equation sums; fp64 vector work, inputs L2-resident after the first iteration")):
    e = res[k_]
    e["roofline"] = roof("latency / fp64 valu", e["bytes"] / e["gpu_s"] / 1e9, HBM, "GB/s", why)
hbm("decode_correspondences_b128", "reads 6 map planes + writes the compacted points once; 6.9 us launch: too short to reach the streaming rate")
e = res["render_depth_b128_64x64_5120F"]
e["roofline"] = roof("latency (LDS atomics)", e["bytes"] / e["gpu_s"] / 1e9, HBM, "GB/s",
                     "one workgroup per ROI: vertex transform into LDS, 5120 triangles rasterised with fp64 edge functions into a ds_min z-buffer; 128 of 256 CUs")
hbm("crop_resize_roi_b128", "writes 1.08 MB per ROI (3x256x256 + 256x256 + 2x64x64 fp32), gathers the source pixels through L2")
hbm("roi_align_b128_3x256x256", "writes 0.79 MB per ROI; 4 bilinear samples x sampling_ratio^2 gathers per output from the NCHW feature map")

res["_peaks"] = dict(hbm_GBs=8000, fp32_vector_TFLOPs=157.3, note="CPU numbers: oracle port, 1 thread, bounded sample, host of the GPU box; GPU numbers: 30 calls replayed from one HIP graph where the wrapper can be captured")
print(json.dumps(res, indent=1))
