"""ORACLE — test infrastructure: the ``cpu_baseline`` leg of bench.py.

Run as a separate process (``python -m oracle.cpu_baseline --inputs x.npz --seconds S``) so that the all-core legs can
``fork`` workers without a HIP runtime in the parent.  Times, on the GPU box's host cores and on a BOUNDED sample of the
very batch the GPU ran:

  refine        reference per-ROI refine loop (gdrn_evaluator.py:485-561) restated — oracle "port": NumPy + the C software
                rasteriser in place of the vispy GL render — 1 thread AND all cores (multiprocessing over ROIs)
  upnp_pn9 / upnp_pn4096   uncertainty-PnP (uncertainty_pnp.cpp:7-92) restated with the Ceres LM schedule, 1 thread and
                all cores
  decode        get_out_mask + get_img_model_points_with_coords2d (engine_utils.py:315-333, gdrn_evaluator.py:115-153), 1 thread
  fps / nnd / flow   the reference's OWN compiled sources (oracle/_ref/*.so: farthest_point_sampling.cpp, nnd_cpu.cpp,
                flow_cpu.cpp), 1 thread — kind "reference"; skipped with a note when oracle/_ref was not built
  warpAffine / solvePnP[Ransac]   OpenCV is not installed: "cv2 unavailable" is recorded instead of a substitute number
  forward_cpu_torch   (``--forward-cfg NAME``) the network forward of the workload's config with PyTorch's CPU operators, all
                host cores, on a bounded number of ROIs — the reference runs this stage on the GPU too; this is the like-for-like
                CPU figure of the stage that is 99.8 % of the step (ResNet-34 / 32 ROIs for configs[0])
  refine_parity_sample   (``--parity-sample N``) the refined translation of N evenly spaced ROIs of the batch from
                ``depth_refine_roi``: bench.py compares the GPU records of the same ROIs with it inside the driver run

Prints one JSON object on stdout.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import postproc as P  # noqa: E402
from oracle import ref_lib  # noqa: E402

_G = {}


def _refine_one(i):
    g = _G
    o = int(g["roi_cls"][i])
    P.depth_refine_roi(g["xyz"][i], g["mask"][i, 0], g["roi_depth"][i, 0], g["K_crop"][i], g["rot"][i], g["trans"][i],
                       g["verts"][o], g["faces"][o], iters=g["iters"], threshold=g["thr"])
    return 1


def _upnp_one(i):
    g = _G["upnp"]
    P.uncertainty_pnp(g["p2"][i], g["p3"], g["w"][i], g["K"], g["init"][i])
    return 1


def _loop(fn, n_items, seconds, min_items):
    """Round-robin over the sample until ``seconds`` of wall time and at least ``min_items`` calls."""
    done, t0 = 0, time.perf_counter()
    while True:
        fn(done % n_items)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and done >= min_items:
            return done, dt


def _pool_loop(fn, n_items, seconds, cores):
    """All-core leg: a fork pool maps the sample round-robin in chunks until ``seconds`` have passed."""
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(fn, range(min(n_items, cores)))             # start-up (fork, page-in) outside the clock
        done, t0 = 0, time.perf_counter()
        chunk = max(cores * 64, n_items)                     # big chunks: the IPC per task must not dominate sub-ms tasks
        while True:
            pool.map(fn, [k % n_items for k in range(done, done + chunk)], chunksize=max(1, chunk // (cores * 2)))
            done += chunk
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return done, dt


def make_upnp_problems(n, pn, rng):
    """pn model points, projections + N(0, 1 px), cov^-1/2 weights, init = GT perturbed by U(0, 0.1)."""
    K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], np.float64)
    p3 = rng.uniform(-0.05, 0.05, (pn, 3))
    p2, w, init = [], [], []
    for _ in range(n):
        rt = np.concatenate([rng.uniform(-1, 1, 3), [rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.5, 1.2)]])
        R = P.rodrigues_exp(rt[:3])
        cam = p3 @ R.T + rt[3:]
        uv = cam[:, :2] / cam[:, 2:] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
        p2.append(uv + rng.normal(0, 1.0, uv.shape))
        w.append(np.stack([rng.uniform(0.5, 1.5, pn), rng.uniform(-0.2, 0.2, pn), rng.uniform(0.5, 1.5, pn)], 1))
        init.append(rt + rng.uniform(0, 0.1, 6))
    return dict(p2=np.array(p2), p3=p3, w=np.array(w), K=K, init=np.array(init))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", required=True, help="npz written by bench.py: maps, poses, depth crops, meshes of the batch")
    ap.add_argument("--seconds", type=float, default=20.0, help="total CPU wall budget over all stages")
    ap.add_argument("--cores", type=int, default=0, help="workers of the all-core legs (0 = os.cpu_count())")
    ap.add_argument("--parity-sample", type=int, default=0, help="also return depth_refine_roi's t for this many ROIs of the batch")
    ap.add_argument("--forward-cfg", default="", help="named config whose network forward is timed with PyTorch CPU operators")
    ap.add_argument("--forward-seconds", type=float, default=6.0)
    ap.add_argument("--upnp-only", action="store_true", help="configs[0] leg: uncertainty-PnP pn = 9 (+ the forward), no refine inputs")
    args = ap.parse_args()
    if args.upnp_only:
        return main_upnp(args)
    cores = args.cores or os.cpu_count() or 1
    z = np.load(args.inputs, allow_pickle=False)
    n = int(z["mask"].shape[0])
    _G.update(roi_cls=z["roi_cls"], mask=P.get_out_mask(z["mask"]), roi_depth=z["roi_depth"], K_crop=z["K_crop"], rot=z["rot"],
              trans=z["trans"], iters=int(z["iters"]), thr=float(z["thr"]),
              xyz=[np.concatenate([z["coor_x"][i], z["coor_y"][i], z["coor_z"][i]], 0).transpose(1, 2, 0) for i in range(n)],
              verts=[z["verts"][o][: z["n_verts"][o]] for o in range(len(z["n_verts"]))],
              faces=[z["faces"][o][: z["n_faces"][o]] for o in range(len(z["n_faces"]))])
    share = args.seconds / 8.0
    stages = {}

    done, dt = _loop(_refine_one, n, 2 * share, min(n, 32))
    stages["refine_1thread"] = dict(value=done / dt, unit="ROIs/s", cores=1, kind="port",
                                    sample=f"{done} ROI refinements over {n} distinct ROIs of the batch, {dt:.2f} s")
    done, dt = _pool_loop(_refine_one, n, 2 * share, cores)
    stages["refine_allcores"] = dict(value=done / dt, unit="ROIs/s", cores=cores, kind="port",
                                     sample=f"{done} ROI refinements, fork pool of {cores} workers over ROIs, {dt:.2f} s")

    rng = np.random.default_rng(20220925 + 1)
    for pn, nprob in ((9, 64), (4096, 8)):
        _G["upnp"] = make_upnp_problems(nprob, pn, rng)
        done, dt = _loop(_upnp_one, nprob, share / 2, min(nprob, 4))
        stages[f"upnp_pn{pn}_1thread"] = dict(value=done / dt, unit="problems/s", cores=1, kind="port",
                                              sample=f"{done} LM solves over {nprob} distinct problems, {dt:.2f} s")
        done, dt = _pool_loop(_upnp_one, nprob, share / 2, cores)
        stages[f"upnp_pn{pn}_allcores"] = dict(value=done / dt, unit="problems/s", cores=cores, kind="port",
                                               sample=f"{done} LM solves, fork pool of {cores} workers, {dt:.2f} s")

    if "coord2d" in z.files:
        def dec(i):
            m = _G["mask"][i, 0]
            P.get_img_model_points_with_coords2d(m, _G["xyz"][i].copy(), z["coord2d"][i].transpose(1, 2, 0), 480, 640,
                                                 z["extent"][i])
        done, dt = _loop(dec, n, share / 2, min(n, 32))
        stages["decode_correspondences_1thread"] = dict(value=done / dt, unit="ROIs/s", cores=1, kind="port",
                                                        sample=f"{done} ROIs, {dt:.2f} s")

    # the reference's own compiled sources (kind "reference"), one thread like test_gdrn.sh's OMP_NUM_THREADS=1
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    fps = ref_lib("fps")
    if fps is not None:
        pts = np.ascontiguousarray(_G["verts"][0], np.float32)
        idx = np.zeros(64, np.int32)

        def run(_):
            fps.farthest_point_sampling_init_center(pts.ctypes.data_as(f32p), idx.ctypes.data_as(i32p), len(pts), 64)
        done, dt = _loop(run, 1, share / 3, 3)
        stages["fps_reference_1thread"] = dict(value=done / dt, unit="clouds/s", cores=1, kind="reference",
                                               sample=f"farthest_point_sampling_init_center, {len(pts)} points -> 64, {done} calls, {dt:.2f} s")
    else:
        stages["fps_reference_1thread"] = dict(value=None, note="oracle/_ref/libfps_ref.so not built (needs /root/reference)")
    nnd = ref_lib("nnd")
    if nnd is not None:
        bb, nn, mm = 2, 1000, 1500
        x1 = rng.uniform(0, 1, (bb, nn, 3)).astype(np.float32)
        x2 = rng.uniform(0, 1, (bb, mm, 3)).astype(np.float32)
        d1, d2 = np.zeros((bb, nn), np.float32), np.zeros((bb, mm), np.float32)
        i1, i2 = np.zeros((bb, nn), np.int32), np.zeros((bb, mm), np.int32)

        def run_nnd(_):
            nnd.ref_nnd_forward(x1.ctypes.data_as(f32p), x2.ctypes.data_as(f32p), d1.ctypes.data_as(f32p), d2.ctypes.data_as(f32p),
                                i1.ctypes.data_as(i32p), i2.ctypes.data_as(i32p), bb, nn, mm)
        done, dt = _loop(run_nnd, 1, share / 3, 2)
        stages["nnd_reference_1thread"] = dict(value=done * 2 * bb * nn * mm / dt / 1e9, unit="Gpairs/s", cores=1, kind="reference",
                                               sample=f"nnd_cpu.cpp forward, b={bb}, n={nn}, m={mm}, {done} calls, {dt:.2f} s")
    else:
        stages["nnd_reference_1thread"] = dict(value=None, note="oracle/_ref/libnnd_ref.so not built (needs /root/reference)")
    flow = ref_lib("flow")
    if flow is not None:
        h, w = 480, 640
        ds = rng.uniform(0.5, 1.5, (h, w)).astype(np.float32)
        KT = np.ascontiguousarray(np.hstack([np.eye(3), np.zeros((3, 1))]) * np.array([[572.4], [573.6], [1.0]]), np.float32)
        Kinv = np.ascontiguousarray(np.linalg.inv(np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1]])), np.float32)
        fl, va = np.zeros((2, h, w), np.float32), np.zeros((1, h, w), np.float32)

        def run_flow(_):
            flow.ref_flow_forward(ds.ctypes.data_as(f32p), ds.ctypes.data_as(f32p), KT.ctypes.data_as(f32p), Kinv.ctypes.data_as(f32p),
                                  fl.ctypes.data_as(f32p), va.ctypes.data_as(f32p), 1, h, w)
        done, dt = _loop(run_flow, 1, share / 3, 2)
        stages["flow_reference_1thread"] = dict(value=done * h * w / dt / 1e6, unit="Mpixels/s", cores=1, kind="reference",
                                                sample=f"flow_cpu.cpp, one 480x640 image per call, {done} calls, {dt:.2f} s")
    else:
        stages["flow_reference_1thread"] = dict(value=None, note="oracle/_ref/libflow_ref.so not built (needs /root/reference)")
    for name in ("cv2.warpAffine (ROI crops, data_loader.py:773-797)", "cv2.solvePnPRansac / solvePnP (lib/pysixd/misc.py:153-208)"):
        stages[name] = dict(value=None, note="cv2 unavailable")
    # ... but the ROI preparation stage is the CPU stage that caps an 8-GPU node, so its PORT is timed: the
    # oracle's scalar-C restatement of the three warpAffine calls of read_data_test (u8 bilinear 256^2 x 3, f32 nearest 256^2,
    # f32 bilinear 64^2 x 2) + the fp64 normalisation, per ROI of a 480 x 640 image.  NOT OpenCV's SIMD code: a floor for what a
    # host core does to one ROI with this algorithm, beside the GPU crop kernel's ~0.45 us per ROI (profiles/*ops_microbench.json).
    crng = np.random.default_rng(20220925 + 7)
    c_img = crng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    c_dep = crng.uniform(0.3, 1.5, (480, 640)).astype(np.float32)
    c_ctr = np.stack([crng.uniform(80, 560, 64), crng.uniform(80, 400, 64)], 1)
    c_scl = crng.uniform(60, 300, 64)
    done, dt = _loop(lambda i: P.crop_resize_roi(c_img, c_dep, c_ctr[i], float(c_scl[i])), 64, share / 2, 4)
    stages["crop_resize_port_1thread"] = dict(value=done / dt, unit="ROIs/s", cores=1, kind="port",
                                              sample=f"{done} ROI crops (3 warps + normalisation each) of one 480x640 image, oracle/warp_oracle.c, {dt:.2f} s",
                                              note="scalar restatement of cv2.warpAffine's algorithm, not OpenCV itself (absent)")

    # the same for the PnP stage of the TEST.USE_PNP branches: solvePnPRansac(EPNP, reprojErr 3, 100 iterations) on the 2D-3D
    # correspondences of a ROI (misc.pnp_v2) — the oracle's NumPy restatement (oracle/epnp.py), not OpenCV
    try:
        from oracle import epnp as EP
        prng = np.random.default_rng(20220925 + 9)
        K_p = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1.0]])
        probs = []
        for _ in range(8):
            pw = prng.uniform(-0.08, 0.08, (600, 3))
            Rq, _ = np.linalg.qr(prng.standard_normal((3, 3)))
            Rq = Rq * np.sign(np.linalg.det(Rq))
            tq = np.array([prng.uniform(-0.1, 0.1), prng.uniform(-0.1, 0.1), prng.uniform(0.5, 1.2)])
            cam_p = pw @ Rq.T + tq
            uv = cam_p[:, :2] / cam_p[:, 2:] * np.array([K_p[0, 0], K_p[1, 1]]) + np.array([K_p[0, 2], K_p[1, 2]]) + prng.normal(0, 0.7, (600, 2))
            uv[:60] += prng.uniform(-40, 40, (60, 2))                      # 10 % outliers
            probs.append((pw, uv))
        done, dt = _loop(lambda i: EP.solve_pnp_ransac_epnp(probs[i][0], probs[i][1], K_p, 3.0, 100), 8, share / 2, 2)
        stages["pnp_ransac_epnp_port_1thread"] = dict(value=done / dt, unit="ROIs/s", cores=1, kind="port",
                                                      sample=f"{done} RANSAC-EPnP solves (600 correspondences, 10 % outliers, 100 iterations max), oracle/epnp.py, {dt:.2f} s",
                                                      note="NumPy restatement of cv2.solvePnPRansac(EPNP), not OpenCV itself (absent)")
    except Exception as e:  # a baseline stage must not take the line down
        stages["pnp_ransac_epnp_port_1thread"] = dict(value=None, note=repr(e))

    if args.forward_cfg:
        stages["forward_cpu_torch"] = forward_cpu_torch(args.forward_cfg, args.forward_seconds)
    top = dict(stages["refine_1thread"])
    top.update(stages=stages, host_cores_available=os.cpu_count(),
               allcores=dict(value=stages["refine_allcores"]["value"], cores=cores))
    if args.parity_sample > 0:
        idx = np.unique(np.linspace(0, n - 1, min(args.parity_sample, n)).astype(int))
        g = _G
        ts = [np.asarray(P.depth_refine_roi(g["xyz"][i], g["mask"][i, 0], g["roi_depth"][i, 0], g["K_crop"][i], g["rot"][i], g["trans"][i],
                                            g["verts"][int(g["roi_cls"][i])], g["faces"][int(g["roi_cls"][i])], iters=g["iters"],
                                            threshold=g["thr"]), np.float64).tolist() for i in idx]
        top["refine_parity_sample"] = dict(idx=idx.tolist(), t=ts, oracle="oracle.postproc.depth_refine_roi (gdrn_evaluator.py:485-561 restated, "
                                           "pinned by the reference's process_depth_refine run from source)")
    print(json.dumps(top))


def forward_cpu_torch(cfg_name, seconds, rois=4):
    """The network forward with PyTorch's CPU operators (this repo's module graph with its HIP layers switched off — the graph
    tests/test_net_golden.py pins against the reference's own modules), all host cores, ``rois`` ROIs per call."""
    import torch

    from gdrnpp_bop2022_amd import synthetic as S
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg

    hip_layers.set_enabled(False)
    torch.set_grad_enabled(False)
    # PyTorch's CPU operators stop scaling (and collapse: 256 threads ran one 4-ROI forward in 53 s on the GPU box's host) long
    # before a 256-core host is full: 32 threads is where oneDNN's convolutions / sgemm of these sizes still scale
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = get_cfg(cfg_name, ["MODEL.DEVICE=cpu"])
    if "resnet" in cfg_name:
        from gdrnpp_bop2022_amd.gdrn_modeling import GDRN as G
    else:
        from gdrnpp_bop2022_amd.gdrn_modeling import GDRN_double_mask as G
    model, _ = G.build_model_optimizer(cfg, is_test=True)
    model = model.to("cpu").eval()
    C = cfg.MODEL.POSE_NET.NUM_CLASSES
    rng = np.random.default_rng(0)
    ext = rng.uniform(0.05, 0.25, (C, 3)).astype(np.float32)
    det = S.make_detections(rois, C, ext, rng)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    x = torch.rand(rois, 3, 256, 256)
    cls, c2d, extent = T(det["roi_cls"]), T(S.coord2d_roi(det["roi_center"], det["scale"])), T(det["roi_extent"])
    # forward_maps = backbone + geometry head + Patch-PnP (everything but the closing 6-D -> R / centroid -> t conversion, which
    # the product does in a HIP kernel and which is a few hundred flops per ROI)
    model.forward_maps(x, cls, c2d, None, extent)            # oneDNN primitive creation outside the clock
    done, dt = _loop(lambda _: model.forward_maps(x, cls, c2d, None, extent), 1, seconds, 1)
    return dict(value=done * rois / dt, unit="ROIs/s", cores=cores, kind="port",
                sample=f"{done} forwards of {rois} ROIs ({cfg_name}, fp32, PyTorch CPU operators, {cores} threads), {dt:.2f} s",
                note="the reference runs this stage on the GPU as well; CPU figure for scale only")


def main_upnp(args):
    """BASELINE configs[0] (LM-O ape, 32 ROIs, ResNet-34 forward + uncertainty-PnP, pn = 9): both stages on the host cores."""
    cores = args.cores or os.cpu_count() or 1
    rng = np.random.default_rng(20220925 + 1)
    stages = {}
    _G["upnp"] = make_upnp_problems(64, 9, rng)
    done, dt = _loop(_upnp_one, 64, args.seconds / 3, 8)
    stages["upnp_pn9_1thread"] = dict(value=done / dt, unit="ROIs/s", cores=1, kind="port",
                                      sample=f"{done} LM solves (pn = 9) over 64 distinct problems, {dt:.2f} s")
    done, dt = _pool_loop(_upnp_one, 64, args.seconds / 3, cores)
    stages["upnp_pn9_allcores"] = dict(value=done / dt, unit="ROIs/s", cores=cores, kind="port",
                                       sample=f"{done} LM solves, fork pool of {cores} workers, {dt:.2f} s")
    if args.forward_cfg:
        stages["forward_cpu_torch"] = forward_cpu_torch(args.forward_cfg, args.seconds / 3, rois=32)
        f, u = stages["forward_cpu_torch"]["value"], stages["upnp_pn9_allcores"]["value"]
        top = dict(value=1.0 / (1.0 / f + 1.0 / u), unit="ROIs/s", cores=cores, kind="port",
                   sample=f"configs[0] end to end on the host: {stages['forward_cpu_torch']['sample']} + {stages['upnp_pn9_allcores']['sample']}",
                   note="value = 1 / (1 / forward + 1 / uncertainty-PnP), both stages on all host cores one after the other")
    else:
        top = dict(stages["upnp_pn9_1thread"])
    top.update(stages=stages, host_cores_available=os.cpu_count())
    print(json.dumps(top))


if __name__ == "__main__":
    main()
