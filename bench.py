#!/usr/bin/env python
"""bench.py — ROIs/sec of the GDRNPP hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic ROIs already resident in HBM:
    GDRN_Net forward (ConvNeXt-B + geometry head + Patch-PnP, fp32)  ->  K_crop  ->  fast depth refine
    (render + compare, 2 iterations, HIP)  ->  pose records  ->  (N>1) one RCCL all-gather of the records.

Workload (config.workload): BASELINE.json configs[2] — "YCB-V convnext_a6 + fast depth refine
(render-compare), 1xMI355X, batch=128 ROIs": it is the configuration the metric
"ROIs/sec (GDRNPP fwd + PnP + depth refine)" is quoted on and it fits one GPU.  --workload rgb selects
configs[1] (batch 64, no refine) for reference.  Multi-GPU: ROIs are sharded, every rank gets its own
batch (weak scaling), the only collective is the all-gather of the [n,16] pose records.

Prints ONE JSON line on rank 0 (driver contract) with two extra objects:
  roofline      dominant kernel of the step: gemm_split_kernel (ConvNeXt MLP + head 3x3 convolutions, ~77 % of the
                step), MFMA-bound: achieved = bf16 MFMA flops actually executed (6 partial products per fp32
                product) / summed launch durations, measured with HIP events on the launch stream inside the timed
                region; peak = dense bf16 MFMA.  roofline_other_kernels: depth_refine_staged_kernel (HBM-bound by
                assignment; algorithmic bytes / launch duration, PMC traffic) and the other
                hand-written kernels.
  cpu_baseline  the CPU restatement of the reference's per-ROI refine path (oracle "port": NumPy +
                C software rasteriser standing in for the GL render), one thread, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gdrnpp_bop2022_amd import hip_lib, synthetic as S  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.engine import GdrnHipPost, GraphedInference, gather_records  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3  # same guide: fp32-input MFMA (1/16 of bf16)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="refine", choices=["refine", "rgb", "bop7"],
                   help="refine = BASELINE configs[2] (default); rgb = configs[1]; bop7 = configs[4]-style mixed stream "
                        "(lmo/ycbv/tless/icbin/hb/itodd/tudl models cycled step by step, with depth refine)")
    p.add_argument("--streams", type=int, default=1,
                   help="split each step's batch over this many HIP streams so that memory-bound layers of one part "
                        "overlap MFMA-bound layers of another")
    p.add_argument("--graph", action="store_true", help="replay the whole step from a captured hipGraph (small batches)")
    p.add_argument("--with-crop", action="store_true",
                   help="start each step from full images: GPU ROI crop-resize (row a1) feeds the forward")
    p.add_argument("--batch", type=int, default=0, help="ROIs per GPU per step (0 = the config's batch)")
    p.add_argument("--subdiv", type=int, default=4, help="icosphere subdivision of the synthetic meshes (4 = 2562V/5120F)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=128, help="distinct ROIs of the batch used by the CPU baseline")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work spent on the baseline sample")
    p.add_argument("--exact-reference-order", action="store_true",
                   help="run the reference's full 1470-channel output layer + gather instead of the class-sliced one")
    p.add_argument("--no-hip-layers", action="store_true", help="A/B: run the memory-bound network layers with PyTorch ops")
    p.add_argument("--no-fused-mlp", action="store_true", help="A/B: hipBLASLt GEMMs + separate GELU / addcmul")
    p.add_argument("--library-below-tiles", type=int, default=None,
                   help="ConvNeXt blocks whose fc2 has fewer output tiles than this use hipBLASLt (A/B for small batches)")
    p.add_argument("--no-conv-split", action="store_true", help="A/B: 3x3 head convolutions in MIOpen (fp32 MFMA)")
    p.add_argument("--mlp-gemm", choices=["split", "f32", "torch"], default="split",
                   help="ConvNeXt MLP GEMM engine: split = exact 3-way bf16 split on the bf16 matrix cores (six partial "
                        "products, fp32 accumulate, fp32-accurate); f32 = fp32 MFMA; torch = hipBLASLt + elementwise")
    p.add_argument("--post-only", action="store_true", help="time only the post-processing (maps from a fixed forward)")
    return p.parse_args()


def algorithmic_bytes_refine(b, iters, n_verts, n_faces):
    """SURVEY.md §8(d) row a8, per ROI: (3+1) maps x 64^2 x 4 B + the used quarter of the 256^2 depth crop
    (64^2 x 4 x 4 B) + K,R,t (84 B) + iters x (12 V + 12 F) mesh bytes + 12 B written."""
    per_roi = 65536 + 65536 + 84 + iters * (12 * n_verts + 12 * n_faces) + 12
    return b * per_roi, per_roi


def make_batch(cfg, b, rng, dev, verts, faces, ext, meshes, with_depth):
    C = cfg.MODEL.POSE_NET.NUM_CLASSES
    det = S.make_detections(b, C, ext, rng)

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    batch = dict(
        roi_img=torch.rand(b, 3, 256, 256, device=dev), roi_cls=T(det["roi_cls"]), roi_cam=T(det["roi_cam"]),
        roi_wh=T(det["roi_wh"]), roi_center=T(det["roi_center"]), resize_ratio=T(det["resize_ratio"]),
        roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extent=T(det["roi_extent"]),
        scale=T(det["scale"]), score=T(det["score"]), im_W=T(det["im_W"]), im_H=T(det["im_H"]))
    K_crop = S.zoom_K_np(det["roi_cam"], det["roi_center"], det["scale"], 64)
    if with_depth:
        # sensor depth: HIP render of the GT pose at 64^2, nearest x4 to 256^2, + N(0, 2 mm), 5 % holes (§8d)
        depth = hip_lib.render_depth(meshes, T(det["roi_cls"].astype(np.int32)), T(K_crop), T(det["R_gt"]),
                                     T(det["t_gt"]), 64)
        big = depth.repeat_interleave(4, 1).repeat_interleave(4, 2)
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        noisy = torch.where(big > 0, big + 0.002 * torch.randn(big.shape, device=dev, generator=g), big)
        drop = torch.rand(big.shape, device=dev, generator=g) < 0.05
        batch["roi_depth"] = torch.where(drop, torch.zeros_like(noisy), noisy)[:, None].contiguous()
    return batch, det, K_crop


def cpu_baseline(det, K_crop, out_np, roi_depth_np, verts, faces, n, iters, thr, seconds=10.0):
    """Reference CPU refine path (gdrn_evaluator.py:485-561 per ROI) restated: oracle port, 1 thread.  The first ``n`` ROIs of
    the batch are processed round-robin until ``seconds`` of CPU work have been spent (bounded sample)."""
    from oracle import postproc as P  # the only place bench.py touches the oracle

    n = min(n, len(det["scale"]))
    mask = P.get_out_mask(out_np["mask"][:n])
    xyzs = [np.concatenate([out_np["coor_x"][i], out_np["coor_y"][i], out_np["coor_z"][i]], 0).transpose(1, 2, 0) for i in range(n)]
    done = 0
    t0 = time.perf_counter()
    while True:
        i = done % n
        o = int(det["roi_cls"][i])
        P.depth_refine_roi(xyzs[i], mask[i, 0], roi_depth_np[i, 0], K_crop[i], out_np["rot"][i], out_np["trans"][i],
                           verts[o], faces[o], iters=iters, threshold=thr)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and done >= n:
            break
    return dict(value=done / dt, unit="ROIs/s", cores=1, kind="port",
                sample=f"{done} ROI refinements ({n} distinct ROIs of the same batch, round-robin) through "
                       f"oracle.postproc.depth_refine_roi (NumPy + C software rasteriser in place of the vispy GL render), "
                       f"post-processing stage only; {dt:.2f} s")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
    hip_lib.load()

    refine = args.workload in ("refine", "bop7")
    opts = ["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"] if refine else []
    datasets = ["lmo", "ycbv", "tless", "icbin", "hb", "itodd", "tudl"] if args.workload == "bop7" else ["ycbv"]
    b = args.batch or (128 if refine else 64)
    torch.backends.cudnn.benchmark = True  # MIOpen find mode during warm-up
    hip_layers.set_enabled(not args.no_hip_layers)
    hip_layers.set_mlp_gemm("torch" if args.no_fused_mlp else args.mlp_gemm)
    hip_layers.set_conv_split(not args.no_conv_split)
    if args.library_below_tiles is not None:
        hip_layers.set_library_below_tiles(args.library_below_tiles)

    streams = []  # one (cfg, model, post, batch, det, K_crop, meshes, verts, faces) per dataset of the stream
    for di, ds in enumerate(datasets):
        cfg = get_cfg(f"{ds}_convnext_a6", opts)
        torch.manual_seed(20220925)  # identical weights on every rank; the data below is per-rank
        rng = np.random.default_rng(20220925 + 3 + rank + 100 * di)
        model, _ = build_model_optimizer(cfg, is_test=True)
        model.exact_reference_order = bool(args.exact_reference_order)
        # Random-init weights predict t ~ 0 (object at the camera centre), which no trained model does and which
        # would make every triangle straddle the camera plane.  Set the translation head's bias to the dataset
        # prior of the scale-invariant depth z_rel = t_z / resize_ratio (SITE parametrisation,
        # pose_from_pred_centroid_z.py:84-90) so that predicted poses land in the view frustum.
        with torch.no_grad():
            model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 1.5 * float(S.YCBV_K[0, 0]) * 0.19 / 64.0]))
        C = cfg.MODEL.POSE_NET.NUM_CLASSES
        verts, faces, ext = S.make_models(C, np.random.default_rng(20220925 + di), subdiv=args.subdiv)
        meshes = hip_lib.MeshSet(verts, faces, device=dev)
        post = GdrnHipPost(cfg, meshes if refine else None)
        batch, det, K_crop = make_batch(cfg, b, rng, dev, verts, faces, ext, meshes, refine)
        if args.with_crop:
            g = torch.Generator(device=dev).manual_seed(7 + di)
            batch["images"] = torch.randint(0, 256, (16, S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=dev, generator=g)
            batch["depths"] = torch.rand((16, S.IM_H, S.IM_W), device=dev, generator=g) + 0.3
            batch["im_idx"] = torch.from_numpy(rng.integers(0, 16, b).astype(np.int32)).to(dev)
            batch["center64"] = torch.from_numpy(det["roi_center"].astype(np.float64)).to(dev)
            batch["scale64"] = torch.from_numpy(det["scale"].astype(np.float64)).to(dev)
        streams.append(dict(cfg=cfg, model=model, post=post, batch=batch, det=det, K_crop=K_crop, meshes=meshes,
                            verts=verts, faces=faces, C=C, cls_i32=batch["roi_cls"].to(torch.int32)))
    cfg, model, post, batch, det, K_crop, meshes, verts, faces, C = (streams[0][k] for k in (
        "cfg", "model", "post", "batch", "det", "K_crop", "meshes", "verts", "faces", "C"))
    roi_ids = torch.arange(rank * b, (rank + 1) * b, dtype=torch.int32, device=dev)
    step_counter = [0]

    ev_pairs = []
    t_ref_buf = torch.empty((b, 3), dtype=torch.float64, device=dev)

    @torch.no_grad()
    def forward_only(st=None):
        st = st or streams[0]
        bt, mdl = st["batch"], st["model"]
        roi_img, roi_c2d = bt["roi_img"], bt["roi_coord_2d"]
        if args.with_crop:  # ROI preparation on the GPU (data_loader.py:754-797): full images -> ROI tensors
            roi_img, _roi_depth_from_image, roi_c2d = hip_lib.crop_resize_roi(
                bt["images"], bt["depths"], bt["im_idx"], bt["center64"], bt["scale64"])
        return mdl(roi_img, roi_classes=bt["roi_cls"], roi_cams=bt["roi_cam"], roi_whs=bt["roi_wh"],
                   roi_centers=bt["roi_center"], resize_ratios=bt["resize_ratio"], roi_coord_2d=roi_c2d,
                   roi_extents=bt["roi_extent"])

    fixed_out = forward_only() if args.post_only else None

    graphed = {}
    record_events_only_path = False
    side_streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else []
    if args.streams > 1:
        for st_ in streams:
            parts = []
            for ci in range(args.streams):
                lo, hi = ci * b // args.streams, (ci + 1) * b // args.streams
                sub = {k: (v[lo:hi].contiguous() if isinstance(v, torch.Tensor) and v.shape[:1] == (b,) else v)
                       for k, v in st_["batch"].items()}
                sub["roi_ids"] = roi_ids[lo:hi].contiguous()
                parts.append(sub)
            st_["parts"] = parts

    @torch.no_grad()
    def step(record_events=False):
        st = streams[step_counter[0] % len(streams)]
        step_counter[0] += 1
        if args.streams > 1 and not record_events_only_path:
            outs = []
            cur = torch.cuda.current_stream()
            for si, (sub, ss) in enumerate(zip(st["parts"], side_streams)):
                ss.wait_stream(cur)
                with torch.cuda.stream(ss):
                    o = st["model"](sub["roi_img"], roi_classes=sub["roi_cls"], roi_cams=sub["roi_cam"],
                                    roi_whs=sub["roi_wh"], roi_centers=sub["roi_center"],
                                    resize_ratios=sub["resize_ratio"], roi_coord_2d=sub["roi_coord_2d"],
                                    roi_extents=sub["roi_extent"])
                    outs.append(st["post"].process(sub, o, sub["roi_ids"]))
            for ss in side_streams:
                cur.wait_stream(ss)
            return gather_records(torch.cat(outs, 0), b)
        if args.graph and not args.with_crop and not args.post_only:
            key = id(st)
            if key not in graphed:
                graphed[key] = GraphedInference(st["model"], st["post"], st["batch"], roi_ids)
            graphed[key].graph.replay()  # inputs already live in the graph's static buffers (resident in HBM)
            return gather_records(graphed[key].records, b)
        bt, cfg_s = st["batch"], st["cfg"]
        out = fixed_out if args.post_only else forward_only(st)
        if refine and record_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K_c = hip_lib.zoom_K(bt["roi_cam"].reshape(b, 9), bt["roi_center"], bt["scale"], 64)
            maps = [out[k].contiguous() for k in ("coor_x", "coor_y", "coor_z", "mask")]
            rot9, trans = out["rot"].reshape(b, 9).contiguous(), out["trans"].contiguous()
            e0.record()  # torch's current stream == the stream hip_lib launches on; brackets ONLY the refine launch
            t_ref = hip_lib.depth_refine(st["meshes"], st["cls_i32"], maps[0], maps[1], maps[2], maps[3], bt["roi_depth"],
                                         K_c, rot9, trans, iters=cfg_s.TEST.DEPTH_REFINE_ITER,
                                         threshold=cfg_s.TEST.DEPTH_REFINE_THRESHOLD, out=t_ref_buf)
            e1.record()
            ev_pairs.append((e0, e1))
            rec = hip_lib.pack_pose_records(rot9, t_ref, trans, bt["score"], st["cls_i32"], roi_ids)
        else:
            rec = st["post"].process(bt, out, roi_ids)
        return gather_records(rec, b)

    # at least one untimed step: MIOpen's find mode and hipFuncSetAttribute run on first use
    for _ in range(max(args.warmup, 1) * len(streams)):
        step()
    step_counter[0] = 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gemm_timer = hip_lib.LaunchTimer() if (not args.graph and args.streams == 1) else None
    hip_lib.set_launch_timer(gemm_timer)
    for _ in range(args.steps):
        rec = step(record_events=True)
    hip_lib.set_launch_timer(None)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-stage split (outside the timed region)
    def timed(fn, n=5):
        torch.cuda.synchronize()
        s = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - s) / n * 1e3

    fwd_ms = timed(forward_only) if not args.post_only else None

    roofline = None
    refine_roofline = None
    if gemm_timer is not None and gemm_timer.records:
        # every gemm_split_kernel launch of the timed region: fp32-equivalent flops and event-measured duration
        fl = sum(r[1] for r in gemm_timer.records)
        ms_all = sum(r[2].elapsed_time(r[3]) for r in gemm_timer.records)
        n_l = len(gemm_timer.records)
        bf16_tflops = 6.0 * fl / (ms_all * 1e-3) / 1e12
        by_kind = {}
        for kind in ("linear", "linear_splitk", "conv3x3", "conv"):
            rs = [r for r in gemm_timer.records if r[0] == kind]
            if rs:
                t_k = sum(r[2].elapsed_time(r[3]) for r in rs)
                by_kind[kind] = dict(launches_per_step=len(rs) // args.steps, ms_per_step=t_k / args.steps,
                                     fp32_equiv_tflops=sum(r[1] for r in rs) / (t_k * 1e-3) / 1e12)
        g_traffic = None  # HBM-side bytes per launch from the committed PMC passes (same workload only)
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and b == 128 and args.workload == "refine" and args.mlp_gemm == "split" and not args.no_fused_mlp:
            g_traffic = json.load(open(pmc)).get("gemm_split_kernel", {}).get("traffic_bytes_per_launch")
        roofline = dict(kernel="gemm_split_kernel", bound="mfma", achieved=bf16_tflops, peak=BF16_MFMA_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=bf16_tflops / BF16_MFMA_PEAK_TFLOPS, traffic=g_traffic,
                        algorithmic_bytes_per_launch=sum(r[4] for r in gemm_timer.records) / n_l,
                        launch_ms=ms_all / n_l, launches_per_step=n_l // args.steps, ms_per_step=ms_all / args.steps,
                        flops_per_launch=6.0 * fl / n_l, fp32_equiv_tflops=fl / (ms_all * 1e-3) / 1e12,
                        fp32_mfma_peak_tflops=F32_MFMA_PEAK_TFLOPS, by_kind=by_kind,
                        note="bf16 MFMA flops executed = 6 x fp32-equivalent flops (exact 3-way operand split, six "
                             "partial products, fp32 accumulate)")
    if refine and ev_pairs:
        ms = [a.elapsed_time(bb) for a, bb in ev_pairs]
        mean_ms = float(np.mean(ms))
        nv, nf = len(verts[0]), len(faces[0])
        bytes_launch, per_roi = algorithmic_bytes_refine(b, cfg.TEST.DEPTH_REFINE_ITER, nv, nf)
        achieved = bytes_launch / (mean_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), same workload only
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and b == 128 and args.subdiv == 4 and args.workload == "refine":
            traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
        refine_roofline = dict(kernel="depth_refine_staged_kernel", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS,
                               unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic, launch_ms=mean_ms,
                               bytes_per_launch=bytes_launch, bytes_per_roi=per_roi, rois_per_launch=b)
        if roofline is None:
            roofline = refine_roofline

    # secondary hand-written kernels at the same batch, timed live with HIP events after the timed region
    def ev_time(fn, n=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    others = []
    if rank == 0 and refine and not args.post_only:
        out = forward_only()
        maps = [out[k].contiguous() for k in ("coor_x", "coor_y", "coor_z", "mask")]
        imwh = torch.stack([batch["im_W"], batch["im_H"]], 1).contiguous()
        t = ev_time(lambda: hip_lib.decode_correspondences(maps[0], maps[1], maps[2], maps[3], batch["roi_coord_2d"],
                                                           batch["roi_extent"], imwh))
        nsel = int(hip_lib.decode_correspondences(maps[0], maps[1], maps[2], maps[3], batch["roi_coord_2d"],
                                                  batch["roi_extent"], imwh)[0].sum())
        by = b * (98304 + 48 + 16384 + 4) + nsel * 24
        others.append(dict(kernel="decode_corr_kernel", bound="hbm", achieved=by / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=by / t / 1e9 / HBM_PEAK_GBS, launch_ms=t * 1e3, bytes_per_launch=by))
        imgs = torch.randint(0, 256, (16, S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=dev)
        deps = torch.rand((16, S.IM_H, S.IM_W), device=dev)
        imi = torch.from_numpy(np.random.default_rng(1).integers(0, 16, b).astype(np.int32)).to(dev)
        c64 = torch.from_numpy(det["roi_center"].astype(np.float64)).to(dev)
        s64 = torch.from_numpy(det["scale"].astype(np.float64)).to(dev)
        t = ev_time(lambda: hip_lib.crop_resize_roi(imgs, deps, imi, c64, s64))
        by = b * (3 * 256 * 256 * 4 + 256 * 256 * 4 + 2 * 64 * 64 * 4) + int(sum(7 * float(v) ** 2 for v in det["scale"]))
        others.append(dict(kernel="crop_img_depth_kernel+crop_coord2d_kernel", bound="hbm", achieved=by / t / 1e9,
                           peak=HBM_PEAK_GBS, unit="GB/s", frac=by / t / 1e9 / HBM_PEAK_GBS, launch_ms=t * 1e3,
                           bytes_per_launch=by))

    if rank == 0:
        cpu = None
        if refine and not args.no_cpu_baseline and world == 1:   # the CPU leg runs on rank 0 at N=1 only
            out = forward_only()
            torch.cuda.synchronize()
            out_np = {k: out[k].detach().cpu().numpy() for k in ("mask", "coor_x", "coor_y", "coor_z", "rot", "trans")}
            cpu = cpu_baseline(det, K_crop, out_np, batch["roi_depth"].cpu().numpy(), verts, faces, args.cpu_sample,
                               cfg.TEST.DEPTH_REFINE_ITER, cfg.TEST.DEPTH_REFINE_THRESHOLD, args.cpu_seconds)
            cpu["host_cores_available"] = os.cpu_count()
        total_rois = world * b * args.steps
        line = {
            "metric": "ROIs/sec (GDRNPP fwd + PnP + depth refine), 256x256 crops" if refine
            else "ROIs/sec (GDRNPP fwd + Patch-PnP, RGB only), 256x256 crops",
            "value": total_rois / dt, "unit": "ROIs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded ROIs, ellipsoid meshes 2562V/5120F, random-init weights, t-head bias = z_rel prior)",
            "config": {
                "workload": ("BOP-7 mixed stream (lmo/ycbv/tless/icbin/hb/itodd/tudl convnext_a6 models cycled per step) "
                             "+ fast depth refine, batch=%d ROIs/GPU" % b) if args.workload == "bop7" else
                            (("YCB-V convnext_a6 + fast depth refine (render-compare), batch=%d ROIs/GPU"
                              % b) if refine else ("YCB-V convnext_a6, RGB-only Patch-PnP, batch=%d ROIs/GPU" % b)),
                "baseline_config_index": 4 if args.workload == "bop7" else (2 if refine else 1),
                "roi_prep_on_gpu": bool(args.with_crop), "hipgraph": bool(args.graph), "streams": args.streams, "global_batch": world * b, "rois_per_gpu": b,
                "num_classes": C, "input_res": 256, "output_res": 64, "refine_iters": cfg.TEST.DEPTH_REFINE_ITER if refine else 0,
                "parallelism": f"roi-shard x{world}", "class_sliced_out_layer": not args.exact_reference_order, "hip_network_layers": not args.no_hip_layers, "mlp_gemm": "torch" if args.no_fused_mlp else args.mlp_gemm, "conv3x3_split": not args.no_conv_split and not args.no_fused_mlp and args.mlp_gemm == "split",
                "post_only": bool(args.post_only)},
            "roofline": roofline,
            "roofline_other_kernels": ([refine_roofline] if refine_roofline and refine_roofline is not roofline else []) + others,
            "cpu_baseline": cpu,
            "stages_ms": {"forward": fwd_ms, "depth_refine": refine_roofline["launch_ms"] if refine_roofline else None},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
