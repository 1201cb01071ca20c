#!/usr/bin/env python
"""Where do the A operands of the three-product launches sit relative to the range check of csrc/gemm_split2_pipe.hip?

Runs one forward of the headline model (YCB-V convnext_a6, seeded O(1) parameters = bench.py's, or --random-init) at --batch ROIs
with every three-product wrapper of hip_lib instrumented: for each launch the per-row mean square of the A operand (im2col rows
for the convolutions: box sum over the taps, zero padded) is compared with the kernel's threshold 2^-8 (rms 2^-4):
    rows_below   share of non-zero rows below the threshold          (what the per-row verdict of the kernel sees)
    tiles_below  share of 64-row wave tiles whose MEAN is below it    (what a per-tile verdict would see)
    min_rms      smallest non-zero row rms
and the kernel's own verdict (range word of the launch) is printed beside it — they must agree.
Usage: python tools/x3_row_scale_survey.py [--batch 128] [--random-init] [--image rand|noise|flat]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdrnpp_bop2022_amd import hip_lib, synthetic as S  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer  # noqa: E402

THR = 2.0 ** -8
rows_out = []


def stats(name, ms, slot):
    """ms: per-row mean square, 1-D."""
    ms = ms.reshape(-1).double()
    nz = ms > 0
    below = (ms < THR) & nz
    n = ms.numel()
    pad = (-n) % 64
    t = F.pad(ms, (0, pad)).view(-1, 64).mean(1)
    rows_out.append(dict(name=name, slot=slot, rows=n, rows_below=float(below.sum()) / max(int(nz.sum()), 1),
                         tiles_below=float(((t < THR) & (t > 0)).double().mean()), min_rms=float(ms[nz].min().sqrt()) if nz.any() else 0.0,
                         tensor_rms=float(ms.mean().sqrt())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--random-init", action="store_true")
    ap.add_argument("--image", default="rand", choices=["rand", "noise", "flat"])
    args = ap.parse_args()
    dev = torch.device("cuda")
    cfg = get_cfg("ycbv_convnext_a6", ["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    torch.manual_seed(20220925)
    model, _ = build_model_optimizer(cfg, is_test=True)
    if not args.random_init:
        model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 20220925), strict=True)
    b, C = args.batch, cfg.MODEL.POSE_NET.NUM_CLASSES
    rng = np.random.default_rng(3)
    _, _, ext = S.make_models(C, np.random.default_rng(20220925), subdiv=2)
    det = S.make_detections(b, C, ext, rng)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    img = {"rand": lambda: torch.rand(b, 3, 256, 256, device=dev), "noise": lambda: torch.randn(b, 3, 256, 256, device=dev),
           "flat": lambda: torch.full((b, 3, 256, 256), 0.3, device=dev)}[args.image]()

    lin, conv, cgn = hip_lib.linear_f32_split, hip_lib.conv2d_f32_split, hip_lib.conv3x3_groupnorm_act

    def conv_rows_ms(x_cl, kh, kw, stride, pad):
        px = x_cl.double().square().sum(1, keepdim=True)                       # per-pixel sum of squares over C
        box = F.avg_pool2d(px, (kh, kw), stride=stride, padding=pad, count_include_pad=True, divisor_override=1)
        return box / (kh * kw * x_cl.shape[1])

    def lin_w(x2d, wp, *a, **k):
        if wp.dtype == torch.float16:
            stats(f"{k.get('_kind', 'linear')} M={x2d.shape[0]} K={x2d.shape[1]} N={wp.shape[0] * 128}", x2d.double().square().mean(1),
                  k.get("x3_slot", 0))
        return lin(x2d, wp, *a, **k)

    def conv_w(x_cl, wp, bias, kh, kw, stride, pad, *a, **k):
        if wp.dtype == torch.float16:
            stats(f"{k.get('_kind', 'conv')} {tuple(x_cl.shape)} {kh}x{kw}/{stride}", conv_rows_ms(x_cl, kh, kw, stride, pad), k.get("x3_slot", 0))
        return conv(x_cl, wp, bias, kh, kw, stride, pad, *a, **k)

    def cgn_w(x_cl, wp, *a, **k):
        if wp.dtype == torch.float16:
            stats(f"conv3x3+gn {tuple(x_cl.shape)}", conv_rows_ms(x_cl, 3, 3, 1, 1), k.get("x3_slot", 0))
        return cgn(x_cl, wp, *a, **k)

    hip_lib.linear_f32_split, hip_lib.conv2d_f32_split, hip_lib.conv3x3_groupnorm_act = lin_w, conv_w, cgn_w
    with torch.no_grad():
        model(img, roi_classes=T(det["roi_cls"]), roi_cams=T(det["roi_cam"]), roi_whs=T(det["roi_wh"]), roi_centers=T(det["roi_center"]),
              resize_ratios=T(det["resize_ratio"]), roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])),
              roi_extents=T(det["roi_extent"]))
    words = hip_lib.split2_range_words()
    print(f"# batch={b} parameters={'default-init' if args.random_init else 'seeded O(1)'} image={args.image}: "
          f"{len(rows_out)} three-product launches, kernel range words {words}")
    print("| slot | launch | rows | tensor rms | min row rms | rows below 2^-4 | 64-row tiles below | kernel word |")
    print("|---|---|---|---|---|---|---|---|")
    bad = 0
    for r in rows_out:
        w = words.get(r["slot"], 0)
        expect = r["rows_below"] > 0
        bad += int(expect != bool(w & hip_lib.X3_SMALL_ROWS))
        print(f"| {r['slot']} | {r['name']} | {r['rows']} | {r['tensor_rms']:.3g} | {r['min_rms']:.3g} | {r['rows_below']:.2e} | "
              f"{r['tiles_below']:.2e} | {w} |")
    print(f"# launches where the kernel's verdict differs from this survey's: {bad} (rows within 1 ulp of the threshold may differ)")


if __name__ == "__main__":
    main()
