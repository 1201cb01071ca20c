"""Why does dwconv7x7+LN lose x2.4 when a second step shares the chip (round-5 verdict item 4)?  A controlled experiment instead of
counters (rocprofv3 --pmc serialises the dispatches of the two streams, so a counter pass cannot see the sharing):

  dwconv launches of one ConvNeXt stage on stream A, the stage-2 fc1 three-product GEMM (80 KB of LDS per workgroup, two
  workgroups per CU) on stream B; both timed alone and beside each other, for these forms of the dwconv kernel:
    lds          the product form: [49][C] weights in LDS (98 KB at C = 512), persistent workgroups
    nolds        weights through L1 / L2, no LDS at all
    nolds+98K / +80K / +60K   the SAME no-LDS code, holding that much LDS without using it
  If the slowdown is the LDS allocation (a 98 KB workgroup cannot start on a CU that still holds ONE 80 KB GEMM workgroup, while a
  new GEMM workgroup can), "nolds" shares well and "nolds+98K" loses again — with identical instructions.

    python tools/dwconv_shared_probe.py > gpurun_out/dwconv_shared_probe.jsonl"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdrnpp_bop2022_amd import hip_lib  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling import engine  # noqa: E402

dev = torch.device("cuda", 0)
hip_lib.load()
torch.manual_seed(0)
N = 128
STAGES = {"stage0": (64, 128), "stage1": (32, 256), "stage2": (16, 512)}
FORMS = [("lds", 1, 0), ("nolds", 0, 0), ("nolds+98K", 0, 100352), ("nolds+80K", 0, 81920), ("nolds+60K", 0, 61440)]
K_DW, K_GEMM = 12, 6

xg = torch.randn(N * 256, 512, device=dev)
wg = hip_lib.pack_weight_f16x2(torch.randn(2048, 512, device=dev) * 512 ** -0.5)
bg = torch.randn(2048, device=dev)


def gemm():
    for _ in range(K_GEMM):
        hip_lib.linear_f32_split(xg, wg, bg, "gelu")


def timed(fn_a, fn_b, sa, sb, reps=5):
    """-> (ms of fn_a on sa, ms of fn_b on sb, makespan ms) with both launched together (either may be None)."""
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        if fn_b is not None:
            with torch.cuda.stream(sb):
                sb.wait_event(ev[0])
                ev[3].record()
                fn_b()
                ev[4].record()
        if fn_a is not None:
            with torch.cuda.stream(sa):
                sa.wait_event(ev[0])
                ev[1].record()
                fn_a()
                ev[2].record()
        torch.cuda.synchronize()
        a = ev[1].elapsed_time(ev[2]) if fn_a is not None else 0.0
        b = ev[3].elapsed_time(ev[4]) if fn_b is not None else 0.0
        span = max(ev[0].elapsed_time(ev[2]) if fn_a is not None else 0.0, ev[0].elapsed_time(ev[4]) if fn_b is not None else 0.0)
        if best is None or span < best[2]:
            best = (a, b, span)
    return best


streams = engine.StepStreams(2, dev)
sa, sb = streams.streams
print(json.dumps({"overlap_probe": streams.overlap_probe, "device": torch.cuda.get_device_name(0)}), flush=True)
gemm()
torch.cuda.synchronize()
_, g_alone, _ = timed(None, gemm, sa, sb)
for stage, (hw, c) in STAGES.items():
    x = torch.randn(N, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w49 = torch.randn(49, c, device=dev) * 0.1
    b, lw, lb = torch.randn(c, device=dev) * 0.1, 1 + 0.1 * torch.randn(c, device=dev), 0.1 * torch.randn(c, device=dev)
    ref = None
    for form, lds_w, pad in FORMS:
        hip_lib.set_option("dwconv_lds_w", lds_w)
        hip_lib.set_option("dwconv_lds_pad", pad)

        def dw():
            y = None
            for _ in range(K_DW):
                y = hip_lib.dwconv7x7_ln(x, w49, b, lw, lb, 1e-6, y_rows=True)
            return y

        y = dw()
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(y, ref))
        d_alone, _, _ = timed(dw, None, sa, sb)
        d_be, g_be, span = timed(dw, gemm, sa, sb)
        print(json.dumps({"stage": stage, "C": c, "HxW": hw, "form": form, "bit_equal_to_lds_form": same,
                          "dwconv_us_alone": 1e3 * d_alone / K_DW, "dwconv_us_beside_gemm": 1e3 * d_be / K_DW,
                          "slowdown": d_be / d_alone, "gemm_us_alone": 1e3 * g_alone / K_GEMM, "gemm_us_beside_dwconv": 1e3 * g_be / K_GEMM,
                          "serial_ms": d_alone + g_alone, "together_ms": span, "gain_ms": d_alone + g_alone - span}), flush=True)
hip_lib.set_option("dwconv_lds_w", 1)
hip_lib.set_option("dwconv_lds_pad", 0)
