"""Bound of a Winograd F(2x2,3x3) form of the head's 3x3 convolutions on the split-GEMM engine.

F(2x2,3x3) replaces the direct implicit GEMM [P x 9*Cin] x [9*Cin x Cout] (P = 128*64*64 pixels) by 16 independent GEMMs
[P/4 x Cin] x [Cin x Cout] — one per position of the 4x4 transformed tile — 4/9 of the multiplies.  Whatever a fused kernel
does around them (4-pixel input transform at fragment-read time, 16 weight matrices through LDS, output transform across
the 16 accumulator sets), it cannot be faster than those 16 short-K GEMMs run by the same pipelined kernel with PERFECT
weight reuse (256-row tiles of one position share a weight tile) and NO result traffic.  That lower bound is measured here
with the grouped launch: 16 groups (positions) of P/4 rows, K = Cin = 256, N = Cout = 256, n_store = 4 (no stores), against
the direct convolution the path runs today."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdrnpp_bop2022_amd import hip_lib as hip  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
n, c, h = 128, 256, 64
tiles = n * h * h // 4


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.randn(n, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(c, c, 3, 3, device=dev) * 0.03
pk = hip.pack_conv_weight_bf16x3(wt)
t_direct = timeit(lambda: hip.conv3x3_f32_split(x, pk, None))
g = torch.rand(c, device=dev) + 0.5
be = torch.randn(c, device=dev) * 0.1
t_fused = timeit(lambda: hip.conv3x3_groupnorm_act(x, pk, None, g, be, 32, 1e-5, gelu=True))

v = torch.randn(16 * tiles, c, device=dev)                      # stands for the transformed input V[16][tiles][Cin]
u = torch.randn(16 * c, c, device=dev) * (c ** -0.5)            # 16 transformed weight matrices U[16][Cout][Cin]
upk = hip.pack_weight_bf16x3(u)
ub = torch.zeros(16, c, device=dev)
sel = torch.arange(16, device=dev, dtype=torch.int32)
t_nostore = timeit(lambda: hip.linear_f32_split_grouped(v, upk, ub, sel, tiles, n_store=4))
t_store = timeit(lambda: hip.linear_f32_split_grouped(v, upk, ub, sel, tiles, n_store=c))
fl_direct = 2.0 * n * h * h * 9 * c * c
fl_wino = 2.0 * 16 * tiles * c * c
print(f"direct conv3x3 128x64x64x256->256: {t_direct:.3f} ms ({fl_direct / t_direct / 1e9:.0f} TFLOP/s fp32-equivalent); "
      f"+ fused GroupNorm/GELU apply pass: {t_fused:.3f} ms")
print(f"16 x [{tiles} x 256] x [256 x 256] GEMMs, perfect weight reuse, no stores: {t_nostore:.3f} ms "
      f"({fl_wino / t_nostore / 1e9:.0f} TFLOP/s); with the 4x-size M tensor written: {t_store:.3f} ms")
print(f"upper bound of the gain per convolution: {t_direct - t_nostore:.3f} ms; x 6 convolutions per step: {6 * (t_direct - t_nostore):.2f} ms")
