"""Bound of a Winograd F(2x2,3x3) form of the head's 3x3 convolutions on the split-GEMM engine.

F(2x2,3x3) replaces the direct implicit GEMM [P x 9*Cin] x [9*Cin x Cout] (P pixels) by 16 independent GEMMs
[P/4 x Cin] x [Cin x Cout] — one per position of the 4x4 transformed tile — 4/9 of the multiplies.  Two bounds per map size:

  (a) whatever a kernel does around them, it cannot beat those 16 short-K GEMMs run by the pipelined kernel with PERFECT weight
      reuse (256-row tiles of one position share a weight tile) and NO result traffic: the grouped launch with n_store = 4;
  (b) the form that fits this engine: the 16 accumulator sets of a tile cannot share a workgroup (16 weight matrices through one
      CU's LDS: 196 KB per k-tile against 12 KB today), so the products M[16][P/4][Cout] are written (4x the output bytes) and an
      output-transform pass reads them back: (a) with stores + that pass at the box's measured copy rate.
      The input transform is assumed free (done at fragment-read time) in (a) and (b) — which this engine cannot do either: the
      LDS-DMA moves one source pixel per A row, the transform needs four (64 KB per k-tile stage of 256 rows instead of 16);
  (c) the form that can be built from the existing kernels: an input-transform pass that writes V[16][P/4][Cin] (4x the input
      bytes), the grouped GEMM reading V and writing M, the output-transform pass: (b) + one more streaming pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdrnpp_bop2022_amd import hip_lib as hip  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
n, c = int(os.environ.get("B", "128")), 256


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


a = torch.empty(64 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
copy_bps = 2 * a.numel() * 4 / (timeit(lambda: b.copy_(a)) * 1e-3)
del a, b
print(f"copy rate of this box: {copy_bps / 1e12:.2f} TB/s read + written")
gain_a = gain_b = 0.0
for h in (64, 32, 16):                                # the head has two 3x3 convolutions at each of these map sizes
    tiles = n * h * h // 4
    x = torch.randn(n, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.03
    pk = hip.pack_conv_weight_bf16x3(wt)
    hip.set_conv_splitk(False)
    t_direct = timeit(lambda: hip.conv3x3_f32_split(x, pk, None))
    v = torch.randn(16 * tiles, c, device=dev)            # stands for the transformed input V[16][tiles][Cin]
    u = torch.randn(16 * c, c, device=dev) * (c ** -0.5)  # 16 transformed weight matrices U[16][Cout][Cin]
    upk = hip.pack_weight_bf16x3(u)
    ub = torch.zeros(16, c, device=dev)
    sel = torch.arange(16, device=dev, dtype=torch.int32)
    t_a = timeit(lambda: hip.linear_f32_split_grouped(v, upk, ub, sel, tiles, n_store=4))
    t_store = timeit(lambda: hip.linear_f32_split_grouped(v, upk, ub, sel, tiles, n_store=c))
    t_pass = (16 * tiles * c * 4 + n * h * h * c * 4) / copy_bps * 1e3          # read M, write y
    t_b = t_store + t_pass
    t_in = (n * h * h * c * 4 + 16 * tiles * c * 4) / copy_bps * 1e3           # read x, write V
    t_c = t_b + t_in
    gain_c = globals().get("gain_c", 0.0) + 2 * max(t_direct - t_c, 0.0)
    print(f"{h}x{h}: direct {t_direct:.3f} ms ({2.0 * n * h * h * 9 * c * c / t_direct / 1e9:.0f} TFLOP/s fp32-eq) | (a) 16 GEMMs, no stores "
          f"{t_a:.3f} ms | (b) M written {t_store:.3f} + output-transform pass {t_pass:.3f} = {t_b:.3f} ms | (c) + input-transform pass {t_in:.3f} = {t_c:.3f} ms")
    gain_a += 2 * (t_direct - t_a)
    gain_b += 2 * max(t_direct - t_b, 0.0)
print(f"per step (two convolutions per size): bound (a) {gain_a:.2f} ms, bound (b) {gain_b:.2f} ms, bound (c) {gain_c:.2f} ms")
