"""Per-shape time of the split GEMM at the ConvNeXt-B MLP shapes of B ROIs (default 128): fc1 (GELU) and fc2 (layer scale +
residual) of the four stages — all eight have the same 34.4 G multiply-adds, so the spread IS the short-K / wide-output loss."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdrnpp_bop2022_amd import hip_lib as hip  # noqa: E402

B = int(os.environ.get("B", "128"))
dev = "cuda"
torch.manual_seed(0)
tot = 0.0
for st, (hw, c, nblk) in enumerate([(64, 128, 3), (32, 256, 3), (16, 512, 27), (8, 1024, 3)]):
    m = B * hw * hw
    for name, k, n, epi in (("fc1", c, 4 * c, "gelu"), ("fc2", 4 * c, c, "scale_res")):
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) * k ** -0.5
        b = torch.randn(n, device=dev)
        g = torch.randn(n, device=dev) if epi == "scale_res" else None
        r = torch.randn(m, n, device=dev) if epi == "scale_res" else None
        pk = hip.pack_weight_bf16x3(w)
        fn = lambda: hip.linear_f32_split(x, pk, b, epi, g, r)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        by = 4.0 * m * k + 6.0 * n * k + 4.0 * m * n * (2 if epi == "scale_res" else 1)
        print(f"stage {st} {name}: M={m} K={k} N={n}: {t * 1e3:.0f} us  {2.0 * m * n * k / t / 1e9:.0f} TFLOP/s fp32-equivalent  "
              f"{by / t / 1e6:.0f} GB/s algorithmic")
        tot += nblk * t
print(f"36 blocks: {tot:.2f} ms")
# the fused one-launch form of the stage-0 block (C = 128)
m, c = B * 64 * 64, 128
x = torch.randn(m, c, device=dev); res = torch.randn(m, c, device=dev)
w1 = torch.randn(4 * c, c, device=dev) * c ** -0.5; w2 = torch.randn(c, 4 * c, device=dev) * (4 * c) ** -0.5
b1 = torch.randn(4 * c, device=dev); b2 = torch.randn(c, device=dev); g = torch.rand(c, device=dev)
p1, p2 = hip.pack_weight_bf16x3(w1), hip.pack_weight_bf16x3(w2)
fn = lambda: hip.convnext_mlp_f32_split(x, p1, b1, p2, b2, g, res)  # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print(f"stage 0 fused fc1+GELU+fc2+residual: M={m}: {t * 1e3:.0f} us  {2.0 * m * 8 * c * c / t / 1e9:.0f} TFLOP/s fp32-equivalent")
