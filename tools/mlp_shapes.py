"""Per-shape time of the split GEMM at the ConvNeXt-B MLP shapes of B ROIs (default 128): fc1 (GELU) and fc2 (layer scale +
residual) of the four stages — all eight have the same 34.4 G multiply-adds, so the spread IS the short-K / wide-output loss."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdrnpp_bop2022_amd import hip_lib as hip  # noqa: E402

B = int(os.environ.get("B", "128"))
X3 = os.environ.get("X3", "0") == "1"   # three-product (fp16x2) kernels
dev = "cuda"
for o in os.environ.get("OPTS", "").split():   # e.g. OPTS="split2_wide=1"
    hip.set_option(o.split("=")[0], int(o.split("=")[1]))
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
        pk = hip.pack_weight_f16x2(w) if X3 else hip.pack_weight_bf16x3(w)
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
