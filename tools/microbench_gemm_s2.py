"""Split GEMM at the stage-2 ConvNeXt-B MLP shapes only (128 ROIs): quick A/B of kernel variants (GDRNPP_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
torch.manual_seed(0)
M, c = 128 * 256, 512
x = torch.randn(M, c, device=dev)
w1 = torch.randn(4 * c, c, device=dev) * 0.05; b1 = torch.randn(4 * c, device=dev)
w2 = torch.randn(c, 4 * c, device=dev) * 0.05; b2 = torch.randn(c, device=dev)
gamma = torch.randn(c, device=dev); sc = torch.randn(M, c, device=dev)
p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
h = hip_lib.linear_f32_split(x, p1, b1, "gelu")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t1 = t(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu"))
t2 = t(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc))
fl = 2.0 * M * c * 4 * c
print(f"fc1 {t1:.3f} ms ({fl / t1 / 1e9:.0f} TF eff)  fc2 {t2:.3f} ms ({fl / t2 / 1e9:.0f} TF eff)  "
      f"lib={os.path.basename(os.environ.get('GDRNPP_HIP_LIB', 'default'))} 4wave={os.environ.get('GDRNPP_SPLIT_4WAVE', '0')}")
