"""Split GEMM at the stage-0 ConvNeXt-B MLP shapes (128 ROIs): epilogue cost and tile-size A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
torch.manual_seed(0)
M, c = 128 * 4096, 128
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


for mi4 in ("0", "1"):
    os.environ["GDRNPP_SPLIT_MI4"] = mi4
    r = [t(lambda: hip_lib.linear_f32_split(x, p1, b1, "none")), t(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu")),
         t(lambda: hip_lib.linear_f32_split(h, p2, b2, "none")), t(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc))]
    print(f"MI4={mi4}: fc1 none {r[0]:.3f} gelu {r[1]:.3f} | fc2 none {r[2]:.3f} scale_res {r[3]:.3f} ms")
