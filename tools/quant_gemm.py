"""Tile-count quantisation of the split GEMM: row counts of 224x224 crops (56/28/14/7-pixel stages, uneven tile counts) against the
next evenly dividing ones (the 256x256 headline workload has 64/32/16/8-pixel stages and divides evenly).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib
dev = "cuda"; torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = int(os.environ.get("ROIS", "128"))
for name, hw, C in [("s0", 56 * 56, 128), ("s1", 28 * 28, 256), ("s2", 14 * 14, 512), ("s3", 7 * 7, 1024)]:
    for M in (B * hw, ((B * hw + 131071) // 131072) * 131072 if hw > 400 else ((B * hw + 8191) // 8192) * 8192):
        x = torch.randn(M, C, device=dev); h = torch.randn(M, 4 * C, device=dev)
        w1 = torch.randn(4 * C, C, device=dev) * 0.05; w2 = torch.randn(C, 4 * C, device=dev) * 0.05
        b1 = torch.randn(4 * C, device=dev); b2 = torch.randn(C, device=dev); g = torch.rand(C, device=dev); r = torch.randn(M, C, device=dev)
        p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
        t1 = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu"))
        t2 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r))
        fl = 2.0 * M * C * 4 * C
        tm = (M + 255) // 256
        print(f"{name} M={M:7d} m-tiles={tm:5d}  fc1 tiles={tm * 4 * C // 128:6d} {t1:.4f} ms {fl / t1 / 1e9:6.1f} TF | fc2 tiles={tm * C // 128:6d} {t2:.4f} ms {fl / t2 / 1e9:6.1f} TF")
