import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (32768, 25088, 16384):
    C = 512
    torch.manual_seed(1)
    x = torch.randn(M, C, device=dev); h = torch.randn(M, 4 * C, device=dev)
    w1 = torch.randn(4 * C, C, device=dev) * C ** -0.5; w2 = torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5
    b1 = torch.randn(4 * C, device=dev); b2 = torch.randn(C, device=dev); g = torch.rand(C, device=dev); r = torch.randn(M, C, device=dev)
    p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
    out = []
    for pipe in (3, 2):
        hip_lib.set_option("split_gemm_pipe", pipe)
        for sk in (0, 2):
            hip_lib.set_option("split_gemm_sk", sk)
            t1 = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu")); t2 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r))
            t0 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "none"))
            out.append(f"NA={pipe} sk={sk}: fc1 {t1:.4f} fc2 {t2:.4f} fc2-noepi {t0:.4f}")
    print(f"M={M}: " + " | ".join(out))
hip_lib.set_option("split_gemm_pipe", 3); hip_lib.set_option("split_gemm_sk", 0)
