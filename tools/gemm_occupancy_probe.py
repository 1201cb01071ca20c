"""One vs two workgroups per CU of the pipelined split GEMM: the stage-2 fc2 / fc1 shapes at 256 and at 512 output tiles
(one / two resident workgroups per CU, one round each for fc2).  If half the tiles take half the time, a lone wave per SIMD keeps
the matrix pipe as busy as two do.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib
dev = "cuda"; torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
C = 512
w1 = torch.randn(4 * C, C, device=dev) * 0.05; w2 = torch.randn(C, 4 * C, device=dev) * 0.05
b1 = torch.randn(4 * C, device=dev); b2 = torch.randn(C, device=dev); g = torch.rand(C, device=dev)
p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
for rep in range(2):
    for M in (8192, 16384, 32768, 65536):
        x = torch.randn(M, C, device=dev); h = torch.randn(M, 4 * C, device=dev); r = torch.randn(M, C, device=dev)
        t1 = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu"))
        t2 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r))
        fl = 2.0 * M * C * 4 * C
        print(f"M={M:6d}: fc1 {M // 256 * 16:5d} tiles {t1 * 1e3:7.1f} us {6 * fl / t1 / 1e12:.3f} PF | fc2 {M // 256 * 4:5d} tiles {t2 * 1e3:7.1f} us {6 * fl / t2 / 1e12:.3f} PF")
