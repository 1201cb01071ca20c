"""Upper bound of moving the A-operand split out of the GEMM: the stage-2 / stage-0 MLP shapes and the
64x64 head convolution on the active library (GDRNPP_HIP_LIB) — run once with the product build and once with a timing-only build
of gemm_split_pipe.hip compiled with -DGDRNPP_TIMING_NO_SPLIT (k-loop without the split arithmetic; results invalid).  GPU box."""
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
for name, M, C in [("stage 2", 32768, 512), ("stage 1", 131072, 256), ("stage 0", 524288, 128)]:
    x = torch.randn(M, C, device=dev); h = torch.randn(M, 4 * C, device=dev)
    w1 = torch.randn(4 * C, C, device=dev) * 0.05; w2 = torch.randn(C, 4 * C, device=dev) * 0.05
    b1 = torch.randn(4 * C, device=dev); b2 = torch.randn(C, device=dev); g = torch.rand(C, device=dev); r = torch.randn(M, C, device=dev)
    p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
    t1 = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu"))
    t2 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r))
    fl = 2.0 * M * C * 4 * C
    print(f"{name}: fc1 {t1 * 1e3:.1f} us {6 * fl / t1 / 1e12:.3f} PF | fc2 {t2 * 1e3:.1f} us {6 * fl / t2 / 1e12:.3f} PF   lib={os.environ.get('GDRNPP_HIP_LIB', 'default')[-40:]}")
