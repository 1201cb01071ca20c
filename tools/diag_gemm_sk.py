"""Stream-K schedule of the split GEMM against the one-tile-per-workgroup schedule: closeness (fp32 rounding of the partial
sums), determinism, error vs fp64, and timing at the real row counts of the 128-ROI step (GPU box)."""
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
bad = 0
B = int(os.environ.get("ROIS", "128"))
shapes = [("s2", B * 196, 512), ("s3", B * 49, 1024), ("s1", B * 784, 256), ("s0", B * 3136, 128), ("odd", 70001, 512), ("small", 3000, 512)]
for name, M, C in shapes:
    torch.manual_seed(1)
    x = torch.randn(M, C, device=dev); h = torch.randn(M, 4 * C, device=dev)
    w1 = torch.randn(4 * C, C, device=dev) * C ** -0.5; w2 = torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5
    b1 = torch.randn(4 * C, device=dev); b2 = torch.randn(C, device=dev); g = torch.rand(C, device=dev); r = torch.randn(M, C, device=dev)
    p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
    res = {}
    for sk in (0, 2):
        hip_lib.set_option("split_gemm_sk", sk)
        y1 = hip_lib.linear_f32_split(x, p1, b1, "gelu"); y2 = hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r)
        y1b = hip_lib.linear_f32_split(x, p1, b1, "gelu"); y2b = hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r)
        torch.cuda.synchronize()
        t1 = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu")); t2 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r))
        res[sk] = (y1, y2, t1, t2, torch.equal(y1, y1b) and torch.equal(y2, y2b))
    hip_lib.set_option("split_gemm_sk", 1)
    t1a = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu")); t2a = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", g, r))
    ref2 = (h[:4096].double() @ w2.double().T + b2.double()) * g.double() + r[:4096].double()
    e0 = ((res[0][1][:4096].double() - ref2).abs().max() / ref2.abs().max()).item(); e2 = ((res[2][1][:4096].double() - ref2).abs().max() / ref2.abs().max()).item()
    d1 = (res[0][0] - res[2][0]).abs().max().item() / res[0][0].abs().max().item(); d2 = (res[0][1] - res[2][1]).abs().max().item() / res[0][1].abs().max().item()
    fl = 2.0 * M * C * 4 * C
    ok = res[0][4] and res[2][4] and d1 < 2e-6 and d2 < 2e-6 and e2 < 1.5 * e0 + 2e-7
    bad += 0 if ok else 1
    print(f"{name} M={M} C={C}: fc1 dp {res[0][2]:.4f} sk {res[2][2]:.4f} auto {t1a:.4f} ms ({fl/res[0][2]/1e9:.0f} -> {fl/res[2][2]/1e9:.0f} TF) | fc2 dp {res[0][3]:.4f} sk {res[2][3]:.4f} auto {t2a:.4f} ms "
          f"({fl/res[0][3]/1e9:.0f} -> {fl/res[2][3]/1e9:.0f} TF) | rel diff {d1:.1e} {d2:.1e} err64 dp {e0:.2e} sk {e2:.2e} deterministic {res[0][4]} {res[2][4]} {'OK' if ok else 'BAD'}")
hip_lib.set_option("split_gemm_sk", 0)
print("SK_OK" if not bad else "SK_BAD")
