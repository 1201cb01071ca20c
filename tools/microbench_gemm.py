"""gdrnpp_linear_f32 (fused epilogues) vs PyTorch-ROCm (hipBLASLt GEMM + separate GELU / addcmul) on the 8 MLP shapes
of ConvNeXt-B at 128 ROIs (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
B = int(os.environ.get("B", "128"))
torch.manual_seed(0)


def t(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot_t, tot_h, tot_s = 0.0, 0.0, 0.0
for (hw, c, nblk) in [(4096, 128, 3), (1024, 256, 3), (256, 512, 27), (64, 1024, 3)]:
    M = B * hw
    x = torch.randn(M, c, device=dev)
    w1 = torch.randn(4 * c, c, device=dev) * 0.05; b1 = torch.randn(4 * c, device=dev)
    w2 = torch.randn(c, 4 * c, device=dev) * 0.05; b2 = torch.randn(c, device=dev)
    gamma = torch.randn(c, device=dev); sc = torch.randn(M, c, device=dev)
    ref1 = F.gelu(F.linear(x, w1, b1))
    out1 = hip_lib.linear_f32(x, w1, b1, "gelu")
    ref2 = torch.addcmul(sc, F.linear(ref1, w2, b2), gamma)
    out2 = hip_lib.linear_f32(ref1, w2, b2, "scale_res", gamma, sc)
    e1 = ((out1 - ref1).abs().max() / ref1.abs().max()).item()
    e2 = ((out2 - ref2).abs().max() / ref2.abs().max()).item()
    tt1 = t(lambda: F.gelu(F.linear(x, w1, b1))); th1 = t(lambda: hip_lib.linear_f32(x, w1, b1, "gelu"))
    tt2 = t(lambda: torch.addcmul(sc, F.linear(ref1, w2, b2), gamma)); th2 = t(lambda: hip_lib.linear_f32(ref1, w2, b2, "scale_res", gamma, sc))
    fl = 2.0 * M * c * 4 * c
    p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
    assert torch.equal(hip_lib.unpack_weight_bf16x3(p1).float().sum(0), w1), "split is not exact"
    s1 = hip_lib.linear_f32_split(x, p1, b1, "gelu")
    s2 = hip_lib.linear_f32_split(ref1, p2, b2, "scale_res", gamma, sc)
    sub = slice(0, 2048)
    r1 = F.gelu(x[sub].double() @ w1.double().t() + b1.double())
    r2 = sc[sub].double() + gamma.double() * (ref1[sub].double() @ w2.double().t() + b2.double())
    d = lambda y, r: ((y[sub].double() - r).abs().max() / r.abs().max()).item()
    print(f"   vs fp64: fc1 torch {d(ref1, r1):.1e} hip-f32 {d(out1, r1):.1e} split {d(s1, r1):.1e} | "
          f"fc2 torch {d(ref2, r2):.1e} hip-f32 {d(out2, r2):.1e} split {d(s2, r2):.1e}")
    ts1 = t(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu")); ts2 = t(lambda: hip_lib.linear_f32_split(ref1, p2, b2, "scale_res", gamma, sc))
    print(f"   split: fc1 {ts1:.3f} ms ({fl / ts1 / 1e9:.0f} TF eff) fc2 {ts2:.3f} ms ({fl / ts2 / 1e9:.0f} TF eff)")
    tot_s += nblk * (ts1 + ts2)
    print(f"hw={hw} C={c}: fc1+gelu torch {tt1:.3f} ms / hip {th1:.3f} ms ({fl / th1 / 1e9:.0f} TF) err {e1:.1e} | "
          f"fc2+scale+res torch {tt2:.3f} / hip {th2:.3f} ms ({fl / th2 / 1e9:.0f} TF) err {e2:.1e}")
    tot_t += nblk * (tt1 + tt2); tot_h += nblk * (th1 + th2)
print(f"MLP total per forward: torch {tot_t:.2f} ms, hip-f32 {tot_h:.2f} ms, hip-split {tot_s:.2f} ms")
