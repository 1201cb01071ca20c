"""A/B of the two 256-row split-GEMM kernels (register-staged vs LDS-DMA) on the shapes of the ConvNeXt-B MLPs and the head
convolutions at 128 ROIs: bitwise comparison of the results and event-timed TFLOP/s.  Run on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


res = []
B = int(os.environ.get("ROIS", "128"))
lin = [("s0", B * 4096, 128), ("s1", B * 1024, 256), ("s2", B * 256, 512), ("s3", B * 64, 1024)]
for name, M, Cc in lin:
    x = torch.randn(M, Cc, device=dev)
    w1 = torch.randn(4 * Cc, Cc, device=dev) * 0.05; b1 = torch.randn(4 * Cc, device=dev)
    w2 = torch.randn(Cc, 4 * Cc, device=dev) * 0.05; b2 = torch.randn(Cc, device=dev)
    gamma = torch.rand(Cc, device=dev); sc = torch.randn(M, Cc, device=dev)
    p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
    out = {}
    for mode in (0, 1):
        hip_lib.set_option("split_gemm_glds", mode)
        h = hip_lib.linear_f32_split(x, p1, b1, "gelu")
        y = hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc)
        t1 = timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu"))
        t2 = timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc))
        out[mode] = (h, y, t1, t2)
    fl = 2.0 * M * Cc * 4 * Cc
    res.append(dict(shape=f"{name} fc1 M={M} K={Cc} N={4*Cc}", equal=bool(torch.equal(out[0][0], out[1][0])),
                    reg_tflops=fl / out[0][2] / 1e12, glds_tflops=fl / out[1][2] / 1e12, reg_ms=out[0][2] * 1e3, glds_ms=out[1][2] * 1e3))
    res.append(dict(shape=f"{name} fc2 M={M} K={4*Cc} N={Cc}", equal=bool(torch.equal(out[0][1], out[1][1])),
                    reg_tflops=fl / out[0][3] / 1e12, glds_tflops=fl / out[1][3] / 1e12, reg_ms=out[0][3] * 1e3, glds_ms=out[1][3] * 1e3))
    del x, w1, w2, sc, out
for (hw, cin) in [(64, 256), (32, 256), (16, 256)]:
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(256, cin, 3, 3, device=dev) * 0.05
    pk = hip_lib.pack_conv_weight_bf16x3(w)
    out = {}
    for mode in (0, 1):
        hip_lib.set_option("split_gemm_glds", mode)
        y = hip_lib.conv3x3_f32_split(x, pk, None)
        t = timeit(lambda: hip_lib.conv3x3_f32_split(x, pk, None))
        out[mode] = (y, t)
    fl = 2.0 * B * hw * hw * 256 * cin * 9
    res.append(dict(shape=f"conv3x3 {hw}x{hw} cin={cin}", equal=bool(torch.equal(out[0][0], out[1][0])),
                    reg_tflops=fl / out[0][1] / 1e12, glds_tflops=fl / out[1][1] / 1e12, reg_ms=out[0][1] * 1e3, glds_ms=out[1][1] * 1e3))
# general conv (2x2/2 downsample) and odd M
x = torch.randn(B, 256, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(512, 256, 2, 2, device=dev) * 0.05
pk = hip_lib.pack_conv_weight_bf16x3(w)
ys = []
for mode in (0, 1):
    hip_lib.set_option("split_gemm_glds", mode); hip_lib.set_option("split_gemm_mi4", 1 if mode else -1)
    ys.append(hip_lib.conv2d_f32_split(x, pk, None, 2, 2, 2, 0))
hip_lib.set_option("split_gemm_mi4", -1)
res.append(dict(shape="conv2x2/2 256->512 @32", equal=bool(torch.equal(ys[0], ys[1]))))
xo = torch.randn(70001, 512, device=dev); wo = torch.randn(2048, 512, device=dev) * 0.05; po = hip_lib.pack_weight_bf16x3(wo)
ys = []
for mode in (0, 1):
    hip_lib.set_option("split_gemm_glds", mode); hip_lib.set_option("split_gemm_mi4", 1 if mode else -1)
    ys.append(hip_lib.linear_f32_split(xo, po, None))
hip_lib.set_option("split_gemm_mi4", -1); hip_lib.set_option("split_gemm_glds", 0)
res.append(dict(shape="linear odd M=70001", equal=bool(torch.equal(ys[0], ys[1]))))
tot_r = sum(r.get("reg_ms", 0) * (27 if r["shape"].startswith("s2") else 3 if r["shape"][0] == "s" else 2) for r in res)
tot_g = sum(r.get("glds_ms", 0) * (27 if r["shape"].startswith("s2") else 3 if r["shape"][0] == "s" else 2) for r in res)
for r in res:
    print(json.dumps(r))
print(json.dumps(dict(step_estimate_reg_ms=tot_r, step_estimate_glds_ms=tot_g)))
