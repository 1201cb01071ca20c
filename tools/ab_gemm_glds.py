"""A/B of the 256-row split-GEMM kernels (register-staged, LDS-DMA, software-pipelined LDS-DMA with 2 / 3 A stages) on the
shapes of the ConvNeXt-B MLPs and the head convolutions at 128 ROIs: bitwise comparison of the results and event-timed
TFLOP/s (fp32-equivalent, 2MNK).  Run on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
torch.manual_seed(0)
MODES = [("reg", 0, 0), ("glds", 1, 0), ("pipe2", 1, 2), ("pipe3", 1, 3)]
if os.environ.get("MODES"):
    MODES = [m for m in MODES if m[0] in os.environ["MODES"].split(",")]


def set_mode(m):
    hip_lib.set_option("split_gemm_glds", m[1])
    hip_lib.set_option("split_gemm_pipe", m[2])
    hip_lib.set_option("split_gemm_pipe_conv", 1 if m[2] else 0)


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


def row(shape, fl, outs, times, weight):
    r = dict(shape=shape, equal=all(torch.equal(outs[0], o) for o in outs[1:]), weight=weight)
    for m, t in zip(MODES, times):
        r[m[0] + "_ms"] = round(t * 1e3, 4)
        r[m[0] + "_tflops"] = round(fl / t / 1e12, 1)
    return r


res = []
B = int(os.environ.get("ROIS", "128"))
lin = [("s0", B * 4096, 128, 3), ("s1", B * 1024, 256, 3), ("s2", B * 256, 512, 27), ("s3", B * 64, 1024, 3)]
for name, M, Cc, wgt in lin:
    x = torch.randn(M, Cc, device=dev)
    w1 = torch.randn(4 * Cc, Cc, device=dev) * 0.05; b1 = torch.randn(4 * Cc, device=dev)
    w2 = torch.randn(Cc, 4 * Cc, device=dev) * 0.05; b2 = torch.randn(Cc, device=dev)
    gamma = torch.rand(Cc, device=dev); sc = torch.randn(M, Cc, device=dev)
    p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
    hs, ys, t1s, t2s = [], [], [], []
    for m in MODES:
        set_mode(m)
        h = hip_lib.linear_f32_split(x, p1, b1, "gelu")
        y = hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc)
        t1s.append(timeit(lambda: hip_lib.linear_f32_split(x, p1, b1, "gelu")))
        t2s.append(timeit(lambda: hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc)))
        hs.append(h); ys.append(y)
    fl = 2.0 * M * Cc * 4 * Cc
    res.append(row(f"{name} fc1 M={M} K={Cc} N={4*Cc}", fl, hs, t1s, wgt))
    res.append(row(f"{name} fc2 M={M} K={4*Cc} N={Cc}", fl, ys, t2s, wgt))
    del x, w1, w2, sc, hs, ys
for (hw, cin) in [(64, 256), (32, 256), (16, 256)]:
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(256, cin, 3, 3, device=dev) * 0.05
    bc = torch.randn(256, device=dev)
    pk = hip_lib.pack_conv_weight_bf16x3(w)
    ys, ts = [], []
    for m in MODES:
        set_mode(m)
        ys.append(hip_lib.conv3x3_f32_split(x, pk, bc))
        ts.append(timeit(lambda: hip_lib.conv3x3_f32_split(x, pk, bc)))
    res.append(row(f"conv3x3 {hw}x{hw} cin={cin}", 2.0 * B * hw * hw * 256 * cin * 9, ys, ts, 2))
    del x, ys
# general conv (2x2/2 downsample: stays on the kernels of gemm_split.hip in every mode) and odd row counts
x = torch.randn(B, 256, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(512, 256, 2, 2, device=dev) * 0.05
pk = hip_lib.pack_conv_weight_bf16x3(w)
ys = []
for m in MODES:
    set_mode(m); hip_lib.set_option("split_gemm_mi4", 1 if m[1] else -1)
    ys.append(hip_lib.conv2d_f32_split(x, pk, None, 2, 2, 2, 0))
res.append(dict(shape="conv2x2/2 256->512 @32", equal=all(torch.equal(ys[0], y) for y in ys[1:])))
xo = torch.randn(70001, 512, device=dev); wo = torch.randn(2048, 512, device=dev) * 0.05; po = hip_lib.pack_weight_bf16x3(wo)
xc = torch.randn(3, 256, 37, 29, device=dev).contiguous(memory_format=torch.channels_last)
pc = hip_lib.pack_conv_weight_bf16x3(torch.randn(128, 256, 3, 3, device=dev) * 0.05)
ys, yc = [], []
for m in MODES:
    set_mode(m); hip_lib.set_option("split_gemm_mi4", 1 if m[1] else -1)
    ys.append(hip_lib.linear_f32_split(xo, po, None))
    yc.append(hip_lib.conv3x3_f32_split(xc, pc, None))
hip_lib.set_option("split_gemm_mi4", -1)
res.append(dict(shape="linear odd M=70001", equal=all(torch.equal(ys[0], y) for y in ys[1:])))
res.append(dict(shape="conv3x3 odd image 3x37x29", equal=all(torch.equal(yc[0], y) for y in yc[1:])))
hip_lib.set_option("split_gemm_glds", 1); hip_lib.set_option("split_gemm_pipe", 3); hip_lib.set_option("split_gemm_pipe_conv", 0)
for r in res:
    print(json.dumps(r))
print(json.dumps({"step_estimate_" + m[0] + "_ms": round(sum(r.get(m[0] + "_ms", 0) * r.get("weight", 0) for r in res), 3) for m in MODES}))
