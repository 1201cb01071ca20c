"""Stage-0 ConvNeXt MLP block (C = 128, hidden 512) at B ROIs: the fused three-product kernel against the two three-product launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib as hip

B = int(os.environ.get("B", "128"))
for o in os.environ.get("OPTS", "").split():
    hip.set_option(o.split("=")[0], int(o.split("=")[1]))
c = int(os.environ.get("C", "128"))
m = B * 64 * 64 * 128 // c * 128 // c        # stage 0: 64 x 64 pixels per ROI at C = 128, stage 1: 32 x 32 at C = 256
torch.manual_seed(0)
x = torch.randn(m, c, device="cuda"); res = torch.randn(m, c, device="cuda")
w1 = torch.randn(4 * c, c, device="cuda") * c ** -0.5; w2 = torch.randn(c, 4 * c, device="cuda") * (4 * c) ** -0.5
b1 = torch.randn(4 * c, device="cuda"); b2 = torch.randn(c, device="cuda"); g = torch.randn(c, device="cuda")
pkf = hip.pack_mlp_fused_f16x2(w1, w2); pk1 = hip.pack_weight_f16x2(w1); pk2 = hip.pack_weight_f16x2(w2)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fused = t(lambda: hip.convnext_mlp_f32_fused(x, pkf, b1, b2, g, res))
two = t(lambda: hip.linear_f32_split(hip.linear_f32_split(x, pk1, b1, "gelu"), pk2, b2, "scale_res", g, res))
fl = 4.0 * m * c * 4 * c
print(f"B={B} M={m} C={c}: fused {fused:.0f} us ({fl / fused / 1e6:.0f} TFLOP/s fp32-equivalent, {12.0 * m * c / fused / 1e3:.0f} GB/s algorithmic)   two launches {two:.0f} us   lib={os.environ.get('GDRNPP_HIP_LIB', 'default')} opts={os.environ.get('OPTS', '')}")
