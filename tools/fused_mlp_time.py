"""Time of the fused ConvNeXt MLP (C = 128) at B ROIs; GDRNPP_HIP_LIB selects the build (timing-only dissection variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib as hip
B = int(os.environ.get("B", "128")); dev = "cuda"; torch.manual_seed(0)
m, c = B * 64 * 64, 128
x = torch.randn(m, c, device=dev); res = torch.randn(m, c, device=dev)
w1 = torch.randn(4 * c, c, device=dev) * c ** -0.5; w2 = torch.randn(c, 4 * c, device=dev) * (4 * c) ** -0.5
b1 = torch.randn(4 * c, device=dev); b2 = torch.randn(c, device=dev); g = torch.rand(c, device=dev)
p1, p2 = hip.pack_weight_bf16x3(w1), hip.pack_weight_bf16x3(w2)
fn = lambda: hip.convnext_mlp_f32_split(x, p1, b1, p2, b2, g, res)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print(f"fused MLP M={m}: {t * 1e3:.0f} us  {2.0 * m * 8 * c * c / t / 1e9:.0f} TFLOP/s fp32-equivalent  lib={os.environ.get('GDRNPP_HIP_LIB', 'default')[-40:]}")
