"""fc1 / fc2 of a real stage-2 ConvNeXt block: per-launch time inside the block loop vs the same launches replayed in isolation on
the very same tensors and weights (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib
from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer
torch.manual_seed(0)
cfg = get_cfg("ycbv_convnext_a6", ["TEST.USE_DEPTH_REFINE=True"])
model, _ = build_model_optimizer(cfg, is_test=True)
bb = model.backbone
x = torch.randn(128, 512, 14, 14, device="cuda").contiguous(memory_format=torch.channels_last)
blocks = list(bb.stages_2.blocks)[:8]
def timed_blocks():
    t = hip_lib.LaunchTimer(); hip_lib.set_launch_timer(t)
    y = x
    with torch.no_grad():
        for b in blocks: y = b(y)
    hip_lib.set_launch_timer(None); torch.cuda.synchronize()
    return [round(r[2].elapsed_time(r[3]) * 1e3) for r in t.records]
timed_blocks(); timed_blocks()
print("in-block (fc1, fc2 per block, us):", timed_blocks())
# isolated replay with block 3's tensors
b = blocks[3]
with torch.no_grad():
    xin = hip_layers.dwconv_ln(b.conv_dw, b.norm, x, b._cache)
    m = xin.numel() // 512
    p1 = hip_layers._packed(b.mlp.fc1, b._cache, "fc1_pk"); p2 = hip_layers._packed(b.mlp.fc2, b._cache, "fc2_pk")
    x2 = xin.reshape(m, 512)
    h = hip_lib.linear_f32_split(x2, p1, b.mlp.fc1.bias, "gelu")
    sc = x.permute(0, 2, 3, 1).reshape(m, 512)
def ev(): return torch.cuda.Event(enable_timing=True)
def rep(fn, n=10):
    for _ in range(2): fn()
    es = []
    for _ in range(n):
        a, c = ev(), ev(); a.record(); fn(); c.record(); es.append((a, c))
    torch.cuda.synchronize()
    return round(sum(a.elapsed_time(c) for a, c in es) / n * 1e3)
with torch.no_grad():
    print("isolated, block tensors: fc1", rep(lambda: hip_lib.linear_f32_split(x2, p1, b.mlp.fc1.bias, "gelu")),
          "fc2", rep(lambda: hip_lib.linear_f32_split(h, p2, b.mlp.fc2.bias, "scale_res", b.gamma, sc)))
    xr = torch.randn_like(x2); w = torch.randn(2048, 512, device="cuda") * 0.05; pr = hip_lib.pack_weight_bf16x3(w); br = torch.randn(2048, device="cuda")
    print("isolated, randn x + block weights: fc1", rep(lambda: hip_lib.linear_f32_split(xr, p1, b.mlp.fc1.bias, "gelu")))
    print("isolated, block x + randn weights/bias: fc1", rep(lambda: hip_lib.linear_f32_split(x2, pr, br, "gelu")))
    print("isolated, randn x + randn weights/bias: fc1", rep(lambda: hip_lib.linear_f32_split(xr, pr, br, "gelu")))
    print("isolated, block x + block weights, epilogue none: fc1", rep(lambda: hip_lib.linear_f32_split(x2, p1, b.mlp.fc1.bias, "none")))
    print("isolated, randn x + randn weights, epilogue none: fc1", rep(lambda: hip_lib.linear_f32_split(xr, pr, br, "none")))
