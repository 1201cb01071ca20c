"""Workload for rocprofv3 --pmc on the split GEMM: a few launches of the stage-2 ConvNeXt-B MLP shapes at 128 ROIs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
torch.manual_seed(0)
M, c = 128 * 256, 512
x = torch.randn(M, c, device=dev)
w1 = torch.randn(4 * c, c, device=dev) * 0.05; b1 = torch.randn(4 * c, device=dev)
w2 = torch.randn(c, 4 * c, device=dev) * 0.05; b2 = torch.randn(c, device=dev)
gamma = torch.randn(c, device=dev); sc = torch.randn(M, c, device=dev)
p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
for _ in range(4):
    h = hip_lib.linear_f32_split(x, p1, b1, "gelu")
    y = hip_lib.linear_f32_split(h, p2, b2, "scale_res", gamma, sc)
torch.cuda.synchronize()
