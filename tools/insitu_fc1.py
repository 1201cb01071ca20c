"""Why is fc1 slower inside the step than alone?  Times fc1 / fc2 of a stage-2 block (M = 25088, C = 512) alone, alternating
(fc1 -> fc2 -> fc1 ... as in the network) and with a memory-bound kernel in front (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib
dev = "cuda"; torch.manual_seed(0)
M, C = int(os.environ.get("M", 25088)), int(os.environ.get("C", 512))
x = torch.randn(M, C, device=dev); r = torch.randn(M, C, device=dev)
w1 = torch.randn(4 * C, C, device=dev) * 0.05; w2 = torch.randn(C, 4 * C, device=dev) * 0.02
b1 = torch.randn(4 * C, device=dev); b2 = torch.randn(C, device=dev); g = torch.rand(C, device=dev)
p1, p2 = hip_lib.pack_weight_bf16x3(w1), hip_lib.pack_weight_bf16x3(w2)
big = torch.randn(64 * 1024 * 1024, device=dev)   # 256 MB: evicts the MALL
def ev(): return torch.cuda.Event(enable_timing=True)
def run(pattern, n=12):
    """pattern: list of ops per iteration; returns mean us per op kind"""
    acc = {}
    for it in range(n + 2):
        h = None
        for op in pattern:
            e0, e1 = ev(), ev()
            if op == "fc1":
                e0.record(); h = hip_lib.linear_f32_split(x, p1, b1, "gelu"); e1.record()
            elif op == "fc2":
                hh = h if h is not None else torch.empty(M, 4 * C, device=dev)
                e0.record(); y = hip_lib.linear_f32_split(hh, p2, b2, "scale_res", g, r); e1.record()
            elif op == "evict":
                e0.record(); big.mul_(1.0); e1.record()
            elif op == "copy":
                e0.record(); x.copy_(r); e1.record()
            if it >= 2: acc.setdefault(op, []).append((e0, e1))
    torch.cuda.synchronize()
    return {k: round(sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e3, 1) for k, v in acc.items()}
print("fc1 alone        ", run(["fc1"]))
print("fc2 alone        ", run(["fc2"]))
print("fc1,fc2          ", run(["fc1", "fc2"]))
print("copy,fc1,fc2     ", run(["copy", "fc1", "fc2"]))
print("evict,fc1,fc2    ", run(["evict", "fc1", "fc2"]))
print("evict,fc1        ", run(["evict", "fc1"]))
print("fc1,evict,fc2    ", run(["fc1", "evict", "fc2"]))
