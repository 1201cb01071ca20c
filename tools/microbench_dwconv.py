"""dwconv7x7+LN kernel at the four ConvNeXt-B stage shapes (128 ROIs); GDRNPP_HIP_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
B = int(os.environ.get("B", "128"))
torch.manual_seed(0)
tot = 0.0
for hw, c, nblk in [(64, 128, 3), (32, 256, 3), (16, 512, 27), (8, 1024, 3)]:
    x = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(49, c, device=dev) * 0.1
    b = torch.randn(c, device=dev); g = torch.randn(c, device=dev); be = torch.randn(c, device=dev)
    fn = (lambda: hip_lib.dwconv7x7_ln(x, w, b, g, be, 1e-6)) if os.environ.get("LN", "1") == "1" else (lambda: hip_lib.dwconv7x7_ln(x, w, b))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    by = 2 * x.numel() * 4
    print(f"{hw}x{hw} C={c}: {t * 1e3:.1f} us  {by / t / 1e6:.0f} GB/s  {49 * 2 * x.numel() / t / 1e9:.1f} TFLOP/s")
    tot += nblk * t
print(f"total per forward {tot:.3f} ms  lib={os.environ.get('GDRNPP_HIP_LIB', 'default')}")
