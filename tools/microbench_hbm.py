"""HBM streaming rates on this box: pure write (fill), pure read (sum), copy — 1 GiB buffers (> 256 MB Infinity Cache)."""
import torch
dev = "cuda"
n = 256 * 1024 * 1024
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)


def t(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


gb = n * 4 / 1e9
print(f"fill  {gb / t(lambda: a.fill_(1.0)):.0f} GB/s written")
print(f"sum   {gb / t(lambda: a.sum()):.0f} GB/s read")
print(f"copy  {2 * gb / t(lambda: b.copy_(a)):.0f} GB/s read+written")
print(f"gelu  {2 * gb / t(lambda: torch.nn.functional.gelu(a, approximate='none')):.0f} GB/s read+written (alloc'd out)")
