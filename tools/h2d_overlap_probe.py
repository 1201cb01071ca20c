"""Does a pinned host-to-device copy on a side stream run UNDER kernels of the current stream?  Default-priority vs high-priority
side stream (HIP multiplexes streams over a few hardware queues).  Prints the share of the copy's device time that lies inside the
busy interval of a ~20 ms kernel train on the current stream.  Run on the GPU box."""
import torch

dev = torch.device("cuda", 0)
x = torch.randn(8192, 8192, device=dev)
host = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()          # 64 MiB
torch.cuda.synchronize()


def probe(stream):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(2):                      # second pass is the measured one (first touches allocators / code objects)
        ev[0].record()
        y = x
        for _ in range(12):
            y = y @ x                       # ~12 x 1.1 TFLOP fp32: tens of ms on the current stream
        ev[1].record()
        with torch.cuda.stream(stream):
            ev[2].record()
            d = host.to(dev, non_blocking=True)
            ev[3].record()
        d.record_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
    t = [ev[0].elapsed_time(e) for e in ev]
    busy = (t[0], t[1])
    lo, hi = max(t[2], busy[0]), min(t[3], busy[1])
    return dict(kernels_ms=t[1] - t[0], copy_start_ms=t[2], copy_ms=t[3] - t[2], copy_gbs=host.numel() / 1e9 / ((t[3] - t[2]) * 1e-3),
                overlapped_frac=max(0.0, hi - lo) / (t[3] - t[2]))


print("default-priority side stream:", probe(torch.cuda.Stream(device=dev)))
print("high-priority side stream   :", probe(torch.cuda.Stream(device=dev, priority=-1)))
for i in range(3):
    print(f"another default stream #{i}  :", probe(torch.cuda.Stream(device=dev)))
