"""Experiment: consecutive 128-ROI steps issued alternately on TWO HIP streams (two steps in flight on the device, each a complete,
independent pass over its own batch) against the single-stream schedule of bench.py.  Same model, same two batches, same records."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

args = B.parse(["--steps", "40", "--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wname = B.resolve_workload(args)
cfg_no, cfg_names, b_default, refine, label = B.WORKLOADS[wname]
b = args.batch or b_default
state = B.build_state(args, cfg_names, refine, wname, b, 0, dev, 0)
launch = state["launch"]


def single(n):
    prev, out = None, []
    for i in range(n):
        cur = launch(i)
        if prev is not None:
            out.append(prev())
        prev = cur
    out.append(prev())
    return out


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def dual(n, depth=2):
    pend, out = [], []
    for i in range(n):
        s = streams[i % 2]
        with torch.cuda.stream(s):
            pend.append((s, launch(i)))
        if len(pend) > depth:
            s0, h = pend.pop(0)
            with torch.cuda.stream(s0):
                out.append(h())
    for s0, h in pend:
        with torch.cuda.stream(s0):
            out.append(h())
    return out


def timed(fn, n):
    fn(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for s in streams:
    s.wait_stream(torch.cuda.current_stream())
ms1, r1 = timed(single, args.steps)
ms2, r2 = timed(dual, args.steps)
ms1b, _ = timed(single, args.steps)
ms2b, r2b = timed(dual, args.steps)
same = all(torch.equal(a, c) for a, c in zip(r1, r2)) and all(torch.equal(a, c) for a, c in zip(r1, r2b))
def diff(x, y):
    d = [(a - c).abs() for a, c in zip(x, y)]
    return dict(dR=max(float(v[:, :9].max()) for v in d), dt=max(float(v[:, 9:12].max()) for v in d), rest=max(float(v[:, 12:].max()) for v in d),
                steps_differing=sum(int(not torch.equal(a, c)) for a, c in zip(x, y)), first=[i for i, (a, c) in enumerate(zip(x, y)) if not torch.equal(a, c)][:6])
_, r1c = timed(single, args.steps)
print("single vs single:", diff(r1, r1c))
print("single vs dual  :", diff(r1, r2))
print("dual vs dual    :", diff(r2, r2b))
print("step 0 vs step 2 (same batch) single:", diff(r1[0:1], r1[2:3]), "dual:", diff(r2[0:1], r2[2:3]))
print(f"batch {b}: single stream {ms1:.3f} / {ms1b:.3f} ms per step = {b * 1e3 / ms1:.0f} / {b * 1e3 / ms1b:.0f} ROIs/s | two streams {ms2:.3f} / {ms2b:.3f} ms per step = "
      f"{b * 1e3 / ms2:.0f} / {b * 1e3 / ms2b:.0f} ROIs/s | records bit-equal: {same}")
