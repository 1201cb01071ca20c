"""Consecutive steps on ONE HIP stream against the same steps dealt to TWO (engine.StepStreams through bench.py's own launch path): time per
step and the records of EVERY step compared bit for bit — both schedules with the kernel choice of the shared chip
(StepStreams.shared_min_tiles), so that only the scheduling differs.  `--steps 200` is the soak run of profiles/r05last_two_stream_soak.txt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gdrnpp_bop2022_amd import hip_lib

args = B.parse(["--steps", "40", "--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wname = B.resolve_workload(args)
cfg_no, cfg_names, b_default, refine, label = B.WORKLOADS[wname]
b = args.batch or b_default
state = B.build_state(args, cfg_names, refine, wname, b, 0, dev, 0)
launch = state["launch"]
hip_lib.SPLIT2_SHARED_MIN_TILES = hip_lib.SPLIT2_MIN_TILES // 2          # the same kernels on one stream and on two


def run(n, depth):
    pend, out = [], []
    for i in range(n):
        pend.append(launch(i))
        if len(pend) > depth:
            out.append(pend.pop(0)().clone())
    while pend:
        out.append(pend.pop(0)().clone())
    return out


def timed(n_streams, n):
    state["set_compute_streams"](n_streams)
    run(6, n_streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(n, n_streams)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms1, r1 = timed(1, args.steps)
ms2, r2 = timed(2, args.steps)
ms1b, r1b = timed(1, args.steps)
ms2b, r2b = timed(2, args.steps)
def differing(x, y):
    return [i for i, (a, c) in enumerate(zip(x, y)) if not torch.equal(a, c)]
print(f"{wname} batch {b}, {args.steps} steps per run: one stream {ms1:.3f} / {ms1b:.3f} ms per step = {b * 1e3 / ms1:.0f} / {b * 1e3 / ms1b:.0f} ROIs/s | "
      f"two streams {ms2:.3f} / {ms2b:.3f} ms per step = {b * 1e3 / ms2:.0f} / {b * 1e3 / ms2b:.0f} ROIs/s")
print(f"  steps whose records differ: one vs one {len(differing(r1, r1b))}, one vs two {len(differing(r1, r2))} and {len(differing(r1, r2b))}, two vs two {len(differing(r2, r2b))} "
      f"(of {args.steps}); all finite: {all(bool(torch.isfinite(r).all()) for r in r2)}")
