"""Host time per step of the stream workload: how long launch_next (admission, packing, upload, crop + step launches) holds the host, and
how long resolve() waits — with one and with two compute streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

for cs in (1, 2):
    args = B.parse(["--steps", "30", "--no-cpu-baseline", "--no-pmc", "--workload", "stream", "--compute-streams", str(cs)] + sys.argv[1:])
    dev = torch.device("cuda", 0)
    state = B.build_state(args, ["ycbv_convnext_a6"], True, "stream", args.batch or 128, 0, dev, 0)
    launch = state["launch"]
    pend = []
    for i in range(8):
        pend.append(launch(i))
        if len(pend) > cs:
            pend.pop(0)()
    torch.cuda.synchronize()
    tl, tr = [], []
    t_all = time.perf_counter()
    for i in range(args.steps):
        t0 = time.perf_counter(); pend.append(launch(i)); tl.append(time.perf_counter() - t0)
        if len(pend) > cs:
            t0 = time.perf_counter(); pend.pop(0)(); tr.append(time.perf_counter() - t0)
    while pend:
        pend.pop(0)()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / args.steps * 1e3
    import numpy as np
    print(f"compute streams {cs}: {wall:.2f} ms per step wall; launch_next host time mean {np.mean(tl) * 1e3:.2f} ms (max {np.max(tl) * 1e3:.2f}); resolve mean {np.mean(tr) * 1e3:.2f} ms")
    # where the launch time goes: cProfile over a few launches
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for i in range(6):
        pend.append(launch(i))
        if len(pend) > cs:
            pend.pop(0)()
    pr.disable()
    while pend:
        pend.pop(0)()
    torch.cuda.synchronize()
    if cs == 2:
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28); print(st.getvalue()[:6000])
