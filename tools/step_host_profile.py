"""Where the host time of launching one step goes (cProfile by own time), at a small batch where the launch rate is the bound."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

args = B.parse(["--steps", "30", "--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
state = B.build_state(args, ["ycbv_convnext_a6"], True, "refine", args.batch or 8, 0, dev, 0)
launch = state["launch"]
pend = []
def go(n):
    for i in range(n):
        pend.append(launch(i))
        if len(pend) > 2:
            pend.pop(0)()
go(10); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); go(40); pr.disable()
while pend: pend.pop(0)()
torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32); print(st.getvalue()[:7000])
