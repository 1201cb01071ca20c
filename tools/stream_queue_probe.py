"""Which of torch's pool streams execute concurrently with the first one (HIP hardware-queue mapping), and what StepStreams picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd.gdrn_modeling import engine as E
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
ss = [torch.cuda.Stream(dev) for _ in range(12)]
print("overlap ratio of pool stream k with pool stream 0 (2 = concurrent, 1 = same hardware queue):",
      [round(E.streams_overlap_ratio(ss[0], ss[k]), 2) for k in range(1, 12)])
print("with the default stream:", [round(E.streams_overlap_ratio(torch.cuda.default_stream(dev), ss[k]), 2) for k in range(0, 6)])
for _ in range(4):
    d = E.StepStreams(2, dev); print("StepStreams(2): probe (candidates tried, ratio) =", d.overlap_probe)
