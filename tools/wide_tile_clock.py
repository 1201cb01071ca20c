"""Stage-2 MLP GEMMs with 256x128 (two workgroups per CU) and 256x256 (one, option split2_wide) block tiles: run under
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace to see time, effective clock and matrix-pipe busy of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib as hip
torch.manual_seed(0)
N_REP = int(os.environ.get("REP", "60"))
for (m, k, n, epi) in [(32768, 512, 2048, "gelu"), (32768, 2048, 512, "scale_res")]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5; b = torch.randn(n, device="cuda")
    g = torch.randn(n, device="cuda") if epi == "scale_res" else None
    r = torch.randn(m, n, device="cuda") if epi == "scale_res" else None
    pk = hip.pack_weight_f16x2(w)
    for wide in (0, 1):
        hip.set_option("split2_wide", wide)
        for _ in range(N_REP):
            hip.linear_f32_split(x, pk, b, epi, g, r)
        torch.cuda.synchronize()
hip.set_option("split2_wide", 0)
