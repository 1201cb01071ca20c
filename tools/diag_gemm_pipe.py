"""Localise mismatches of the software-pipelined split GEMM against the register-staged kernel (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnpp_bop2022_amd import hip_lib

dev = "cuda"
hip_lib.set_option("split_gemm_mi4", 1)
bad = 0
for (m, k, n) in [(512, 32, 128), (512, 64, 128), (512, 96, 256), (1061, 512, 256), (256, 2048, 128)]:
    torch.manual_seed(1)
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * k ** -0.5
    pk = hip_lib.pack_weight_bf16x3(w)
    hip_lib.set_option("split_gemm_glds", 0); hip_lib.set_option("split_gemm_pipe", 0)
    ref = hip_lib.linear_f32_split(x, pk, None, "none")
    ref64 = (x.double() @ w.double().T)
    for pipe in (2, 3):
        hip_lib.set_option("split_gemm_glds", 1); hip_lib.set_option("split_gemm_pipe", pipe)
        y = hip_lib.linear_f32_split(x, pk, None, "none")
        torch.cuda.synchronize()
        ne = (y != ref)
        e64 = ((y.double() - ref64).abs().max() / ref64.abs().max()).item()
        print(f"M={m} K={k} N={n} pipe{pipe}: mismatches {int(ne.sum())}/{ne.numel()} maxdiff {(y-ref).abs().max().item():.3e} err_vs_fp64 {e64:.2e}")
        if ne.any():
            bad += 1
            rows = ne.any(1).nonzero().flatten(); cols = ne.any(0).nonzero().flatten()
            print("   rows%256 hist(32):", torch.bincount((rows % 256) // 32, minlength=8).tolist(),
                  " cols%128 hist(32):", torch.bincount((cols % 128) // 32, minlength=4).tolist())
hip_lib.set_option("split_gemm_mi4", -1); hip_lib.set_option("split_gemm_pipe", 3)
print("DIAG_OK" if not bad else "DIAG_BAD")
