#!/bin/bash
# Where the waves of the three- and six-product GEMM kernels spend their cycles (stage-2 fc2 / fc1 shapes of 128 ROIs):
# SQ wait / issue / active buckets per kernel (quad-cycles; MI355X_MICROARCH.md "rocprofv3 PMC slots").  Own pass, kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_split2
rm -rf $O; mkdir -p $O
cd $R
cat > /tmp/_s2.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from gdrnpp_bop2022_amd import hip_lib as hip
torch.manual_seed(0)
for (m, k, n, epi) in [(32768, 2048, 512, "scale_res"), (32768, 512, 2048, "gelu")]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5; b = torch.randn(n, device="cuda")
    g = torch.randn(n, device="cuda") if epi == "scale_res" else None
    r = torch.randn(m, n, device="cuda") if epi == "scale_res" else None
    for pk in (hip.pack_weight_f16x2(w), hip.pack_weight_bf16x3(w)):
        for _ in range(3):
            hip.linear_f32_split(x, pk, b, epi, g, r)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1 -- python /tmp/_s2.py > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- python /tmp/_s2.py > /dev/null 2> $O/p2.err
python tools/pmc_any_kernel.py $O/p1 gemm_split > $O/stalls.txt
python tools/pmc_any_kernel.py $O/p2 gemm_split >> $O/stalls.txt
rm -rf $O/p1 $O/p2
cat $O/stalls.txt; tail -3 $O/p1.err $O/p2.err
