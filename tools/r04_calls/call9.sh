#!/bin/bash
# Where the fused stage-0 MLP kernel's time goes: ablation builds (timing only) + SQ wait / issue / active buckets.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( for a in default 1 2 4 8 16 3 12; do
    if [ $a = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/mlpf_abl$a/libgdrnpp_hip.so; fi
    OPTS="mlp_fused_pipe=0" B=128 timeout 120 python tools/fused_mlp_time.py 2>&1 | grep -v amdgpu
  done ) > $O/ablation.txt
unset GDRNPP_HIP_LIB
for pipe in 1 0; do
OPTS="mlp_fused_pipe=$pipe" B=128 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1_$pipe -- python tools/fused_mlp_time.py > /dev/null 2> $O/p1_$pipe.err
OPTS="mlp_fused_pipe=$pipe" B=128 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2_$pipe -- python tools/fused_mlp_time.py > /dev/null 2> $O/p2_$pipe.err
echo "pipe=$pipe" >> $O/stalls.txt
python tools/pmc_any_kernel.py $O/p1_$pipe mlp_fused >> $O/stalls.txt
python tools/pmc_any_kernel.py $O/p2_$pipe mlp_fused >> $O/stalls.txt
python tools/pmc_any_kernel.py $O/p1_$pipe gemm_split2 >> $O/stalls.txt
python tools/pmc_any_kernel.py $O/p2_$pipe gemm_split2 >> $O/stalls.txt
rm -rf $O/p1_$pipe $O/p2_$pipe
done
cat $O/ablation.txt $O/stalls.txt
