#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( for a in default 16 32 64 96; do
    if [ $a = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/mlpf_abl$a/libgdrnpp_hip.so; fi
    for o in "mlp_fused_pipe=1"; do OPTS="$o" B=128 timeout 120 python tools/fused_mlp_time.py 2>&1 | grep -v amdgpu; done
  done ) > $O/fused_time.txt
unset GDRNPP_HIP_LIB
for pipe in 1 0; do
OPTS="mlp_fused_pipe=$pipe" B=128 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1_$pipe -- python tools/fused_mlp_time.py > /dev/null 2> $O/p1_$pipe.err
OPTS="mlp_fused_pipe=$pipe" B=128 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_LEVEL_WAVES --kernel-trace --output-format csv -d $O/p2_$pipe -- python tools/fused_mlp_time.py > /dev/null 2> $O/p2_$pipe.err
echo "pipe=$pipe" >> $O/stalls.txt
python tools/pmc_any_kernel.py $O/p1_$pipe mlp_fused_x3 >> $O/stalls.txt
python tools/pmc_any_kernel.py $O/p2_$pipe mlp_fused_x3 >> $O/stalls.txt
rm -rf $O/p1_$pipe $O/p2_$pipe
done
cat $O/fused_time.txt $O/stalls.txt
