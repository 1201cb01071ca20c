#!/bin/bash
# SQ / cache counters of the dwconv7x7+LN kernel at the four ConvNeXt-B stage shapes (run on the GPU box).  One counter group
# per rocprofv3 pass (no --stats / trace domains beside --pmc); unknown counter names only lose their own pass.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_dwconv}
mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/microbench_dwconv.py > $O/p$i.out 2> $O/p$i.err || echo "pass $i failed: $grp"
done
cd $R
python tools/pmc_parse_any.py $O dwconv7_ln > $O/summary.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/summary.txt
