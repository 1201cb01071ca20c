#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04l
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
OPTS="mlp_fused_pipe=1" B=128 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -- python tools/fused_mlp_time.py > $O/p.out 2> $O/p.err
python tools/pmc_clock.py $O/p mlp_fused_x3 > $O/clock.txt 2>&1
python tools/pmc_clock.py $O/p gemm_split2 >> $O/clock.txt 2>&1
head -3 $(find $O/p -name "*counter_collection.csv" | head -1) >> $O/clock.txt
rm -rf $O/p
cat $O/clock.txt $O/p.out
