#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_f16x2_rows.py tests/test_gpu_mlp_fused.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -40 ) > $O/tests_rows.txt
for v in "" "--no-f16x2-rows" "" "--no-f16x2-rows"; do
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_ab.jsonl
done
cat $O/tests_rows.txt
