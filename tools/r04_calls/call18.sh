#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
export GDRNPP_HIP_LIB=$R/_ab/pairbar/libgdrnpp_hip.so
( timeout 900 python -m pytest tests/test_gpu_split2.py tests/test_gpu_f16x2_rows.py tests/test_gpu_headline_shapes.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/tests_pair.txt
unset GDRNPP_HIP_LIB
for v in pair base pair base; do
  if [ $v = pair ]; then export GDRNPP_HIP_LIB=$R/_ab/pairbar/libgdrnpp_hip.so; else unset GDRNPP_HIP_LIB; fi
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line 2>> $O/bench.err | tail -1 ) >> $O/bench_ab.jsonl
done
cat $O/tests_pair.txt
