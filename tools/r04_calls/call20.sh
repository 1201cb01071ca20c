#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04u
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
for b in 32 16 8; do
for v in "" "--fused-mlp-min-rows 32768" "--fused-mlp-min-rows 16384" "--no-fused-mlp" "" "--fused-mlp-min-rows 32768" "--fused-mlp-min-rows 16384" "--no-fused-mlp"; do
  ( echo -n "$b $v | "; timeout 300 python bench.py --batch $b --steps 40 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ) >> $O/bench_small.txt
done
done
cat $O/bench_small.txt
