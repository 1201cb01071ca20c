#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04t
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
for v in "" "--fused-mlp-max-c 128" "--no-fused-mlp" "" "--fused-mlp-max-c 128" "--no-fused-mlp"; do
  ( timeout 300 python bench.py --batch 64 --steps 30 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_b64.jsonl
done
for v in "" "--fused-mlp-max-c 128" "" "--fused-mlp-max-c 128"; do
  ( timeout 300 python bench.py --batch 96 --steps 30 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_b96.jsonl
done
