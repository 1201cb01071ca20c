#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04r
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_mlp_fused.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -40 ) > $O/tests_fused.txt
( for o in "mlp_fused_pipe=1" "mlp_fused_pipe=0" "mlp_fused_pipe=1" "mlp_fused_pipe=0"; do C=256 OPTS="$o" B=128 timeout 120 python tools/fused_mlp_time.py 2>&1 | grep -v amdgpu; done ) > $O/fused_time.txt
for v in "" "--fused-mlp-max-c 128" "--opt mlp_fused_pipe=0" "" "--fused-mlp-max-c 128" "--opt mlp_fused_pipe=0"; do
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_ab.jsonl
done
cat $O/tests_fused.txt $O/fused_time.txt
