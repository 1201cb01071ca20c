#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_mlp_fused.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -15 ) > $O/tests_fused.txt
( for o in "mlp_fused_waves=4" "mlp_fused_waves=8" "mlp_fused_waves=4" "mlp_fused_waves=8"; do OPTS=$o B=128 timeout 120 python tools/fused_mlp_time.py 2>&1 | grep -v amdgpu; done ) > $O/fused_time.txt
for v in "" "--opt mlp_fused_waves=8" "--no-fused-mlp" "" "--opt mlp_fused_waves=8" "--no-fused-mlp"; do
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_ab.jsonl
done
ls -la $O
