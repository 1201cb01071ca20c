#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_mlp_fused.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -40 ) > $O/tests_fused.txt
( for b in 128 64 16; do B=$b timeout 120 python tools/fused_mlp_time.py 2>&1 | grep -v amdgpu; done ) > $O/fused_time.txt
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > $O/gpu_tests.txt
for v in "" "--no-fused-mlp" "" "--no-fused-mlp"; do
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_ab.jsonl
done
ls -la $O
