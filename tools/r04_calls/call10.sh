#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_mlp_fused.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -25 ) > $O/tests_fused.txt
( for a in default 16 12; do
    if [ $a = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/mlpf_abl$a/libgdrnpp_hip.so; fi
    for o in "mlp_fused_pipe=1" "mlp_fused_pipe=0"; do OPTS="$o" B=128 timeout 120 python tools/fused_mlp_time.py 2>&1 | grep -v amdgpu; done
  done ) > $O/fused_time.txt
unset GDRNPP_HIP_LIB
for v in "" "--opt mlp_fused_pipe=0" "--no-fused-mlp" "" "--opt mlp_fused_pipe=0" "--no-fused-mlp"; do
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line $v 2>> $O/bench.err | tail -1 ) >> $O/bench_ab.jsonl
done
cat $O/tests_fused.txt $O/fused_time.txt
