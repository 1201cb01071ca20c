#!/bin/bash
# Round 4, GPU call 2: the fixed range test, the timing-only A-direct k-loop against the product kernel (isolated MLP shapes and
# in situ), the extended VALU probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_split2.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > $O/gpu_tests_split2.txt
( timeout 60 _ab/valu_rate_probe 2>&1 ) > $O/valu_rate_probe.txt
for v in default adirect default adirect; do
  if [ $v = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/$v/libgdrnpp_hip.so; fi
  echo "== $v" >> $O/mlp_shapes_adirect.txt
  ( X3=1 timeout 200 python tools/mlp_shapes.py 2>&1 | grep -v amdgpu.ids ) >> $O/mlp_shapes_adirect.txt
done
for v in default adirect default adirect; do
  if [ $v = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/$v/libgdrnpp_hip.so; fi
  ( timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-other-mode-line 2>> $O/bench_$v.err | tail -1 ) >> $O/bench_$v.jsonl
done
unset GDRNPP_HIP_LIB
ls -la $O
