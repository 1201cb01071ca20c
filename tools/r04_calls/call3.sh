#!/bin/bash
# Round 4, GPU call 3: full GPU suite (P3P on the device, small GroupNorm, packer), range-check slot placement A/B, profile set.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > $O/gpu_tests.txt
for v in default rc_b rc_c rc_d norange default rc_b rc_c rc_d norange; do
  if [ $v = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/$v/libgdrnpp_hip.so; fi
  echo "== $v" >> $O/mlp_shapes_rc.txt
  ( X3=1 timeout 200 python tools/mlp_shapes.py 2>&1 | grep -v amdgpu.ids ) >> $O/mlp_shapes_rc.txt
done
unset GDRNPP_HIP_LIB
bash tools/profile_bench.sh r04c > $O/profile.log 2>&1
ls -la $O $R/gpurun_out/prof_r04c
