#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04v
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
for v in pair single pair single pair single; do
  if [ $v = single ]; then export GDRNPP_HIP_LIB=$R/_ab/singlebar/libgdrnpp_hip.so; else unset GDRNPP_HIP_LIB; fi
  ( echo -n "$v | "; timeout 300 python bench.py --workload lmo_upnp --steps 100 --no-cpu-baseline --no-other-mode-line 2>> $O/bench.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ) >> $O/lmo.txt
done
cat $O/lmo.txt
