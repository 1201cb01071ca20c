#!/bin/bash
# round 5, call 6: gn_apply cooperative statistics prologue + stem with packed FMAs, same-box A/B against the previous net_kernels.hip
O=gpurun_out/r05f; mkdir -p $O
run() { python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), [(o['kernel'], round(o.get('ms_per_step', o['launch_ms']),3), round(o.get('hbm_frac', o['frac']),3)) for o in d['roofline_other_kernels']])"; }
for i in 1 2 3; do
  GDRNPP_HIP_LIB=ab_libs/old_net/libgdrnpp_hip.so run old
  run new
done 2>&1 | tee $O/ab_net_kernels.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc > /dev/null 2> /tmp/tr.err
find /tmp/tr -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/kernel_stats.csv \;
head -25 $GRAFT_REPO_ROOT/$O/kernel_stats.csv | cut -c1-200
