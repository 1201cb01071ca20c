#!/bin/bash
# round 5, call 9: where does an 8-ROI step go? kernel trace at 8 and 32 ROIs
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 8 32; do
  rm -rf /tmp/tr$b
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child > $O/b$b.json 2> /tmp/tr$b.err
  find /tmp/tr$b -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_b$b.csv \;
  cd $GRAFT_REPO_ROOT && python tools/step_breakdown.py /tmp/tr$b "steady-state step, $b ROIs" > $O/step_breakdown_b$b.md; cd /tmp
  cat $O/b$b.json; head -45 $O/step_breakdown_b$b.md | cut -c1-170
done
