#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30 ) > $O/tests_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -5 ) > $O/smoke.txt
cat $O/tests_gpu.txt $O/smoke.txt
