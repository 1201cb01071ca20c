#!/bin/bash
O=gpurun_out/r05final; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> $O/gpu_tests.txt
cat $O/gpu_tests.txt
