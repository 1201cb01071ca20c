#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
for bsz in 128; do timeout 300 python tools/two_stream_steps.py --batch $bsz 2>/dev/null | tail -6; done | tee $O/two_streams.txt
