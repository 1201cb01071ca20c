#!/bin/bash
O=gpurun_out/r05soak; mkdir -p $O
( timeout 200 python tools/two_stream_steps.py --steps 200 2>/dev/null | tail -2
  timeout 200 python tools/two_stream_steps.py --steps 300 --batch 32 2>/dev/null | tail -2
  timeout 200 python tools/two_stream_steps.py --steps 200 --workload tless 2>/dev/null | tail -2 ) | tee $O/two_stream_soak.txt
