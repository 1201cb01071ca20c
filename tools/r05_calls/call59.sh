#!/bin/bash
O=gpurun_out/r05soak; mkdir -p $O
( timeout 100 python tools/stream_queue_probe.py 2>&1 | grep -v amdgpu.ids
  timeout 200 python tools/two_stream_steps.py --steps 60 2>/dev/null | tail -2 ) | tee $O/stream_queue_probe.txt
