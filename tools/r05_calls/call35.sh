#!/bin/bash
mkdir -p gpurun_out/r05p
timeout 300 python tools/pk_hazard_probe.py 2>&1 | grep -vE "amdgpu.ids" > gpurun_out/r05p/pk_hazard_probe.txt
cat gpurun_out/r05p/pk_hazard_probe.txt
