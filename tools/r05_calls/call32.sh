#!/bin/bash
for v in ab_libs/D; do
  echo "== $v"; GDRNPP_HIP_LIB=$v/libgdrnpp_hip.so timeout 250 python tools/two_stream_diag6.py --batch 128 2>&1 | grep -E "upsample beside conv|Error|error" | tail -4
done
