#!/bin/bash
for v in ab_libs/C ""; do
  if [ -n "$v" ]; then export GDRNPP_HIP_LIB=$v/libgdrnpp_hip.so; else unset GDRNPP_HIP_LIB; fi
  timeout 200 python tools/two_stream_diag3.py --batch 128 2>&1 | grep -E "lib:|tail (hip|sync)"
done
