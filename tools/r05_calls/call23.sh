#!/bin/bash
echo "== round-4 net_kernels.hip (one-output upsample)"; GDRNPP_HIP_LIB=ab_libs/old_net/libgdrnpp_hip.so timeout 300 python tools/two_stream_diag2.py --batch 128 2>&1 | grep "max diff" | sed -n 5,7p
echo "== current"; timeout 300 python tools/two_stream_diag2.py --batch 128 2>&1 | grep "max diff" | sed -n 5,7p
