#!/bin/bash
# Diagnostic kernels of tools/pk_hazard_probe.py / tools/two_stream_diag8.py (not part of the product).
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
  pk_probe.hip pk_probe2.hip companions.hip -o libpk_probe.so
