#!/bin/bash
timeout 300 python tools/two_stream_diag7.py --batch 128 2>&1 | grep -vE "amdgpu.ids|^$" | tail -60
