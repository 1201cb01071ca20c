#!/bin/bash
timeout 400 python tools/stream_host_time.py 2>&1 | grep -v amdgpu.ids | tail -60
