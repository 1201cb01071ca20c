#!/bin/bash
timeout 200 python tools/upsample_concurrency.py 2>&1 | grep -v amdgpu
