#!/bin/bash
timeout 300 python tools/step_host_profile.py --batch 8 2>&1 | grep -v amdgpu.ids | tail -50
