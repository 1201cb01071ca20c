#!/bin/bash
timeout 400 python -m pytest tests/test_gpu_net_golden.py tests/test_gpu_configs.py -m gpu -x -q -k "resnet34 or online_xyz" 2>&1 | tail -8
