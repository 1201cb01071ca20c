#!/bin/bash
timeout 400 python -m pytest tests/test_gpu_crop_resize.py -m gpu -x -q 2>&1 | tail -12
