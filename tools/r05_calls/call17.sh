#!/bin/bash
timeout 400 python -m pytest tests/test_gpu_evaluator.py tests/test_gpu_epnp.py tests/test_gpu_parity.py -m gpu -x -q -k "pnp or yolox" 2>&1 | tail -12
