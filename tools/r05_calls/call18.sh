#!/bin/bash
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trans_type or rot_types or pose_from_pred" 2>&1 | tail -12
