#!/bin/bash
timeout 250 python tools/two_stream_diag6.py --batch 128 2>&1 | grep -E "upsample|Error|error" | tail -12
