#!/bin/bash
timeout 250 python tools/two_stream_diag5.py --batch 128 2>&1 | grep -E "upsample|chain|Error|error" | tail -12
