#!/bin/bash
timeout 400 python tools/two_stream_diag8.py --batch 128 2>&1 | grep -E "beside|----|Error|error|Traceback" | tail -30
