#!/bin/bash
timeout 200 python tools/two_stream_diag4.py --batch 128 2>&1 | grep -E "serial|rep "
