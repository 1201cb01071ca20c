#!/bin/bash
for bsz in 32 64; do echo "== $bsz ROIs"; timeout 300 python tools/shared_rule_diff.py --batch $bsz 2>&1 | grep -E "rule|Error|error" ; done
