#!/bin/bash
for i in 1 2; do for w in tless refine; do python bench.py --workload $w --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line --no-roofline-pass 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms'])"; done; done
