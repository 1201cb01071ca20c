#!/bin/bash
# round 5, call 10: small batches — from how many 256x128 tiles on does the three-product kernel beat the six-product 128-row / split-K forms?
O=gpurun_out/r05j; mkdir -p $O
for b in 8 16 32 64; do
  for mt in 256 128 64 32 16; do
    GDRNPP_SPLIT2_MIN_TILES=$mt python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-other-mode-line --no-roofline-pass 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b min_tiles $mt', round(d['value'],1), 'ROIs/s', round(d['ms_per_step'],3), 'ms', 'parity dR', d.get('parity_in_run',{}).get('max_abs_dR'), 'reruns', d['range_check']['steps_repeated_with_six_products'])"
  done
done 2>&1 | tee $O/min_tiles.txt
