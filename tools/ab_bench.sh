#!/bin/bash
# same-box A/B of bench.py variants: tools/ab_bench.sh "<flags A>" "<flags B>" [rounds]
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    python bench.py --steps 20 --no-cpu-baseline --no-roofline-pass $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '|', round(d['value'],1), 'ROIs/s', round(d['ms_per_step'],3), 'ms')"
  done
done
