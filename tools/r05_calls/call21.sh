#!/bin/bash
# large shards on one GPU (288 GB HBM): 256 / 512 / 1024 ROIs per step, incl. configs[3]'s whole 1024-ROI T-LESS iteration
O=gpurun_out/r05m; mkdir -p $O
: > $O/large_shards.jsonl
for w in "refine 256" "refine 512" "tless 1024"; do set -- $w
  timeout 400 python bench.py --workload $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-other-mode-line 2>>$O/err.txt | grep '^{' | tee -a $O/large_shards.jsonl | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 batch $2', round(d['value'],1), 'ROIs/s', round(d['ms_per_step'],2), 'ms', 'parity', d['parity_in_run'].get('max_abs_dR'), 'reruns', d['range_check'], 'frac', round(d['roofline']['frac'],3), 'peak mem GB', None)"
done
python -c "import torch; print('device mem', torch.cuda.mem_get_info())"
tail -3 $O/err.txt
