#!/bin/bash
# final commit check: what the driver runs — pytest -m gpu, smoke(), default bench.py
O=gpurun_out/r05final; mkdir -p $O
S=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench.py wall: $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05final/bench_default.json') if l.startswith('{')][-1])
print(round(d['value'],1), round(d['ms_per_step'],3), d['steps'], d['warmup'], d['parity_in_run']['max_abs_dR'], d['roofline']['frac'], d['roofline']['traffic'], d.get('pmc_live_seconds'), d['cpu_baseline']['value'])
PY
