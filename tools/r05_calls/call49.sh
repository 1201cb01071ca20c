#!/bin/bash
O=gpurun_out/r05x; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> $O/gpu_tests.txt
S=$(date +%s); timeout 600 python bench.py > $O/bench_default_flags.json 2> $O/default.err; E=$(date +%s); echo "python bench.py (no flags): $((E - S)) s wall" >> $O/gpu_tests.txt
timeout 900 bash tools/bench_configs.sh $O/bench_configs.jsonl > $O/bench_configs.txt 2>&1
cat $O/gpu_tests.txt; tail -6 $O/bench_configs.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05x/bench_default_flags.json").read().strip().splitlines()[-1])
print("default", round(d["value"], 1), round(d["ms_per_step"], 3), d["config"]["compute_streams"], d["single_stream_mode"], d["parity_in_run"]["max_abs_dR"], d["roofline"]["frac"], d["roofline"]["traffic_over_algorithmic"])
PY
