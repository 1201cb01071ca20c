#!/bin/bash
O=gpurun_out/r05fin2; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> $O/gpu_tests.txt
S=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/driver.err; E=$(date +%s); echo "python bench.py --gpus 1 --steps 20 --warmup 5: $((E - S)) s wall" >> $O/gpu_tests.txt
cat $O/gpu_tests.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05fin2/bench_driver_flags.json").read().strip().splitlines()[-1])
print("driver flags", round(d["value"], 1), round(d["ms_per_step"], 3), d["config"]["compute_streams"], d["single_stream_mode"], d["parity_in_run"]["max_abs_dR"], d["roofline"]["frac"], d["six_product_mode"]["value"])
PY
