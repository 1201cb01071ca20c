#!/bin/bash
O=gpurun_out/r05fin3; mkdir -p $O
for i in 1 2; do timeout 100 python bench.py --workload stream --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass 2>/dev/null | tail -1 > $O/stream_d2h_prio_$i.json; done
python - <<'PY'
import json
for i in (1, 2):
    d = json.loads(open(f"gpurun_out/r05fin3/stream_d2h_prio_{i}.json").read()); print("stream", round(d["value"], 1), round(d["ms_per_step"], 3), d["config"]["compute_stream_overlap_probe"])
PY
