#!/bin/bash
O=gpurun_out/r05fin2; mkdir -p $O
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass 2>$O/e1.err | tail -1 > $O/bench_probe_refine.json
timeout 200 python bench.py --workload stream --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass 2>$O/e2.err | tail -1 > $O/bench_probe_stream.json
python - <<'PY'
import json
for n in ("bench_probe_refine", "bench_probe_stream"):
    try:
        d = json.loads(open(f"gpurun_out/r05fin2/{n}.json").read()); print(n, round(d["value"], 1), d["config"]["compute_streams"], d["config"]["compute_stream_overlap_probe"])
    except Exception as e:
        print(n, "ERR", e)
PY
tail -2 $O/e1.err $O/e2.err
