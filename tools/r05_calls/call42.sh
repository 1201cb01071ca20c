#!/bin/bash
bash tools/r05_evidence.sh r05t > gpurun_out/r05t_evidence.log 2>&1
tail -5 gpurun_out/r05t/gpu_tests.txt; python - <<'PY'
import json
for n in ("bench_refine_b128", "bench_stream", "bench_stream_hostfed", "bench_bop7_stream_hostfed", "bench_lmo_upnp"):
    try:
        d = json.loads(open(f"gpurun_out/r05t/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), round(d["ms_per_step"], 3), (d.get("single_stream_mode") or {}).get("value"), (d.get("parity_in_run") or {}).get("max_abs_dR"))
    except Exception as e:
        print(n, "ERR", e)
PY
cat gpurun_out/r05t/small_batch.md; cat gpurun_out/r05t/bench_configs.txt | tail -6
