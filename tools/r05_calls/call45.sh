#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams2.py tests/test_gpu_stream.py tests/test_gpu_split2.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python bench.py --steps 20 2> $O/bench.err | tail -1 > $O/bench_refine_b128.json
timeout 300 python bench.py --workload lmo_upnp --steps 30 --warmup 5 --no-pmc 2> $O/lmo.err | tail -1 > $O/bench_lmo_upnp.json
timeout 300 python bench.py --batch 16 --steps 40 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/bench_b16.json
python - <<'PY'
import json
for n in ("bench_refine_b128", "bench_lmo_upnp", "bench_b16"):
    d = json.loads(open(f"gpurun_out/r05v/{n}.json").read())
    print(n, round(d["value"], 1), round(d["ms_per_step"], 3), d.get("single_stream_mode"))
PY
