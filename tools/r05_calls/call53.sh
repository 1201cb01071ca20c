#!/bin/bash
O=gpurun_out/r05y2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams2.py tests/test_gpu_stream.py tests/test_gpu_split2.py -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass"
timeout 300 $B 2>$O/b128.err | tail -1 > $O/bench_b128.json
timeout 300 $B --batch 32 --steps 40 2>$O/b32.err | tail -1 > $O/bench_b32.json
timeout 300 $B --workload stream 2>$O/stream.err | tail -1 > $O/bench_stream.json
python - <<'PY'
import json
for n in ("bench_b128", "bench_b32", "bench_stream"):
    try:
        d = json.loads(open(f"gpurun_out/r05y2/{n}.json").read()); print(n, round(d["value"], 1), round(d["ms_per_step"], 3), d.get("single_stream_mode"), (d.get("parity_in_run") or {}).get("max_abs_dR"))
    except Exception as e:
        print(n, "ERR", e)
PY
tail -3 $O/b128.err
