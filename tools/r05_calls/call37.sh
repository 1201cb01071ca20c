#!/bin/bash
O=gpurun_out/r05r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_streams2.py tests/test_gpu_net_kernels.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -8 | tee $O/pytest_streams2.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc 2>$O/bench.err | tail -1 > $O/bench_two_streams.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05r/bench_two_streams.json").read())
print("headline", d["value"], d["ms_per_step"], d["config"]["compute_streams"])
print("single", d.get("single_stream_mode")); print("six", {k: d["six_product_mode"].get(k) for k in ("value", "ms_per_step", "error")})
print("parity", d.get("parity_in_run", {}).get("max_abs_dR"))
PY
tail -3 $O/bench.err
