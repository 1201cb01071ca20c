#!/bin/bash
O=gpurun_out/r05raw; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_streams2.py tests/test_gpu_stream.py tests/test_gpu_split2.py tests/test_gpu_net_kernels.py -q -m gpu 2>&1 | tail -2
timeout 300 python tools/stream_host_time.py 2>&1 | grep "compute streams"
timeout 900 bash tools/small_batch_lines.sh $O/small_batch.jsonl > $O/small_batch.md 2>&1; cat $O/small_batch.md
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass 2>/dev/null | tail -1 > $O/bench_b128.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05raw/bench_b128.json").read()); print("b128", round(d["value"], 1), d["ms_per_step"], d["single_stream_mode"]["value"], d["single_stream_mode"]["last_step_records_bit_equal_to_timed_region"])
PY
