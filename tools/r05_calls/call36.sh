#!/bin/bash
O=gpurun_out/r05q; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-mode-line"
for i in 1 2; do
  timeout 200 $B 2>/dev/null | tail -1 > $O/bench_default_$i.json
  GDRNPP_HIP_LIB=ab_libs/noslp/libgdrnpp_hip.so timeout 200 $B 2>/dev/null | tail -1 > $O/bench_noslp_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05q/bench_*.json")):
    d = json.loads(open(f).read()); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
for bsz in 128 64; do GDRNPP_HIP_LIB=ab_libs/noslp/libgdrnpp_hip.so timeout 300 python tools/two_stream_steps.py --batch $bsz 2>/dev/null | tail -5; done | tee $O/two_streams_noslp.txt
