#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass --no-other-mode-line"
for bsz in 8 16 32 64 128; do for rule in "0 4096" "128 4096" "128 8192"; do
  set -- $rule
  GDRNPP_SPLIT2_SHARED_MIN_TILES=$1 GDRNPP_SPLIT2_SHARED_MIN_ROWS=$2 timeout 200 $B --batch $bsz 2>/dev/null | tail -1 > $O/rule_b${bsz}_t$1_r$2.json
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05y/rule_*.json")):
    d = json.loads(open(f).read()); print(f.split("/")[-1], "%.0f ROIs/s %.3f ms" % (d["value"], d["ms_per_step"]))
PY
