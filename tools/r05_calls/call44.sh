#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline-pass --no-other-mode-line"
for rep in 1 2; do
for pr in "0,0" "-1,0" "-1,-1"; do
  timeout 300 $B --stream-priorities=$pr 2>/dev/null | tail -1 > $O/prio_${pr}_$rep.json
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05v/prio_*.json")):
    d = json.loads(open(f).read()); print(f.split("/")[-1], "%.0f ROIs/s %.3f ms" % (d["value"], d["ms_per_step"]))
PY
