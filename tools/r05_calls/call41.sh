#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline-pass --no-other-mode-line"
for i in 1 2 3; do for cs in 1 2; do
  timeout 300 $B --workload stream --compute-streams $cs 2>/dev/null | tail -1 > $O/rep_stream_cs${cs}_$i.json
done; done
for i in 1 2; do for cs in 1 2; do
  timeout 300 $B --compute-streams $cs 2>/dev/null | tail -1 > $O/rep_step_cs${cs}_$i.json
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05s/rep_*.json")):
    d = json.loads(open(f).read()); print(f.split("/")[-1], "%.0f ROIs/s %.3f ms" % (d["value"], d["ms_per_step"]))
PY
