#!/bin/bash
# Bench lines of every BASELINE.json config that fits one GPU (run on the GPU box): configs[0..4] -> $1 (jsonl).
# configs[3] / [4] are 8-GPU configurations: their per-rank shard (128 ROIs) is what one GPU runs.
out=${1:-gpurun_out/bench_configs.jsonl}
mkdir -p "$(dirname "$out")"
: > "$out"
for w in lmo_upnp rgb refine tless bop7; do
  extra="--no-cpu-baseline --no-pmc"
  [ "$w" = refine ] && extra=""
  python bench.py --workload $w --steps 20 --warmup 3 $extra 2>>"$out.err" | grep '^{' >> "$out"
done
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    r = d.get("roofline") or {}
    print(d["config"]["workload_key"], round(d["value"], 1), "ROIs/s", round(d["ms_per_step"], 2), "ms/step", "roofline", r.get("kernel"), round(r.get("frac") or 0, 3))
PY
