#!/bin/bash
# Reference-realistic batches (the reference runs ONE image per forward, data_loader.py:901 batch_size=1, i.e. a few to ~30
# ROIs): bench lines at 8 / 16 / 32 / 64 ROIs, eager and replayed from a hipGraph -> $1 (jsonl) + a summary table.
out=${1:-gpurun_out/small_batch.jsonl}
mkdir -p "$(dirname "$out")"
: > "$out"
for b in 8 16 32 64; do
  for g in "" "--graph"; do
    python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-pmc $g 2>>"$out.err" | grep '^{' >> "$out"
  done
done
python - "$out" <<'PY'
import json, sys
print("| ROIs | hipGraph | compute streams | ROIs/s | ms/step | one stream, same run | six-product mode ROIs/s |")
print("|---|---|---|---|---|---|---|")
for l in open(sys.argv[1]):
    d = json.loads(l)
    six = d.get("six_product_mode") or {}
    one = d.get("single_stream_mode") or {}
    print(f"| {d['config']['rois_per_gpu']} | {d['config']['hipgraph']} | {d['config'].get('compute_streams', 1)} | {d['value']:.0f} | {d['ms_per_step']:.3f} | "
          f"{one.get('value', float('nan')):.0f} | {six.get('value', float('nan')):.0f} |")
PY
