#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline-pass --no-other-mode-line"
for bsz in 16 32 64; do for mt in 256 128 64; do
  GDRNPP_SPLIT2_MIN_TILES=$mt timeout 200 $B --batch $bsz 2>/dev/null | tail -1 > $O/b${bsz}_mt$mt.json
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05y/b*_mt*.json")):
    d = json.loads(open(f).read()); print(f.split("/")[-1], "%.0f ROIs/s %.3f ms" % (d["value"], d["ms_per_step"]), "reruns", d["range_check"]["steps_repeated_with_six_products"], "parity", (d.get("parity_in_run") or {}).get("max_abs_dR"))
PY
