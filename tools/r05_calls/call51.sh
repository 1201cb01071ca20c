#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline-pass --no-other-mode-line"
for rep in 1 2; do for mt in 256 128; do
  GDRNPP_SPLIT2_MIN_TILES=$mt timeout 200 $B 2>/dev/null | tail -1 > $O/b128_mt${mt}_$rep.json
done; done
GDRNPP_SPLIT2_MIN_TILES=128 timeout 200 $B --batch 8 2>/dev/null | tail -1 > $O/b8_mt128.json
GDRNPP_SPLIT2_MIN_TILES=256 timeout 200 $B --batch 8 2>/dev/null | tail -1 > $O/b8_mt256.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05y/b128_mt*.json") + glob.glob("gpurun_out/r05y/b8_mt*.json")):
    d = json.loads(open(f).read()); print(f.split("/")[-1], "%.0f ROIs/s %.3f ms" % (d["value"], d["ms_per_step"]), "parity", (d.get("parity_in_run") or {}).get("max_abs_dR"))
PY
