#!/bin/bash
O=gpurun_out/r05r; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline-pass"
for cs in 2 3 4; do timeout 300 $B --compute-streams $cs 2>/dev/null | tail -1 > $O/bench_cs$cs.json; done
for bsz in 8 16 32 64 256; do timeout 300 $B --batch $bsz --steps 30 2>/dev/null | tail -1 > $O/bench_b${bsz}_cs2.json; done
timeout 300 $B --workload tless 2>/dev/null | tail -1 > $O/bench_tless_cs2.json
timeout 300 $B --workload lmo_upnp 2>/dev/null | tail -1 > $O/bench_lmo_upnp_cs2.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05r/bench_*cs*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        print(f, "unreadable", e); continue
    s1 = d.get("single_stream_mode") or {}
    print(f.split("/")[-1], "b", d["config"]["rois_per_gpu"], "streams", d["config"]["compute_streams"], "value %.0f (%.3f ms)" % (d["value"], d["ms_per_step"]),
          "single-stream %.0f" % s1.get("value", 0), "gain %.3f" % (d["value"] / s1["value"] if s1.get("value") else 0), "reruns", d["range_check"]["steps_repeated_with_six_products"])
PY
