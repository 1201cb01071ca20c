#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams2.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest_streams.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline-pass"
for cs in 2 1; do
  timeout 300 $B --workload stream --compute-streams $cs 2>/dev/null | tail -1 > $O/bench_stream_cs$cs.json
  timeout 300 $B --workload stream --host-fed --compute-streams $cs 2>/dev/null | tail -1 > $O/bench_stream_hostfed_cs$cs.json
  timeout 400 $B --workload bop7_stream --host-fed --compute-streams $cs 2>/dev/null | tail -1 > $O/bench_bop7_stream_hostfed_cs$cs.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05s/bench_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        print(f, "unreadable", e); continue
    h = d.get("host_fed") or {}
    print(f.split("/")[-1], "streams", d["config"]["compute_streams"], "value %.0f (%.3f ms)" % (d["value"], d["ms_per_step"]), "h2d overlapped", h.get("h2d_overlapped_frac"), "resident pool", h.get("resident_pool_rois_per_s"))
PY
