#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05w; mkdir -p $O; cd $R
for bsz in 8 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b$bsz -- python bench.py --batch $bsz --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child --compute-streams 1 > /dev/null 2> $O/trace_b$bsz.err
  python tools/step_breakdown.py $O/trace_b$bsz "steady-state step, YCB-V convnext_a6 + refine, $bsz ROIs" > $O/step_breakdown_b$bsz.md
  rm -rf $O/trace_b$bsz
  echo "== b=$bsz"; head -4 $O/step_breakdown_b$bsz.md; grep -v "anonymous namespace\|gdrnpp::" $O/step_breakdown_b$bsz.md | grep "^| [0-9]" | cut -c1-160
done
timeout 300 python bench.py --workload lmo_upnp --steps 30 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/bench_lmo_upnp_default.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05w/bench_lmo_upnp_default.json").read()); print("lmo_upnp default:", round(d["value"], 1), d["ms_per_step"], "streams", d["config"]["compute_streams"], d.get("single_stream_mode"))
PY
