#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05w; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --workload lmo_upnp --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child > /dev/null 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/lmo_kernel_stats.csv \;
rm -rf $O/trace
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05w/lmo_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if "anonymous namespace" in n or n.startswith("gdrnpp") or "rocclr" in n: continue
    print(r["Calls"], r["TotalDurationNs"], n[:150])
PY
