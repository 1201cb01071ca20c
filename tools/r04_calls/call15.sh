#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
/opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --showmaxpower --showperflevel > $O/smi_idle.txt 2>&1
/opt/rocm/bin/rocm-smi -a > $O/smi_all.txt 2>&1
( for i in $(seq 1 300); do echo "== $(date +%s.%N)"; timeout 5 /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|busy" ; sleep 0.2; done ) > $O/smi_during_bench.txt &
SMI=$!
( timeout 400 python bench.py --steps 200 --no-cpu-baseline --no-other-mode-line 2> $O/bench.err | tail -1 ) > $O/bench.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
cat $O/smi_idle.txt; grep -c "==" $O/smi_during_bench.txt; grep -i "power\|sclk" $O/smi_during_bench.txt | sort | uniq -c | sort -rn | head -40
