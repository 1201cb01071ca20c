#!/bin/bash
# Evidence set of a code state from ONE box: GPU tests, smoke, headline bench line, stream line, every BASELINE config, the
# reference's own batch sizes, rocprofv3 kernel stats / step breakdown / PMC traffic / matrix-pipe busy.  usage: r04_evidence.sh <tag>
tag=${1:-r04d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> $O/gpu_tests.txt
# the chip's own clock / power read-out while the bench runs (one sample per ~0.3 s; rocm-smi, text form)
( for i in $(seq 1 400); do echo "== $(date +%s.%N)"; timeout 5 /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|busy" ; sleep 0.25; done ) > $O/smi_during_bench.txt &
SMI=$!
( timeout 400 python bench.py --steps 20 2> $O/bench.err ) > $O/bench_refine_b128.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
( timeout 300 python bench.py --steps 20 --workload stream --no-cpu-baseline 2> $O/bench_stream.err ) > $O/bench_stream.json
timeout 900 bash tools/bench_configs.sh $O/bench_configs.jsonl > $O/bench_configs.txt 2>&1
timeout 900 bash tools/small_batch_lines.sh $O/small_batch.jsonl > $O/small_batch.md 2>&1
bash tools/profile_bench.sh $tag > $O/profile.log 2>&1
ls -la $O $R/gpurun_out/prof_$tag
