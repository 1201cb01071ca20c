#!/bin/bash
# Evidence set of a round-5 code state from ONE box: GPU tests + smoke, headline line (with in-run parity, live PMC traffic, CPU legs),
# stream / host-fed stream / bop7_stream lines, every BASELINE config, the reference's own batch sizes, custom-op microbench, network
# errors vs the reference's fp32 and fp64 runs, rocprofv3 kernel stats / step breakdown / matrix-pipe busy.  usage: r05_evidence.sh <tag>
tag=${1:-r05z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) >> $O/gpu_tests.txt
( for i in $(seq 1 400); do echo "== $(date +%s.%N)"; timeout 5 /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|busy" ; sleep 0.25; done ) > $O/smi_during_bench.txt &
SMI=$!
( timeout 600 python bench.py --steps 20 2> $O/bench.err ) > $O/bench_refine_b128.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
( timeout 300 python bench.py --steps 20 --workload stream --no-cpu-baseline --no-pmc 2> $O/bench_stream.err ) > $O/bench_stream.json
( timeout 300 python bench.py --steps 20 --workload stream --host-fed --no-cpu-baseline --no-pmc 2> $O/bench_stream_hostfed.err ) > $O/bench_stream_hostfed.json
( timeout 400 python bench.py --steps 21 --workload bop7_stream --host-fed --no-cpu-baseline --no-pmc 2> $O/bench_bop7_stream_hostfed.err ) > $O/bench_bop7_stream_hostfed.json
( timeout 300 python bench.py --workload lmo_upnp --steps 30 --warmup 5 --no-pmc 2> $O/bench_lmo_upnp.err ) > $O/bench_lmo_upnp.json
timeout 900 bash tools/bench_configs.sh $O/bench_configs.jsonl > $O/bench_configs.txt 2>&1
timeout 900 bash tools/small_batch_lines.sh $O/small_batch.jsonl > $O/small_batch.md 2>&1
( timeout 400 python tools/microbench_ops.py 2> $O/ops.err ) > $O/ops_microbench.json
( timeout 400 python tools/b128_engine_errors.py 2>&1 | grep -v amdgpu.ids ) > $O/b128_engine_errors_vs_fp64.txt
bash tools/profile_bench.sh $tag > $O/profile.log 2>&1
ls -la $O $R/gpurun_out/prof_$tag
