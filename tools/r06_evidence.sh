#!/bin/bash
# Evidence set of a round-6 code state from ONE box (everything profiles/README.md's index cites for the current state):
#   GPU tests + smoke; headline line (in-run parity, live PMC traffic, CPU legs, two_stream_guard, pk_hazard_probe); rocprofv3 kernel
#   stats of the same command (one-stream trace + two-stream trace), step breakdown, matrix-pipe busy, effective clocks;
#   stream / host-fed / bop7_stream lines, with the detections coming out of gdrnpp_yolox_postprocess (--with-yolox-post);
#   every BASELINE config; 8-64 ROIs eager and as hipGraphs in flight; custom-op microbench + ROIAlign launch shapes;
#   parity at the iteration sizes (1 024 T-LESS / 512 YCB-V ROIs: default path, six products, unfused — and the reference's own
#   fp32 forward through the same bars); the two-stream guard tests with the raw hazard probe; dwconv beside the GEMM.
# usage: bash tools/r06_evidence.sh <tag>        (replaces the one-off tools/r04_calls/, tools/r05_calls/ scripts)
tag=${1:-r06z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) >> $O/gpu_tests.txt
( timeout 300 python -m pytest tests/test_gpu_stream_guard.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12 ) > $O/two_stream_guard_tests.txt
cp gpurun_out/pk_hazard_probe.json $O/pk_hazard_probe.json 2>/dev/null
cp gpurun_out/large_parity_tless_b1024.txt gpurun_out/large_parity_ycbv_b512.txt $O/ 2>/dev/null
( for i in $(seq 1 400); do echo "== $(date +%s.%N)"; timeout 5 /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|busy" ; sleep 0.25; done ) > $O/smi_during_bench.txt &
SMI=$!
( timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench.err ) > $O/bench_refine_b128.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
S="--no-cpu-baseline --no-pmc"
( timeout 300 python bench.py --steps 20 --workload stream $S 2> $O/bench_stream.err ) > $O/bench_stream.json
( timeout 300 python bench.py --steps 20 --workload stream --host-fed $S 2> $O/bench_stream_hostfed.err ) > $O/bench_stream_hostfed.json
( timeout 300 python bench.py --steps 20 --workload stream --host-fed --with-yolox-post $S 2> $O/bench_stream_hostfed_yolox.err ) > $O/bench_stream_hostfed_yolox.json
( timeout 400 python bench.py --steps 28 --workload bop7_stream --host-fed --with-yolox-post $S 2> $O/bench_bop7_stream_hostfed_yolox.err ) > $O/bench_bop7_stream_hostfed_yolox.json
( timeout 400 python bench.py --steps 28 --workload bop7_stream --host-fed $S 2> $O/bench_bop7_stream_hostfed.err ) > $O/bench_bop7_stream_hostfed.json
( timeout 300 python bench.py --workload lmo_upnp --steps 30 --warmup 5 --no-pmc 2> $O/bench_lmo_upnp.err ) > $O/bench_lmo_upnp.json
timeout 900 bash tools/bench_configs.sh $O/bench_configs.jsonl > $O/bench_configs.txt 2>&1
timeout 900 bash tools/small_batch_lines.sh $O/small_batch.jsonl > $O/small_batch.md 2>&1
# the image stream at the reference's own step sizes: eager scheduler (two streams) against graph-backed steps (four in flight)
: > $O/stream_small_steps.jsonl
for b in 8 16 32; do
  python bench.py --workload stream --batch $b --steps 60 --warmup 4 $S --no-roofline-pass --no-other-mode-line 2>/dev/null | grep '^{' >> $O/stream_small_steps.jsonl
  python bench.py --workload stream --graph --batch $b --steps 60 --warmup 4 $S --no-roofline-pass --no-other-mode-line 2>/dev/null | grep '^{' >> $O/stream_small_steps.jsonl
done
( timeout 400 python tools/microbench_ops.py 2> $O/ops.err ) > $O/ops_microbench.json
( timeout 200 python tools/roi_align_variants.py 2>/dev/null ) > $O/roi_align_variants.jsonl
( timeout 400 python tools/b128_engine_errors.py 2>&1 | grep -v amdgpu.ids ) > $O/b128_engine_errors_vs_fp64.txt
( timeout 400 python tools/large_parity_dump.py tless 1024 $O 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/large_parity_dump.log
( timeout 400 python tools/large_parity_dump.py ycbv 512 $O 2>&1 | grep -v amdgpu.ids | tail -3 ) >> $O/large_parity_dump.log
rm -f $O/large_parity_*.npz
( timeout 300 python tools/dwconv_shared_probe.py 2>/dev/null ) > $O/dwconv_shared_probe.jsonl
bash tools/profile_bench.sh $tag > $O/profile.log 2>&1
ls -la $O $R/gpurun_out/prof_$tag
