#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box): kernel stats, last-step breakdown, HBM-side traffic and
# matrix-pipe utilisation of the split GEMM.  Counters go in their own passes (no --stats / other trace domains with --pmc).
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_${1:-r02}
rm -rf $O; mkdir -p $O
cd $R
# --compute-streams 1: the trace that step_breakdown.py cuts into steps must hold ONE step at a time (kernel durations undisturbed by a second
# step on the chip = the durations of bench.py's roofline pass); the two-stream schedule of the headline gets its own stats below
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child --compute-streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/bench_under_trace.json 2> $O/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child > $O/bench_under_trace_two_streams.json 2> $O/trace2.err
find $O/trace2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_two_streams.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child --compute-streams 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child --compute-streams 1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-other-mode-line --no-pmc --pmc-child --compute-streams 1 > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_parse_bench_gemm.py $O/pmc_fetch $O/pmc_write > $O/pmc_gemm_traffic.json
python tools/pmc_mfma_busy.py $O/pmc_mfma > $O/pmc_gemm_mfma_busy.json
( for k in gemm_split2_pipe mlp_fused_x3 dwconv7_ln gn_apply depth_refine; do python tools/pmc_clock.py $O/pmc_mfma $k; done ) > $O/effective_clock.txt 2>&1
python tools/step_breakdown.py $O/trace "steady-state step, YCB-V convnext_a6 + refine, 128 ROIs" > $O/step_breakdown.md
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# keep the merged output small: raw traces stay on the box
rm -rf $O/trace $O/trace2 $O/pmc_fetch $O/pmc_write $O/pmc_mfma
ls -la $O; cat $O/pmc_gemm_traffic.json $O/pmc_gemm_mfma_busy.json; head -30 $O/step_breakdown.md
