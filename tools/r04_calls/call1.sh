#!/bin/bash
# Round 4, GPU call 1: the new range check / packing / tail kernels under test, the row-scale survey, the VALU probe, the cost of
# the range check and the A-direct experiment (isolated MLP shapes), the headline + stream bench lines, dwconv skeleton counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 ) > $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $O/smoke.txt
( timeout 300 python tools/x3_row_scale_survey.py 2>&1 | tail -120 ) > $O/survey_seeded.txt
( timeout 300 python tools/x3_row_scale_survey.py --random-init 2>&1 | tail -120 ) > $O/survey_default_init.txt
( timeout 60 _ab/valu_rate_probe 2>&1 ) > $O/valu_rate_probe.txt
for v in default norange adirect default norange adirect; do
  if [ $v = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/$v/libgdrnpp_hip.so; fi
  echo "== $v" >> $O/mlp_shapes_ab.txt
  ( X3=1 timeout 200 python tools/mlp_shapes.py 2>&1 | grep -v amdgpu.ids ) >> $O/mlp_shapes_ab.txt
done
unset GDRNPP_HIP_LIB
( timeout 400 python bench.py --steps 20 2> $O/bench.err ) > $O/bench_refine_b128.json
( timeout 300 python bench.py --steps 20 --workload stream --no-cpu-baseline 2> $O/bench_stream.err ) > $O/bench_stream.json
# dwconv: skeleton build (no input loads, no weight reads) under SQ counters
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "bank|VALU|VGPR|INST_CYCLES" | head -80 > $O/counters_available.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for lib in dw_skel default; do
    if [ $lib = default ]; then unset GDRNPP_HIP_LIB; else export GDRNPP_HIP_LIB=$R/_ab/$lib/libgdrnpp_hip.so; fi
    LN=0 timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${lib}_$i -- python $R/tools/microbench_dwconv.py > $O/pmc_${lib}_$i.out 2> $O/pmc_${lib}_$i.err || echo "pass $i $lib failed" >> $O/pmc_fail.txt
  done
done
unset GDRNPP_HIP_LIB
cd $R
for lib in dw_skel default; do
  mkdir -p $O/pm_$lib; for i in 1 2; do [ -d $O/pmc_${lib}_$i ] && mv $O/pmc_${lib}_$i $O/pm_$lib/p$i; done
  python tools/pmc_parse_any.py $O/pm_$lib dwconv7_ln > $O/dwconv_counters_$lib.txt 2>&1
  rm -rf $O/pm_$lib
done
ls -la $O
