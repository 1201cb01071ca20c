#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04q
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-other-mode-line > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-other-mode-line > /dev/null 2> $O/pmc_write.err
python tools/pmc_parse_bench_gemm.py $O/pmc_fetch $O/pmc_write > $O/pmc_gemm_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/pmc_gemm_traffic.json
