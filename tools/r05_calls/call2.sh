#!/bin/bash
# round 5, call 2: split refine kernel, fp64-anchored parity test, host-fed stream test + lines, bop7_stream
O=gpurun_out/r05b; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_refine or fused_tail" > $O/pytest_split.txt 2>&1; tail -15 $O/pytest_split.txt
timeout 200 python tools/microbench_refine_split.py > $O/refine_split.md 2>&1; cat $O/refine_split.md
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -8 $O/pytest.txt
timeout 300 python bench.py --workload stream --host-fed --steps 20 --no-cpu-baseline --no-pmc > $O/stream_hostfed.json 2> $O/stream_hostfed.err; tail -c 300 $O/stream_hostfed.err
timeout 400 python bench.py --workload bop7_stream --host-fed --steps 21 --no-cpu-baseline --no-pmc > $O/bop7_stream_hostfed.json 2> $O/bop7_stream_hostfed.err; tail -c 300 $O/bop7_stream_hostfed.err
python - <<'PY'
import json
for f in ('stream_hostfed','bop7_stream_hostfed'):
    try:
        d=json.loads(open(f'gpurun_out/r05b/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('host_fed'), d.get('stream'))
    except Exception as e: print(f, 'ERR', e)
PY
for bsz in 8 32; do timeout 200 python bench.py --batch $bsz --steps 30 --no-cpu-baseline --no-pmc --no-other-mode-line 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['config']['rois_per_gpu'], d['value'], d['ms_per_step'], d['stages_ms'], [ (o['kernel'], o['launch_ms']) for o in d['roofline_other_kernels'] if 'refine' in o['kernel']])"; done
