#!/bin/bash
# round 5, call 3: suite after the split-refine removal, copy-stream probe, host-fed stream lines with the high-priority copy stream
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt
timeout 120 python tools/h2d_overlap_probe.py > $O/h2d_overlap_probe.txt 2>&1; cat $O/h2d_overlap_probe.txt
timeout 300 python bench.py --workload stream --host-fed --steps 20 --no-cpu-baseline --no-pmc > $O/stream_hostfed.json 2> $O/stream_hostfed.err; tail -c 300 $O/stream_hostfed.err
timeout 400 python bench.py --workload bop7_stream --host-fed --steps 21 --no-cpu-baseline --no-pmc > $O/bop7_stream_hostfed.json 2> $O/bop7_stream_hostfed.err; tail -c 300 $O/bop7_stream_hostfed.err
python - <<'PY'
import json
for f in ('stream_hostfed','bop7_stream_hostfed'):
    try:
        d=json.loads(open(f'gpurun_out/r05c/{f}.json').read().strip().splitlines()[-1])
        h=d.get('host_fed') or {}
        print(f, round(d['value'],1), round(d['ms_per_step'],3), {k:h[k] for k in h if k!='note'})
    except Exception as e: print(f, 'ERR', e)
PY
