#!/bin/bash
# round 5, call 5: the scheduler keeps the host a step ahead (record copy on a side stream, packed async upload)
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_stream.py tests/test_gpu_crop_resize.py tests/test_gpu_evaluator.py -m gpu -x -q > $O/pytest_stream.txt 2>&1; tail -4 $O/pytest_stream.txt
GDRNPP_H2D_DEBUG=1 timeout 300 python bench.py --workload stream --host-fed --steps 20 --no-cpu-baseline --no-pmc > $O/stream_hostfed.json 2> $O/stream_hostfed.err
timeout 300 python bench.py --workload stream --steps 20 --no-cpu-baseline --no-pmc > $O/stream.json 2> $O/stream.err
timeout 400 python bench.py --workload bop7_stream --host-fed --steps 21 --no-cpu-baseline --no-pmc > $O/bop7_stream_hostfed.json 2> $O/bop7_stream_hostfed.err
python - <<'PY'
import json
for f in ('stream_hostfed','stream','bop7_stream_hostfed'):
    try:
        d=json.loads(open(f'gpurun_out/r05e/{f}.json').read().strip().splitlines()[-1])
        h=d.get('host_fed') or {}
        print(f, round(d['value'],1), round(d['ms_per_step'],3), {k:h[k] for k in h if k!='note'})
    except Exception as e: print(f, 'ERR', e)
PY
grep H2D_TIMELINE $O/stream_hostfed.err | head -1 | python -c "
import sys, json
l=sys.stdin.read(); d=json.loads(l.split(' ',1)[1])
print('steps', [(round(a,2), round(b,2)) for a,b in d['steps_ms'][:5]])
print('copies', [(round(a,2), round(b,2)) for a,b in d['copies_ms'][:40]])
"
timeout 200 python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fixed batch', d['value'], d['ms_per_step'])"
