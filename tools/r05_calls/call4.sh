#!/bin/bash
O=gpurun_out/r05d; mkdir -p $O
GDRNPP_H2D_DEBUG=1 timeout 300 python bench.py --workload stream --host-fed --steps 8 --no-cpu-baseline --no-pmc --no-roofline-pass --no-other-mode-line > $O/stream_hostfed.json 2> $O/stream_hostfed.err
grep H2D_TIMELINE $O/stream_hostfed.err | head -1 | python -c "
import sys, json
l=sys.stdin.read(); d=json.loads(l.split(' ',1)[1])
print('steps', [(round(a,2), round(b,2)) for a,b in d['steps_ms']])
print('copies', [(round(a,2), round(b,2)) for a,b in d['copies_ms']])
"
