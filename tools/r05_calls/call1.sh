#!/bin/bash
# round 5, call 1: the round-4 tree + this round's host-side changes on a GPU for the first time
O=gpurun_out/r05a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
timeout 400 python tools/b128_engine_errors.py > $O/engine_errors.txt 2>&1; tail -30 $O/engine_errors.txt
timeout 600 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a/bench.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['ms_per_step'], 'parity', d.get('parity_in_run'), 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source','')[:60], d['roofline'].get('traffic_live_error'), 'pmc s', d.get('pmc_live_seconds'))
print('cpu fwd', (d.get('cpu_baseline') or {}).get('stages',{}).get('forward_cpu_torch'))
PY
timeout 300 python bench.py --workload stream --host-fed --steps 20 --no-cpu-baseline --no-pmc > $O/stream_hostfed.json 2> $O/stream_hostfed.err; tail -c 400 $O/stream_hostfed.err
timeout 300 python bench.py --workload stream --steps 20 --no-cpu-baseline --no-pmc > $O/stream.json 2> $O/stream.err
python - <<'PY'
import json
for f in ('stream_hostfed','stream'):
    try:
        d=json.loads(open(f'gpurun_out/r05a/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('host_fed'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 300 python bench.py --workload lmo_upnp --steps 30 --warmup 5 --no-pmc > $O/lmo_upnp.json 2> $O/lmo_upnp.err; tail -c 300 $O/lmo_upnp.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05a/lmo_upnp.json').read().strip().splitlines()[-1])
    print('lmo', d['value'], d['ms_per_step'], 'cpu', {k:v for k,v in (d.get('cpu_baseline') or {}).items() if k!='stages'})
except Exception as e: print('lmo ERR', e)
PY
timeout 400 python tools/microbench_ops.py > $O/ops_microbench.json 2> $O/ops.err; tail -c 300 $O/ops.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r05a/ops_microbench.json'))
    for k,v in d.items():
        if 'roofline' in v: print(k, v['roofline']['bound'], round(v['roofline']['frac'],4))
except Exception as e: print('ops ERR', e)
PY
