#!/bin/bash
# round 5, call 13: the N > 1 code path as far as a 1-GPU box goes — RCCL process group with one rank, spawned by the driver's own launch line
O=gpurun_out/r05k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-other-mode-line > $O/bench_force_dist_torchrun.json 2> $O/torchrun.err; tail -c 400 $O/torchrun.err
timeout 300 python bench.py --force-dist --workload tless --gather-to-rank0 --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line > $O/bench_force_dist_gather0.json 2> $O/gather0.err; tail -c 300 $O/gather0.err
python - <<'PY'
import json
for f in ('bench_force_dist_torchrun','bench_force_dist_gather0'):
    try:
        d=json.loads(open(f'gpurun_out/r05k/{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value'],1), round(d['ms_per_step'],3), d['collective'], d['gather_ms'])
    except Exception as e: print(f,'ERR',e)
PY
