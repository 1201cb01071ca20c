#!/bin/bash
O=gpurun_out/r05u; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > $O/bench_rccl_one_rank_torchrun.json 2> $O/torchrun.err; tail -c 300 $O/torchrun.err
timeout 300 python bench.py --force-dist --workload tless --gather-to-rank0 --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line > $O/bench_rccl_one_rank_gather_to_rank0.json 2> $O/gather0.err; tail -c 300 $O/gather0.err
S=$(date +%s.%N); timeout 600 python bench.py > $O/bench_default_flags.json 2> $O/default.err; E=$(date +%s.%N); echo "default-flags wall $(echo "$E - $S" | bc) s" | tee $O/default_wall.txt
python - <<'PY'
import json
for n in ("bench_rccl_one_rank_torchrun", "bench_rccl_one_rank_gather_to_rank0", "bench_default_flags"):
    try:
        d = json.loads(open(f"gpurun_out/r05u/{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), round(d["ms_per_step"], 3), d["config"]["compute_streams"], d.get("collective"), (d.get("single_stream_mode") or {}).get("value"))
    except Exception as e:
        print(n, "ERR", e)
PY
