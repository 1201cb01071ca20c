#!/bin/bash
# round 5, call 7: upsample2x with 2x2 output blocks per thread (9 loads per 4 outputs), gn_apply prologue; same-box A/B vs the previous net_kernels.hip
O=gpurun_out/r05g; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_net_kernels.py -m gpu -x -q -k "upsample or groupnorm or stem" 2>&1 | tail -3
GDRNPP_HIP_LIB=ab_libs/old_net/libgdrnpp_hip.so python tools/upsample_dump.py /tmp/up_old.pt && python tools/upsample_dump.py /tmp/up_new.pt && python -c "
import torch
a, b = torch.load('/tmp/up_old.pt'), torch.load('/tmp/up_new.pt')
print('upsample bitwise equal to the one-output kernel:', all(torch.equal(x, y) for x, y in zip(a, b)))"
run() { python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), [(o['kernel'], round(o.get('ms_per_step', o['launch_ms']),3), round(o.get('hbm_frac', o['frac']),3)) for o in d['roofline_other_kernels'] if o['kernel'] in ('groupnorm_apply','upsample2x')])"; }
for i in 1 2 3; do
  GDRNPP_HIP_LIB=ab_libs/old_net/libgdrnpp_hip.so run old
  run new
done 2>&1 | tee $O/ab_net_kernels.txt
