#!/bin/bash
# stem with packed FMAs + two pixels per trip vs the scalar form (same box), bitwise comparison
timeout 200 python -m pytest tests/test_gpu_net_kernels.py -m gpu -x -q -k "stem" 2>&1 | tail -2
echo old; GDRNPP_HIP_LIB=ab_libs/old_net/libgdrnpp_hip.so python tools/stem_dump.py /tmp/stem_old.pt
echo new; python tools/stem_dump.py /tmp/stem_new.pt
python -c "
import torch
a, b = torch.load('/tmp/stem_old.pt'), torch.load('/tmp/stem_new.pt')
print('stem bitwise equal to the scalar form:', [bool(torch.equal(x, y)) for x, y in zip(a, b)], [float((x-y).abs().max()) for x, y in zip(a, b)])"
