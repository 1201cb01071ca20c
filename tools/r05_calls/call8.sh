#!/bin/bash
# round 5, call 8: dwconv7x7+LN pixel tiles 4x8 / 3x8 against 2x8 (fewer row loads per output), isolated and in the step
O=gpurun_out/r05h; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_net_kernels.py -m gpu -x -q -k "dwconv" 2>&1 | tail -2
python - <<'PY' 2>&1 | tee gpurun_out/r05h/dwconv_tiles.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gdrnpp_bop2022_amd import hip_lib
dev = "cuda"; torch.manual_seed(0)
ref = {}
for tile, name in ((0, "2x8"), (4, "3x8"), (3, "4x8")):
    hip_lib.set_option("dwconv_tile", tile)
    tot = 0.0; line = []
    for hw, c, nblk in [(64, 128, 3), (32, 256, 3), (16, 512, 27), (8, 1024, 3)]:
        x = torch.randn(128, c, hw, hw, device=dev, generator=torch.Generator(device=dev).manual_seed(hw)).contiguous(memory_format=torch.channels_last)
        g_ = torch.Generator(device=dev).manual_seed(c)
        w = torch.randn(49, c, device=dev, generator=g_) * 0.1
        b = torch.randn(c, device=dev, generator=g_); g = torch.randn(c, device=dev, generator=g_); be = torch.randn(c, device=dev, generator=g_)
        fn = lambda: hip_lib.dwconv7x7_ln(x, w, b, g, be, 1e-6)
        y = fn()
        if tile == 0: ref[hw] = y.clone()
        same = torch.equal(y, ref[hw])
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        tot += nblk * t
        line.append(f"{hw}x{hw}xC{c}: {t*1e3:.1f} us ({'bit-equal' if same else 'max diff %.2e' % (y-ref[hw]).abs().max().item()})")
    print(name, " | ".join(line), f"| total per forward {tot:.3f} ms")
hip_lib.set_option("dwconv_tile", -1)
PY
run() { python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-other-mode-line $2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), [(o['kernel'], round(o.get('ms_per_step', o['launch_ms']),3)) for o in d['roofline_other_kernels'] if o['kernel'] in ('dwconv7_ln',)])"; }
for i in 1 2; do
  run tile2x8 ""
  run tile3x8 "--opt dwconv_tile=4"
  run tile4x8 "--opt dwconv_tile=3"
done 2>&1 | tee -a $O/dwconv_tiles.txt
