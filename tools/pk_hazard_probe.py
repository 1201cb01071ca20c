"""Packed-fp32 VALU results of one stream's waves while another stream's waves issue MFMAs on the same SIMDs (MI355X): which
op_sel / op_sel_hi combinations of v_pk_add_f32 / v_pk_mul_f32 go wrong, in which lanes, beside which companion instruction
(tools/probe/*.hip; diagnostic only).  Companions run one 256-thread workgroup per CU so the probe's waves share their SIMDs."""
import os, sys, ctypes, collections
import numpy as np
import torch
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libpk_probe.so"))
probe.pk_probe2_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
probe.companion_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
MAXLOG = 16384
log = torch.zeros(8 + 8 * MAXLOG, dtype=torch.int32, device=dev)
NT = 1024 * 1024
sink = torch.empty(NT, dtype=torch.float32, device=dev)
sink2 = torch.empty(2048 * 256, dtype=torch.float32, device=dev)
src = torch.randn(32 * 1048576, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
names = ["v_mfma_f32_32x32x16_f16", "v_fma_mix_f32", "v_cvt_pk_f16_f32", "v_dot2c_f32_f16", "global_load_lds_dwordx4", "v_mov_b32_dpp + ds_read_b128",
         "v_accvgpr_write/read", "f64 VALU", "v_pk_fma_f32", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_32x32x16_bf16",
         "v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_fp8_fp8", "v_mfma_f32_16x16x16_f16"]
iters = [120000, 240000, 240000, 240000, 12000, 120000, 120000, 120000, 240000, 240000, 120000, 120000, 60000, 120000, 240000]
def combos(mask16):
    return [f"op_sel:[{k >> 3 & 1},{k >> 2 & 1}] op_sel_hi:[{k >> 1 & 1},{k & 1}]" for k in range(16) if mask16 >> k & 1]
for kind in [-1] + list(range(len(names))):
    log.zero_(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(3):
        if kind >= 0:
            with torch.cuda.stream(streams[1]):
                ev[0].record()
                rc = probe.companion_launch(kind, sink2.data_ptr(), src.data_ptr(), iters[kind], 256, streams[1].cuda_stream); assert rc == 0, rc
                ev[1].record()
        with torch.cuda.stream(streams[0]):
            for _ in range(4):
                rc = probe.pk_probe2_launch(log.data_ptr(), MAXLOG, 100, sink.data_ptr(), NT, streams[0].cuda_stream); assert rc == 0
        torch.cuda.synchronize()
    L = log.cpu().numpy().view(np.uint32); cnt = int(L[0]); E = L[8:8 + 8 * min(cnt, MAXLOG)].reshape(-1, 8)
    ms = ev[0].elapsed_time(ev[1]) if kind >= 0 else 0.0
    print(f"beside {(names[kind] if kind >= 0 else 'nothing'):32s} ({ms:6.2f} ms per companion launch): {cnt} loop trips with a wrong packed result")
    if cnt:
        m = int(np.bitwise_or.reduce(E[:, 3]))
        print(f"     lanes {sorted(set(E[:, 2].tolist()))[0]}..{sorted(set(E[:, 2].tolist()))[-1]} (quarters {dict(collections.Counter((E[:, 2] // 16).tolist()))})")
        print(f"     v_pk_add_f32 wrong with: {combos(m & 0xffff)}")
        print(f"     v_pk_mul_f32 wrong with: {combos(m >> 16)}")
