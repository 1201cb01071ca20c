"""Where the error of the three-product GEMM comes from: kernel output vs (a) fp64, (b) an fp64 evaluation of the SAME three products
of the same fp16 operands (isolates fp32 accumulation inside the MFMA chain), per K."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdrnpp_bop2022_amd import hip_lib as hip  # noqa: E402

torch.manual_seed(0)
for m, k, n in [(2048, 128, 512), (2048, 512, 128), (2048, 1024, 256), (2048, 2048, 512), (2048, 4096, 1024)]:
    x = torch.randn(m, k, device="cuda")
    x[:, :3] *= 40.0
    w = torch.randn(n, k, device="cuda") * k ** -0.5
    pk = hip.pack_weight_f16x2(w)
    planes, inv = hip.unpack_weight_f16x2(pk)
    out3 = hip.linear_f32_split(x, pk, None)
    out6 = hip.linear_f32_split(x, hip.pack_weight_bf16x3(w), None)
    want = x.double() @ w.double().t()
    h = x.half()
    l = (x - h.float()).half()
    D = lambda a, b: a.double() @ b.double().t()  # noqa: E731
    emu = (D(h, planes[1]) + D(l, planes[0]) + D(h, planes[0])) * inv
    s = want.abs().max()
    f32 = torch.nn.functional.linear(x, w)
    print(f"K={k:5d} N={n:5d}: kernel-fp64 {((out3.double() - want).abs().max() / s).item():.2e}  emulation-fp64 "
          f"{((emu - want).abs().max() / s).item():.2e}  kernel-emulation {((out3.double() - emu).abs().max() / s).item():.2e}  "
          f"six products {((out6.double() - want).abs().max() / s).item():.2e}  torch fp32 {((f32.double() - want).abs().max() / s).item():.2e}")
