"""Feasibility probe: fp32-accurate GEMM from bf16 MFMA by 3-way operand splitting (6 partial products).

x = h + m + l with h, m, l bf16 (8 significant bits each, 24 together); the six products hh, hm, mh, hl, lh, mm
carry every term above 2^-25 relative, accumulated in fp32 by the bf16 MFMA.  The probe emulates it with ONE library
bf16 GEMM over a 6x longer K (operands concatenated along K) to answer two questions before a kernel is written:
what does the matrix core's accumulation do to the error (vs an fp64 product), and what rate does the library reach.
"""
import sys
import time

import torch


def split3(x):
    h = x.bfloat16()
    r = x - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    return h, m, l


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda"
    torch.manual_seed(0)
    shapes = [(128 * 64 * 64, 128, 512), (128 * 64 * 64, 512, 128), (128 * 16 * 16, 512, 2048), (128 * 16 * 16, 2048, 512)]
    for m_, k, n in shapes:
        x = torch.randn(m_, k, device=dev)
        w = torch.randn(n, k, device=dev) * k ** -0.5
        sub = slice(0, 4096)
        ref = x[sub].double() @ w.double().t()
        scale = ref.abs().max().item()
        y32 = x @ w.t()
        e32 = ((y32[sub].double() - ref).abs().max().item() / scale, (y32[sub].double() - ref).abs().mean().item() / scale)
        xh, xm, xl = split3(x)
        wh, wm, wl = split3(w)
        a6 = torch.cat([xl, xh, xm, xm, xh, xh], 1).contiguous()     # small terms first
        b6 = torch.cat([wh, wl, wm, wh, wm, wh], 1).contiguous()
        y6 = torch.mm(a6, b6.t(), out_dtype=torch.float32)
        e6 = ((y6[sub].double() - ref).abs().max().item() / scale, (y6[sub].double() - ref).abs().mean().item() / scale)
        a3 = torch.cat([xm, xh, xh], 1).contiguous()
        b3 = torch.cat([wh, wm, wh], 1).contiguous()
        y3 = torch.mm(a3, b3.t(), out_dtype=torch.float32)
        e3 = ((y3[sub].double() - ref).abs().max().item() / scale, (y3[sub].double() - ref).abs().mean().item() / scale)
        t32 = timeit(lambda: x @ w.t())
        t6 = timeit(lambda: torch.mm(a6, b6.t(), out_dtype=torch.float32))
        t3 = timeit(lambda: torch.mm(a3, b3.t(), out_dtype=torch.float32))
        tsplit = timeit(lambda: split3(x))
        fl = 2.0 * m_ * k * n
        print(f"M={m_} K={k} N={n}: fp32 {t32:.3f} ms ({fl / t32 / 1e9:.0f} TF) err max/mean {e32[0]:.2e}/{e32[1]:.2e} | "
              f"bf16x6 {t6:.3f} ms ({fl / t6 / 1e9:.0f} TF eff, {6 * fl / t6 / 1e9:.0f} TF bf16) err {e6[0]:.2e}/{e6[1]:.2e} | "
              f"bf16x3 {t3:.3f} ms err {e3[0]:.2e}/{e3[1]:.2e} | split {tsplit:.3f} ms", flush=True)


if __name__ == "__main__":
    sys.exit(main())
