import numpy as np
from scipy.special import erfc, log_ndtr
np.set_printoptions(precision=17)
U = 5.9
def q_true(u):  # log2 Phi(-u)
    return log_ndtr(-u) / np.log(2.0)
def fit(deg, iters=30):
    # weighted least squares on Chebyshev nodes, then reweight toward minimax of |delta h| = h ln2 |dq|
    n = 4000
    u = 0.5 * U * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))
    h = np.exp2(q_true(u))
    w = h.copy()
    V = np.vander(u / U, deg + 1, increasing=True)
    for it in range(iters):
        c, *_ = np.linalg.lstsq(V * w[:, None], q_true(u) * w, rcond=None)
        err = np.abs(np.exp2(V @ c) - h)
        w = w * (1 + 3 * err / err.max())   # push weight where error is large
        w /= w.max()
    return c / U ** np.arange(deg + 1)
def f32(x): return np.asarray(x, dtype=np.float32)
def fma32(a, b, c): return f32(a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64))
def gelu_fast(x, c32):
    x = f32(x); u = np.minimum(np.abs(x), f32(U))
    q = np.full_like(u, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        q = fma32(q, u, np.full_like(u, c32[k]))
    h = f32(np.exp2(q.astype(np.float64)))   # v_exp_f32 ~1 ulp: emulate correctly rounded, then add 1-ulp noise check below
    return fma32(-np.abs(x), h, np.maximum(x, f32(0))), h
from scipy.special import ndtr
xs = np.concatenate([np.linspace(-9, 9, 2000001), np.random.default_rng(0).normal(0, 2, 1000000)])
want = xs.astype(np.float32).astype(np.float64); want = want * ndtr(want)
for deg in (7, 8, 9, 10, 11):
    c = fit(deg); c32 = f32(c)
    got, h = gelu_fast(xs, c32)
    x32 = xs.astype(np.float32).astype(np.float64)
    err = np.abs(got.astype(np.float64) - want)
    print(deg, "max abs err", err.max(), "max err/max(|x|,1)", (err / np.maximum(np.abs(x32), 1)).max(), "at x=", x32[np.argmax(err)])
    if deg in (9, 10): print("   coeffs:", [float(v) for v in c32])
print("---- degree 8, final")
c = fit(8, iters=60); c32 = f32(c)
print("coeffs:", ", ".join(repr(float(v)) + "f" for v in c32))
rng = np.random.default_rng(1)
def gelu_fast_noisy(x, c32):
    x = f32(x); u = np.minimum(np.abs(x), f32(U))
    q = np.full_like(u, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        q = fma32(q, u, np.full_like(u, c32[k]))
    h = f32(np.exp2(q.astype(np.float64)))
    h = f32(h.astype(np.float64) * (1 + rng.choice([-1.2e-7, 0, 1.2e-7], size=h.shape)))  # 1 ulp of v_exp_f32
    return fma32(-u, h, np.maximum(x, f32(0)))
got = gelu_fast_noisy(xs, c32)
x32 = xs.astype(np.float32).astype(np.float64)
err = np.abs(got.astype(np.float64) - want)
print("noisy exp: max abs", err.max(), "max err/max(|x|,1)", (err / np.maximum(np.abs(x32), 1)).max())
# reference: fp32 GELU with a correctly rounded erf (what a 0.5-ulp libm would give)
from scipy.special import erf
xf = xs.astype(np.float32)
e32 = f32(erf(xf.astype(np.float64) * 0.7071067811865476))
ref = f32(f32(f32(0.5) * xf) * f32(f32(1) + e32))
errr = np.abs(ref.astype(np.float64) - want)
print("fp32 erf-form GELU (correctly rounded erf): max abs", errr.max(), "max err/max(|x|,1)", (errr / np.maximum(np.abs(x32), 1)).max())
# tails and specials
for v in (0.0, -0.0, 1e-30, -1e-30, 5.9, -5.9, 6.5, -6.5, 50.0, -50.0, np.inf, -np.inf):
    print(v, float(gelu_fast_noisy(np.array([v]), c32)[0]))
