"""Coefficients of the GELU of csrc/gemm_kernels.hip: degree-6 minimax-style fit of log2 Phi(-a) on [0, 6] (weighted
Chebyshev least squares, re-weighted towards the maximum error), and its accuracy over every fp16 input next to the
Abramowitz & Stegun 7.1.26 erf it replaces.  CPU only (numpy / scipy)."""
import numpy as np
from numpy.polynomial import Polynomial, chebyshev as C
from scipy.special import log_ndtr, ndtr

A, DEG = 6.0, 6
n = 4000
xs = np.cos(np.pi * (np.arange(n) + 0.5) / n) * A / 2 + A / 2
ys = log_ndtr(-xs) / np.log(2)
w = np.ones(n)
for _ in range(40):
    c = C.chebfit((xs - A / 2) / (A / 2), ys, DEG, w=w)
    e = np.abs(C.chebval((xs - A / 2) / (A / 2), c) - ys)
    w *= (e / e.mean()) ** 0.5
    w /= w.mean()
t = Polynomial([-1, 1 / (A / 2)])
co = sum(coef * t ** i for i, coef in enumerate(C.cheb2poly(c))).coef
print("q(a) coefficients, constant first:", [float(np.float32(v)) for v in co])
grid = np.linspace(0, A, 200001)
print("max |q - log2 Phi(-a)| on [0, 6]:", np.abs(np.polyval(co[::-1], grid) - log_ndtr(-grid) / np.log(2)).max())

co32 = co.astype(np.float32)


def gelu_new(x):
    x = x.astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(A))
    q = np.full_like(a, co32[DEG])
    for k in range(DEG - 1, -1, -1):
        q = (q * a + co32[k]).astype(np.float32)
    u = np.exp2(q.astype(np.float64)).astype(np.float32)
    return (np.maximum(x, 0) - a * u).astype(np.float32)


def gelu_as(x):
    x = x.astype(np.float32)
    ax = np.abs(x) * np.float32(0.70710678)
    tt = (1 / (np.float32(0.3275911) * ax + 1)).astype(np.float32)
    p = np.float32(1.061405429) * tt + np.float32(-1.453152027)
    for cc in (1.421413741, -0.284496736, 0.254829592):
        p = p * tt + np.float32(cc)
    p = p * tt
    e = np.exp2((ax * ax * np.float32(-1.44269504)).astype(np.float64)).astype(np.float32)
    return (0.5 * (np.abs(x) * (1 - p * e) + x)).astype(np.float32)


h = np.arange(0, 65536, dtype=np.uint16).view(np.float16)
h = h[np.isfinite(h)]
exact = h.astype(np.float64) * ndtr(h.astype(np.float64))
normal = np.abs(exact) > 6.2e-5
for name, fn in (("exp2-polynomial (shipped)", gelu_new), ("Abramowitz-Stegun 7.1.26 (round 1-3)", gelu_as)):
    got = fn(h.astype(np.float32)).astype(np.float64)
    rel = np.abs(got - exact) / np.maximum(np.abs(exact), 1e-30)
    print(f"{name}: fp16 results != correctly rounded {np.mean(got.astype(np.float16) != exact.astype(np.float16)):.4f}, "
          f"max rel err (normal range) {rel[normal].max():.3g}, max abs err {np.abs(got - exact).max():.3g}")
