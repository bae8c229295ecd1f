"""Margin analysis for an e4m3 nomination scan, on the bench distribution -- CPU only.

Question: if the corpus / queries were scanned in OCP e4m3 (v_mfma_f32_32x32x64_f8f6f4: twice the fp16 MFMA rate) and the
hits re-scored exactly, how many rows per query would the scan have to nominate so that NO row of the exact top-80 can be
missed?  The threshold of the fp8 scan must be lowered by a rigorous bound on |q.x - q8.x8| (Cauchy-Schwarz, as the
exact-float32 mode does for fp16 roundings: DESIGN.md section 2.6); the nominations are the rows whose fp8 score exceeds
the lowered threshold.  Reported next to it: the same count with a merely statistical margin (6 sigma of the observed
error), which would not be exact."""
import numpy as np
from scipy.stats import norm


def to_e4m3(x):
    """round-to-nearest-even onto OCP e4m3fn (4 exponent bits, bias 7, 3 mantissa bits, max 448, subnormals 2^-9)"""
    x = np.asarray(x, np.float32)
    ax = np.abs(x)
    e = np.floor(np.log2(np.maximum(ax, 2.0 ** -20)))
    e = np.clip(e, -6, 8)                       # normal exponents -6 .. 8; below: subnormal spacing 2^-9
    step = 2.0 ** (e - 3)
    y = np.round(ax / step) * step              # np.round is half-to-even
    y = np.minimum(y, 448.0)
    return np.sign(x) * y.astype(np.float32)


rng = np.random.default_rng(0)
n_sample, nq, N, k = 400_000, 256, 18_000_000, 80
xb = rng.standard_normal((n_sample, 128)).astype(np.float16).astype(np.float32)
xq = rng.standard_normal((nq, 128)).astype(np.float16).astype(np.float32)
# per-tensor power-of-two scale so that the values sit in e4m3's normal range (N(0,1): |x| < 6 -> scale 2^6 keeps 3 bits)
scale = 64.0
xb8, xq8 = to_e4m3(xb * scale) / scale, to_e4m3(xq * scale) / scale
E = np.linalg.norm(xb - xb8, axis=1)
Xn = np.linalg.norm(xb8, axis=1)
qe = np.linalg.norm(xq - xq8, axis=1)
qn = np.linalg.norm(xq8, axis=1)
print(f"relative rounding error of a row: {np.mean(E / np.linalg.norm(xb, axis=1)):.4f} (fp16: {2**-11 / np.sqrt(3):.5f})")
S = xq @ xb.T
S8 = xq8 @ xb8.T
err = S8 - S
sig = np.linalg.norm(xq, axis=1)                      # score sigma of query q over random rows
tau = sig * norm.isf(k / N)                            # its top-80-of-18M threshold
margin_cs = qn * E.max() + qe * Xn.max() + qe * E.max()          # rigorous, per query (max over rows as in section 2.6)
margin_cs_row = qn[:, None] * E[None, :] + qe[:, None] * Xn[None, :] + qe[:, None] * E[None, :]   # per-row norms stored
margin_6s = 6.0 * err.std(axis=1)
print(f"observed score error: sigma {err.std():.3f}, max |err| {np.abs(err).max():.3f}; score sigma {sig.mean():.2f}; tau(top-80 of 18M) {tau.mean():.2f}")
print(f"Cauchy-Schwarz margin: {margin_cs.mean():.2f} (per-row norms: {margin_cs_row.mean():.2f}); 6-sigma statistical margin: {margin_6s.mean():.2f}")
for name, m in (("rigorous (max norms)", margin_cs), ("rigorous (per-row norms)", margin_cs_row.mean(axis=1)), ("6 sigma, NOT exact", margin_6s)):
    # rows whose fp8 score can exceed tau - m: fp8 score ~ N(0, sig^2 + err^2) ~ N(0, sig^2)
    p = norm.sf((tau - m) / sig)
    print(f"  {name:26s}: nominations at the FINAL threshold {np.mean(p) * N:9.0f} per query ({np.mean(p) * N / k:6.1f} x k); "
          f"over the geometric rounds (x ~21 = 1730 / 80 today) ~{np.mean(p) * N * 1730 / 80:11.0f}")
print("today (fp16 scan, exact thresholds): 1730 candidates per query in total, 80 at the final threshold")
