"""Gate 0(a) of the int8 nomination scan -- CPU only.

Question: if the k = 80 search scanned an int8 copy of the corpus (v_mfma_i32_32x32x32_i8: twice the fp16 MFMA rate, exact
i32 sums) and re-scored the nominated rows from the fp16 rows, how many rows per query must the scan nominate so that NO row
of the exact top-k can be missed?

Quantisation as built in csrc/mips_kernels.hip (`quantise_rows_i8`):
    u_d   = x_d / c_d                     c_d = max |x_d| over the corpus (per-dimension equalisation, folded into the queries)
    xi_d  = rint(127 u_d / f_b)           f_b = max |u| over the row's 32-row block (1 for the global-scale variant)
    q'_d  = q_d c_d,  qi_d = rint(q'_d / s_q),  s_q = max |q'| / 127
    score ~ (f_b / 127) s_q sum_d qi_d xi_d
Error, in units of (f_b / 127) s_q:
    q.x - approx = sum_d (q'_d / s_q) r_d + sum_d e_d xi_d,   r = 127 u / f_b - xi (|r_d| <= 1/2),  e = q'/s_q - qi (|e_d| <= 1/2)
    |...| <= ||q'/s_q|| R + ||e|| X,       R = max_rows ||r||,  X = max_rows ||xi||       (Cauchy-Schwarz, maxima kept at add time)
The scan nominates acc > (tau - margin) / unit; nominations at the final threshold and summed over the geometric rounds are
reported for (a) the bench distribution N(0,1) and (b) a non-Gaussian corpus: the embeddings of the end-to-end test's small
encoder (tests/golden/encoder_golden.npz weights, random word sequences), whose dimensions differ in scale and are correlated.
"""
import os
import sys

import numpy as np
from scipy.stats import norm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def quantise(xb, block):
    """xb: the CENTRED rows (x - mean; q.mean is a per-query constant that does not change the ranking)"""
    c = np.abs(xb).max(axis=0)
    c[c == 0] = 1.0
    u = xb / c
    n = xb.shape[0]
    if block:
        nb = n // block
        f = np.abs(u[: nb * block]).reshape(nb, block * xb.shape[1]).max(axis=1)
        f[f == 0] = 1.0
        f_row = np.repeat(f, block)
        u = u[: nb * block]
    else:
        f_row = np.ones(n, np.float32)
    v = 127.0 * u / f_row[:, None]
    xi = np.rint(v)
    R = np.linalg.norm(v - xi, axis=1).max()
    X = np.linalg.norm(xi, axis=1).max()
    return c, f_row, xi.astype(np.float32), R, X


def analyse(name, xb, xq, N, k, rounds_candidates=1730, centre=False):
    xb = xb.astype(np.float16).astype(np.float32)
    xq = xq.astype(np.float16).astype(np.float32)
    if centre:
        xb = xb - xb.mean(axis=0, dtype=np.float64).astype(np.float32)   # scores below are q.(x - mean): the ranking is unchanged
        name += ", centred"
    S = xq @ xb.T
    n = xb.shape[0]
    # top-k-of-N threshold of every query, from the empirical score distribution of the sample where it reaches, else a
    # normal tail fitted to the sample (mean / sigma per query)
    mu, sig = S.mean(axis=1), S.std(axis=1)
    z = norm.isf(k / N)
    tau = mu + sig * z
    print(f"== {name}: sample {n} rows, {xq.shape[0]} queries; score sigma {sig.mean():.3f}, tau(top-{k} of {N:.1e}) {tau.mean():.3f}")
    for label, block in (("global scale", 0), ("32-row block scales", 32)):
        c, f_row, xi, R, X = quantise(xb, block)
        qp = xq * c
        s_q = np.abs(qp).max(axis=1) / 127.0
        qs = qp / s_q[:, None]
        qi = np.rint(qs)
        Mq = np.linalg.norm(qs, axis=1) * R + np.linalg.norm(qs - qi, axis=1) * X        # integer units
        unit = (f_row[None, : xi.shape[0]] / 127.0) * s_q[:, None]                          # score units per integer unit
        approx = (qi @ xi.T) * unit
        err = approx - S[:, : xi.shape[0]]
        margin = Mq[:, None] * unit                                                       # per (query, row block)
        assert np.all(np.abs(err) <= margin * (1 + 1e-5)), "the bound must hold"
        # nominations: rows whose approximate score exceeds tau - margin; counted on the sample, scaled to N
        nom = (approx > (tau[:, None] - margin)).sum(axis=1) * (N / xi.shape[0])
        above = (S > tau[:, None]).sum(axis=1) * (N / n)
        # the same over the geometric rounds: thresholds of rank k among n_r rows seen, n_r growing by 3.05 from 8192
        tot = 0.0
        seen = 8192.0
        while seen < N:
            nxt = min(N, seen * 3.05)
            t_r = mu + sig * norm.isf(min(0.5, k / seen))
            p = (approx > (t_r[:, None] - margin)).mean(axis=1)
            tot += np.mean(p) * (nxt - seen)
            seen = nxt
        print(f"  {label:20s}: R {R:.2f} X {X:.0f}; margin {margin.mean():.3f} (observed error sigma {err.std():.3f}, max {np.abs(err).max():.3f}); "
              f"nominations at the final threshold {nom.mean():7.0f} per query = {nom.mean() / max(above.mean(), 1e-9):5.2f} x the rows above it "
              f"({nom.mean() / k:5.2f} x k); over the rounds {tot:7.0f} per query (fp16 scan: ~{rounds_candidates})")


def gaussian(n_sample=400_000, nq=128):
    rng = np.random.default_rng(0)
    return rng.standard_normal((n_sample, 128)), rng.standard_normal((nq, 128))


def encoder_embeddings(n_docs=100_000, n_q=128):
    """the e2e test's small encoder (tests/golden) on random word sequences, CPU oracle"""
    import json
    import torch
    from oracle import bert_torch_cpu
    g = os.path.join(ROOT, "tests", "golden")
    cfg = json.load(open(os.path.join(g, "encoder_config.json")))
    z = np.load(os.path.join(g, "encoder_golden.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}
    rng = np.random.default_rng(11)
    vocab = cfg["vocab_size"]

    def embed(n, lo, hi, is_q):
        out = []
        for b0 in range(0, n, 2048):
            m = min(2048, n - b0)
            L = hi + 2
            ids = np.zeros((m, L), np.int64)
            mask = np.zeros((m, L), bool)
            for i in range(m):
                ln = int(rng.integers(lo, hi))
                ids[i, 0] = 2
                ids[i, 1: 1 + ln] = rng.integers(5, vocab, size=ln)
                ids[i, 1 + ln] = 3
                mask[i, : ln + 2] = True
            e = bert_torch_cpu.get_embed(sd, torch.from_numpy(ids), torch.from_numpy(mask), is_q, cfg["num_hidden_layers"],
                                         cfg["num_attention_heads"])
            out.append(np.asarray(e, np.float32))
        return np.concatenate(out)

    return embed(n_docs, 5, 40, False), embed(n_q, 4, 8, True)


if __name__ == "__main__":
    xb, xq = gaussian()
    analyse("bench distribution N(0,1)", xb, xq, 18_000_000, 80)
    try:
        xb, xq = encoder_embeddings()
        print(f"   (per-dimension max |x|: min {np.abs(xb).max(axis=0).min():.3f} max {np.abs(xb).max(axis=0).max():.3f}; "
              f"row norms {np.linalg.norm(xb, axis=1).min():.2f} .. {np.linalg.norm(xb, axis=1).max():.2f})")
        analyse("small-encoder embeddings (e2e test model)", xb, xq, 18_000_000, 80)
        analyse("small-encoder embeddings, N = its own 100 k", xb, xq, 100_000, 80)
        analyse("small-encoder embeddings", xb, xq, 18_000_000, 80, centre=True)
        analyse("small-encoder embeddings, N = its own 100 k", xb, xq, 100_000, 80, centre=True)
        xc = xb - xb.mean(axis=0)
        ev = np.linalg.eigvalsh(np.cov(xc.T))[::-1]
        print(f"   centred covariance spectrum: top-1 {ev[0] / ev.sum():.3f}, top-8 {ev[:8].sum() / ev.sum():.3f}, top-32 {ev[:32].sum() / ev.sum():.3f} of the variance")
    except Exception as ex:  # pragma: no cover
        print("encoder corpus skipped:", repr(ex))
