"""NumPy oracle for the exhaustive inner-product top-k search — TEST INFRASTRUCTURE ONLY.

Restates /root/reference/retrieval/eval_retrieval.py:98-104:

    xq = np.load(query_embed).astype('float32'); xb = np.load(indexpath).astype('float32')
    index = faiss.IndexFlatIP(d); index.add(xb); D, I = index.search(xq, topk)

The arithmetic lives in faiss-cpu==1.6.3 (requirements.txt:2), which is not vendored in the
reference and not installed here, so its published semantics are restated: exact fp32 inner
products, top-k by DESCENDING score; D float32 [nq,k], I int64 [nq,k]; when fewer than k rows
exist the tail is I=-1, D=-FLT_MAX (faiss CMin neutral).  FAISS leaves the order of exactly
tied scores unspecified; this oracle (and the HIP path) pins it to ASCENDING row index.

Parity status: UNPINNED by the reference — it has no tests or golden vectors for this path
(SURVEY.md section 4 / 8c), and faiss itself cannot be run here.  The oracle is pinned instead
against brute-force definitions in tests/test_oracle_search.py and the committed SHA-256 of its
outputs on seeded inputs (tests/golden/search_golden.json).
"""
import numpy as np

NEG_FILL = np.float32(-3.4028234663852886e38)


def scores_f32(xq, xb):
    """The reference's score matrix: float32 upcast, float32 GEMM (eval_retrieval.py:99-100)."""
    return np.asarray(xq).astype(np.float32) @ np.asarray(xb).astype(np.float32).T


def _topk_rows(S, k, col_offset=0):
    """Exact top-k of each row of S by (score desc, column asc).  Returns (D, I) with I global."""
    nq, n = S.shape
    kk = min(k, n)
    if n > 4 * kk:
        # preselect: everything >= the kk-th largest value (ties included), then sort exactly
        kth = np.partition(S, n - kk, axis=1)[:, n - kk]
        D = np.empty((nq, kk), dtype=np.float32)
        I = np.empty((nq, kk), dtype=np.int64)
        for q in range(nq):
            cand = np.nonzero(S[q] >= kth[q])[0]
            order = np.lexsort((cand, -S[q, cand].astype(np.float64)))[:kk]
            I[q] = cand[order]
            D[q] = S[q, I[q]]
    else:
        I = np.argsort(-S.astype(np.float64), axis=1, kind="stable")[:, :kk].astype(np.int64)
        D = np.take_along_axis(S, I, axis=1).astype(np.float32)
    return D, I + col_offset


def merge_lists(D_parts, I_parts, k):
    """Merge per-part top lists [(nq,k_i)] into the global top-k with the same ordering rule."""
    D = np.concatenate(D_parts, axis=1)
    I = np.concatenate(I_parts, axis=1)
    nq = D.shape[0]
    outD = np.full((nq, k), NEG_FILL, dtype=np.float32)
    outI = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        valid = np.nonzero(I[q] >= 0)[0]
        order = valid[np.lexsort((I[q, valid], -D[q, valid].astype(np.float64)))][:k]
        outD[q, :len(order)] = D[q, order]
        outI[q, :len(order)] = I[q, order]
    return outD, outI


def topk_ip(xq, xb, k, block_rows=262144, query_block=256):
    """D, I = IndexFlatIP(d).add(xb).search(xq, k) restated in NumPy (blocked to bound memory)."""
    xq = np.asarray(xq)
    xb = np.asarray(xb)
    nq, n = xq.shape[0], xb.shape[0]
    outD = np.full((nq, k), NEG_FILL, dtype=np.float32)
    outI = np.full((nq, k), -1, dtype=np.int64)
    if nq == 0 or n == 0:
        return outD, outI
    xq32 = xq.astype(np.float32)
    for q0 in range(0, nq, query_block):
        q1 = min(nq, q0 + query_block)
        partsD, partsI = [], []
        for r0 in range(0, n, block_rows):
            r1 = min(n, r0 + block_rows)
            S = xq32[q0:q1] @ xb[r0:r1].astype(np.float32).T
            D, I = _topk_rows(S, k, col_offset=r0)
            partsD.append(D)
            partsI.append(I)
        D, I = merge_lists(partsD, partsI, k)
        outD[q0:q1] = D
        outI[q0:q1] = I
    return outD, outI


def topk_ip_argsort(xq, xb, k):
    """Smallest possible definition (full stable argsort) used to pin topk_ip itself."""
    S = scores_f32(xq, xb)
    n = S.shape[1]
    I = np.argsort(-S.astype(np.float64), axis=1, kind="stable")[:, :k].astype(np.int64)
    D = np.take_along_axis(S, I, axis=1).astype(np.float32)
    if n < k:
        pad = k - n
        D = np.concatenate([D, np.full((S.shape[0], pad), NEG_FILL, np.float32)], axis=1)
        I = np.concatenate([I, np.full((S.shape[0], pad), -1, np.int64)], axis=1)
    return D, I
