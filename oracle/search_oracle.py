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
    # invalid slots (I < 0) sort last; then (score desc, id asc) by two stable passes
    big = np.iinfo(np.int64).max
    ids = np.where(I >= 0, I, big)
    o = np.argsort(ids, axis=1, kind="stable")
    ids, Ds = np.take_along_axis(ids, o, axis=1), np.take_along_axis(D, o, axis=1)
    key = np.where(ids != big, -Ds.astype(np.float64), np.inf)
    o = np.argsort(key, axis=1, kind="stable")[:, :k]
    ids, Ds = np.take_along_axis(ids, o, axis=1), np.take_along_axis(Ds, o, axis=1)
    m = ids.shape[1]
    valid = ids != big
    outD[:, :m] = np.where(valid, Ds, NEG_FILL)
    outI[:, :m] = np.where(valid, ids, -1)
    return outD, outI


def topk_ip(xq, xb, k, block_rows=262144, query_block=256):
    """D, I = IndexFlatIP(d).add(xb).search(xq, k) restated in NumPy (blocked to bound memory).

    Like faiss' result heaps, a query keeps its running k-th best score: once k results exist, a
    block contributes only the scores that reach that threshold (one vectorised compare per block)."""
    xq = np.asarray(xq)
    xb = np.asarray(xb)
    nq, n = xq.shape[0], xb.shape[0]
    outD = np.full((nq, k), NEG_FILL, dtype=np.float32)
    outI = np.full((nq, k), -1, dtype=np.int64)
    if nq == 0 or n == 0:
        return outD, outI
    xq32 = xq.astype(np.float32)
    # a short first block fills the lists (full selection), every later block is threshold-filtered
    first = min(block_rows, max(4096, 4 * k))
    bounds = [0] + list(range(first, n, block_rows)) + [n] if n > first else [0, n]
    # score / mask buffers are allocated once and reused (fresh pages per block cost more than the sgemm)
    s_buf = np.empty((min(nq, query_block), min(n, max(first, block_rows))), dtype=np.float32)
    m_buf = np.empty(s_buf.shape, dtype=bool)
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        xb32 = np.asarray(xb[r0:r1], dtype=np.float32)  # the reference's upcast (eval_retrieval.py:99-100)
        for q0 in range(0, nq, query_block):
            q1 = min(nq, q0 + query_block)
            if q1 - q0 == s_buf.shape[0] and r1 - r0 == s_buf.shape[1]:
                S = np.matmul(xq32[q0:q1], xb32.T, out=s_buf)
            else:
                S = xq32[q0:q1] @ xb32.T
            if r0 < k:                                # fewer than k rows seen so far: no threshold yet
                D, I = _topk_rows(S, k, col_offset=r0)
                outD[q0:q1], outI[q0:q1] = merge_lists([outD[q0:q1], D], [outI[q0:q1], I], k)
                continue
            tau = outD[q0:q1, k - 1][:, None]
            mask = np.greater_equal(S, tau, out=m_buf) if S is s_buf else S >= tau
            flat = np.flatnonzero(mask)
            if len(flat) == 0:
                continue
            rows, cols = np.divmod(flat, S.shape[1])
            starts = np.searchsorted(rows, np.arange(q1 - q0 + 1))
            for q in np.unique(rows):
                c = cols[starts[q]:starts[q + 1]]
                D = np.concatenate([outD[q0 + q], S[q, c]])
                I = np.concatenate([outI[q0 + q], c + r0])
                order = np.lexsort((I, -D.astype(np.float64)))[:k]
                outD[q0 + q], outI[q0 + q] = D[order], I[order]
    return outD, outI


def topk_ip_threaded(xq, xb, k, workers=8, query_block=128, block_rows=262144):
    """topk_ip with the query blocks spread over a thread pool (one BLAS thread each) -- the shape of
    faiss' own CPU search (blocked sgemm + per-query heaps under OpenMP).  Same results as topk_ip;
    bench.py times this one as the CPU baseline so that the top-k selection is threaded too."""
    from concurrent.futures import ThreadPoolExecutor
    from threadpoolctl import threadpool_limits
    xq = np.asarray(xq)
    nq = xq.shape[0]
    outD = np.full((nq, k), NEG_FILL, dtype=np.float32)
    outI = np.full((nq, k), -1, dtype=np.int64)
    if nq == 0 or len(xb) == 0:
        return outD, outI
    blocks = [(q0, min(nq, q0 + query_block)) for q0 in range(0, nq, query_block)]
    xb = np.asarray(xb).astype(np.float32)            # upcast once, like eval_retrieval.py:100; shared by the workers

    def run(b):
        return topk_ip(xq[b[0]:b[1]], xb, k, block_rows=block_rows, query_block=query_block)

    with threadpool_limits(limits=1), ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        for b, (D, I) in zip(blocks, pool.map(run, blocks)):
            outD[b[0]:b[1]], outI[b[0]:b[1]] = D, I
    return outD, outI


def topk_ip_exact(xq, xb, k):
    """Order-independent float32 reference for float32 inputs: inner products accumulated in float64
    (products of two float32 are exact there), rounded once to float32, then the usual top-k rule.
    A float32 sgemm (faiss, eval_retrieval.py:102-104) returns these scores up to its own summation
    order, i.e. to ~1e-6 relative; this is what the HIP exact-float32 mode computes bit for bit."""
    xq = np.asarray(xq, dtype=np.float32)
    xb = np.asarray(xb, dtype=np.float32)
    outD = np.full((xq.shape[0], k), NEG_FILL, dtype=np.float32)
    outI = np.full((xq.shape[0], k), -1, dtype=np.int64)
    if xq.shape[0] == 0 or xb.shape[0] == 0:
        return outD, outI
    S = (xq.astype(np.float64) @ xb.astype(np.float64).T).astype(np.float32)
    D, I = _topk_rows(S, k)
    outD[:, :D.shape[1]], outI[:, :I.shape[1]] = D, I
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


def topk_ip_heap(xq, xb, k):
    """The rule for scores that are not finite, as faiss's result heap applies it (library knowledge, like the rest of the
    faiss semantics above: faiss-cpu 1.6.3 knn_inner_product pushes a score only `if (C::cmp(simi[0], ip))` with
    C = CMin<float, int64_t>, whose heap starts at neutral() = -FLT_MAX): a score enters only if it compares strictly
    greater than what the heap holds, so NaN (every comparison false), -inf and -FLT_MAX itself are never returned --
    their slots stay I = -1, D = -FLT_MAX -- and +inf ranks first.  For finite scores this is topk_ip_argsort."""
    with np.errstate(invalid="ignore", over="ignore"):
        S = scores_f32(xq, xb)
    nq, n = S.shape
    outD = np.full((nq, k), NEG_FILL, dtype=np.float32)
    outI = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        ok = np.nonzero(S[q] > NEG_FILL)[0]            # False for NaN
        order = np.lexsort((ok, -S[q, ok].astype(np.float64)))[:k]
        outI[q, :order.size] = ok[order]
        outD[q, :order.size] = S[q, ok[order]]
    return outD, outI

