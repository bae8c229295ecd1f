"""Parity of the HIP exhaustive top-k search against the NumPy oracle (through the C ABI)."""
import numpy as np
import pytest

from oracle import search_oracle

pytestmark = pytest.mark.gpu


def _int_corpus(rng, n, d=128, lo=-4, hi=4):
    # integer-valued fp16: every partial sum is exact in fp32, so ids must match bit for bit
    return rng.integers(lo, hi + 1, (n, d)).astype(np.float16)


@pytest.mark.parametrize("n,nq,k", [(4096, 64, 80), (10000, 256, 80), (1000, 33, 80), (131, 7, 5),
                                    (50000, 600, 80), (128, 1, 1), (257, 300, 200)])
def test_integer_corpus_ids_identical(gpu_device, n, nq, k):
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(n * 7 + nq)
    xb = _int_corpus(rng, n)
    xq = _int_corpus(rng, nq)
    index = IndexFlatIP(128)
    index.add(xb)
    assert index.ntotal == n
    D, I = index.search(xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    st = index.last_stats()
    assert st["fallback_rounds"] == 0


def test_fewer_rows_than_k(gpu_device):
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(5)
    xb = _int_corpus(rng, 37)
    xq = _int_corpus(rng, 9)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, 80)
    Do, Io = search_oracle.topk_ip(xq, xb, 80)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert (I[:, 37:] == -1).all()


def test_empty_index_and_empty_queries(gpu_device):
    from proqa_amd.index import IndexFlatIP
    index = IndexFlatIP(128)
    xq = np.zeros((3, 128), np.float16)
    D, I = index.search(xq, 4)
    assert (I == -1).all() and (D == search_oracle.NEG_FILL).all()
    index.add(np.ones((10, 128), np.float16))
    D, I = index.search(np.zeros((0, 128), np.float16), 4)
    assert D.shape == (0, 4) and I.shape == (0, 4)


def test_all_ties_lowest_index_wins(gpu_device):
    from proqa_amd.index import IndexFlatIP
    xb = np.ones((5000, 128), np.float16)
    xq = np.ones((40, 128), np.float16)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, 80)
    np.testing.assert_array_equal(I, np.tile(np.arange(80), (40, 1)))
    assert (D == 128.0).all()


def test_random_fp16_matches_oracle(gpu_device):
    """Random normal fp16: scores agree to fp32 round-off, id sets agree except at round-off ties."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(0)
    xb = rng.standard_normal((10000, 128)).astype(np.float16)   # BASELINE.json configs[0]
    xq = rng.standard_normal((256, 128)).astype(np.float16)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, 80)
    Do, Io = search_oracle.topk_ip(xq, xb, 80)
    np.testing.assert_allclose(D, Do, rtol=1e-5, atol=1e-4)
    agree = np.mean([len(set(a) & set(b)) / 80.0 for a, b in zip(I, Io)])
    assert agree > 1 - 1e-4          # Recall@80 tolerance of BASELINE.json north_star
    assert (np.diff(D, axis=1) <= 0).all()


def test_adversarial_order_takes_overflow_safe_path(gpu_device):
    """Rows sorted by ascending score: every row beats the running threshold, candidate lists
    overflow, and the slab re-scan must still return the exact answer."""
    from proqa_amd.index import IndexFlatIP
    n = 40000
    base = np.zeros((n, 128), np.float16)
    base[:, 0] = (np.arange(n) // 40).astype(np.float16)      # non-decreasing scores, with ties
    xq = np.zeros((70, 128), np.float16)
    xq[:, 0] = 1
    index = IndexFlatIP(128)
    index.configure(first_slab_rows=128, growth=4)
    index.add(base)
    D, I = index.search(xq, 80)
    Do, Io = search_oracle.topk_ip(xq, base, 80)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert index.last_stats()["fallback_rounds"] > 0


def test_f32_inputs_and_incremental_add(gpu_device):
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(3)
    xb = _int_corpus(rng, 3000)
    xq = _int_corpus(rng, 50)
    index = IndexFlatIP(128)
    index.add(xb[:1000].astype(np.float32))     # eval_retrieval.py upcasts to float32 before add
    index.add(xb[1000:])
    D, I = index.search(xq.astype(np.float32), 10)
    Do, Io = search_oracle.topk_ip(xq, xb, 10)
    np.testing.assert_array_equal(I, Io)
    index.reset()
    assert index.ntotal == 0


def test_inexact_f32_is_refused_unless_allowed(gpu_device):
    """The index stores fp16: float32 values that would change are an error, not a silent rounding."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(5)
    xb = rng.standard_normal((700, 128)).astype(np.float32)       # not fp16-representable
    xq = rng.standard_normal((9, 128)).astype(np.float32)
    index = IndexFlatIP(128)
    with pytest.raises(RuntimeError, match="not exactly representable"):
        index.add(xb)
    assert index.ntotal == 0
    with pytest.raises(RuntimeError, match="not exactly representable"):
        index.add(torch.from_numpy(xb).to(gpu_device))
    assert index.ntotal == 0
    index.add(xb.astype(np.float16))
    with pytest.raises(RuntimeError, match="not exactly representable"):
        index.search(xq, 5)
    nanq = xq.astype(np.float16).astype(np.float32)
    index.search(nanq, 5)                                          # exact float32 queries are fine
    index.reset()
    index.allow_rounding(True)
    index.add(xb)
    D, I = index.search(xq, 5)
    Do, Io = search_oracle.topk_ip(xq.astype(np.float16), xb.astype(np.float16), 5)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_allclose(D, Do, rtol=2e-6, atol=2e-5)      # fp32 summation order only


def test_device_search_offsets_and_merge(gpu_device):
    """Two shards searched separately with global ids, merged on the GPU == unsharded search."""
    import torch
    from proqa_amd.index import IndexFlatIP, merge_topk_device
    rng = np.random.default_rng(11)
    xb = _int_corpus(rng, 9000)
    xq = _int_corpus(rng, 130)
    tq = torch.from_numpy(xq).to(gpu_device)
    parts = []
    for lo, hi in [(0, 4000), (4000, 9000)]:
        ix = IndexFlatIP(128)
        ix.add_device(torch.from_numpy(xb[lo:hi]).to(gpu_device))
        parts.append(ix.search_device(tq, 80, idx_offset=lo))
    D, I = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    Do, Io = search_oracle.topk_ip(xq, xb, 80)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)


@pytest.mark.parametrize("n_parts,k,nq", [(8, 1000, 37), (3, 5000, 21), (8, 10000, 9), (5, 7, 300)])
def test_merge_of_large_lists(gpu_device, n_parts, k, nq):
    """Sharded merge beyond one LDS pass (retrieval/trec_process.py:76 asks for k = 10000): tie-heavy
    integer scores, short shards padded with -1, against a NumPy merge of the same lists."""
    import torch
    from proqa_amd.index import merge_topk_device
    rng = np.random.default_rng(n_parts * 1000 + k)
    Dp = np.empty((n_parts, nq, k), np.float32)
    Ip = np.empty((n_parts, nq, k), np.int64)
    base = 0
    for p in range(n_parts):
        rows = k if p != 1 else k // 2                       # shard 1 holds fewer than k rows
        sc = np.sort(rng.integers(-50, 50, (nq, rows)).astype(np.float32), axis=1)[:, ::-1]
        ids = base + np.sort(rng.permuted(np.tile(np.arange(rows * 3), (nq, 1)), axis=1)[:, :rows], axis=1)
        # within equal scores a shard reports ascending ids (IndexFlatIP rule)
        order = np.lexsort((ids, -sc), axis=1)
        Dp[p, :, :rows] = np.take_along_axis(sc, order, 1)
        Ip[p, :, :rows] = np.take_along_axis(ids, order, 1)
        Dp[p, :, rows:] = np.finfo(np.float32).min
        Ip[p, :, rows:] = -1
        base += rows * 3
    D, I = merge_topk_device(torch.from_numpy(Dp).to(gpu_device), torch.from_numpy(Ip).to(gpu_device))
    D, I = D.cpu().numpy(), I.cpu().numpy()
    allD = Dp.transpose(1, 0, 2).reshape(nq, -1)
    allI = Ip.transpose(1, 0, 2).reshape(nq, -1)
    for q in range(nq):
        valid = allI[q] >= 0
        order = np.lexsort((allI[q][valid], -allD[q][valid]))[:k]
        np.testing.assert_array_equal(I[q, :len(order)], allI[q][valid][order])
        np.testing.assert_array_equal(D[q, :len(order)], allD[q][valid][order])
        assert (I[q, len(order):] == -1).all()


def test_full_size_properties(gpu_device):
    """18M x 128 fp16 (BASELINE.json configs[2] shape) through size-independent properties:
    planted rows must be found at rank 0, scores sorted, ids unique and in range, and a sharded
    search of the same tensor must give the identical result."""
    import torch
    from proqa_amd.index import IndexFlatIP, merge_topk_device
    n, nq, k = 18_000_000, 2032, 80
    g = torch.Generator(device=gpu_device)
    g.manual_seed(1234)
    xb = torch.empty((n, 128), dtype=torch.float16, device=gpu_device)
    step = 2_000_000
    for r0 in range(0, n, step):
        xb[r0:r0 + step] = torch.randn((min(step, n - r0), 128), generator=g, device=gpu_device,
                                       dtype=torch.float32).to(torch.float16)
    xq = torch.randn((nq, 128), generator=g, device=gpu_device, dtype=torch.float32).to(torch.float16)
    # plant: row p_j = 8 * xq_j is the unique best match of query j
    plant = torch.randperm(n, generator=g, device=gpu_device)[:nq]
    xb[plant] = (xq.float() * 8).to(torch.float16)
    index = IndexFlatIP(128)
    index.adopt_device(xb)
    D, I = index.search_device(xq, k)
    assert index.last_stats()["fallback_rounds"] == 0
    assert torch.equal(I[:, 0], plant)
    assert (D[:, 1:] <= D[:, :-1]).all()
    assert (I >= 0).all() and (I < n).all()
    assert all(len(set(row.tolist())) == k for row in I[:64].cpu())
    # re-score the reported ids in fp32 on the GPU: D must be the true inner products
    rows = xb[I[:32].reshape(-1)].float().reshape(32, k, 128)
    ref = torch.einsum("qkd,qd->qk", rows, xq[:32].float())
    assert torch.allclose(D[:32], ref, rtol=1e-5, atol=1e-3)
    # no row outside the list may beat the k-th score (checked exhaustively for 8 queries)
    S = xq[:8].float() @ xb[:6_000_000].float().T
    kth = D[:8, -1:]
    n_better = (S > kth).sum(dim=1)
    in_list = torch.stack([(I[q] < 6_000_000).sum() for q in range(8)])
    assert (n_better <= in_list).all()
    del S
    # sharded == unsharded, bit for bit
    parts = []
    for lo, hi in [(0, 7_000_000), (7_000_000, n)]:
        ix = IndexFlatIP(128)
        ix.adopt_device(xb[lo:hi])
        parts.append(ix.search_device(xq, k, idx_offset=lo))
    Dm, Im = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(Im, I) and torch.equal(Dm, D)


@pytest.mark.parametrize("n,nq,k", [(30000, 70, 2500), (20000, 300, 1025), (3000, 20, 10000), (60000, 33, 10000)])
def test_large_k_paged_search(gpu_device, n, nq, k):
    """k > 1024 (retrieval/trec_process.py:76 asks for k=10000) is served page by page; integer
    corpora are full of exact ties, also across page boundaries."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(n + k)
    xb = _int_corpus(rng, n, lo=-2, hi=2)
    xq = _int_corpus(rng, nq, lo=-2, hi=2)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_large_k_random_scores(gpu_device):
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(9)
    xb = rng.standard_normal((200000, 128)).astype(np.float16)
    xq = rng.standard_normal((40, 128)).astype(np.float16)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, 10000)
    Do, Io = search_oracle.topk_ip(xq, xb, 10000)
    np.testing.assert_allclose(D, Do, rtol=1e-5, atol=1e-4)
    assert (np.diff(D, axis=1) <= 0).all()
    assert all(len(set(r)) == 10000 for r in I)
    agree = np.mean([len(set(a) & set(b)) / 10000.0 for a, b in zip(I, Io)])
    assert agree > 1 - 1e-4


def test_randomised_shapes_against_oracle(gpu_device):
    """Seeded sweep over ragged shapes (rows not a multiple of the 128-row stage or the 32-row tile,
    query counts straddling the 256/512 tile sizes, k from 1 to several hundred, repeated searches on
    one handle with growing index and changing k): ids and scores bit-identical on integer data."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(2024)
    index = IndexFlatIP(128)
    xb_all = np.zeros((0, 128), np.float16)
    for trial in range(24):
        n_add = int(rng.choice([1, 31, 33, 127, 129, 500, 2049, 7777, 20000]))
        nq = int(rng.choice([1, 2, 31, 32, 33, 255, 256, 257, 511, 513, 700]))
        k = int(rng.choice([1, 2, 7, 80, 81, 200, 511, 700]))
        lo, hi = (-1, 1) if trial % 3 == 0 else (-4, 4)      # narrow range => heavy ties
        xb = _int_corpus(rng, n_add, lo=lo, hi=hi)
        xq = _int_corpus(rng, nq, lo=lo, hi=hi)
        if trial % 8 == 7:
            index.reset()
            xb_all = np.zeros((0, 128), np.float16)
        index.add(xb)
        xb_all = np.concatenate([xb_all, xb])
        assert index.ntotal == len(xb_all)
        D, I = index.search(xq, k)
        Do, Io = search_oracle.topk_ip(xq, xb_all, k)
        np.testing.assert_array_equal(I, Io, err_msg=f"trial {trial}: n={len(xb_all)} nq={nq} k={k}")
        np.testing.assert_array_equal(D, Do)
