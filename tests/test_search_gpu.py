"""Parity of the HIP exhaustive top-k search against the NumPy oracle (through the C ABI)."""
import os

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


def _f32_corpus(rng, n, scale=1.0):
    """float32 rows fp16 cannot hold: integers up to 2500 (odd values above 2048 are not fp16 numbers) --
    with small-integer-like queries every float64 partial sum is exact, so scores are order-independent."""
    return (rng.integers(-2500, 2501, (n, 128)) * scale).astype(np.float32)


def test_exact_float32_mode_matches_float64_oracle(gpu_device):
    """float32 inputs that are not fp16 numbers are searched exactly: fp16 scan with an error-bounded
    threshold + re-scoring from the float32 rows (never a silent rounding)."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(5)
    xb = _f32_corpus(rng, 30000)
    xq = (rng.integers(-3, 4, (70, 128)) * (1.0 + 2.0 ** -12)).astype(np.float32)    # not fp16 numbers either
    index = IndexFlatIP(128)
    index.add(xb[:10000].astype(np.float16).astype(np.float32))     # fp16-exact float32: still fp16 mode
    assert not index.exact_f32
    index.reset()
    index.add(xb[:10000])                                            # host path, switches on the first piece
    assert index.exact_f32 and index.ntotal == 10000
    index.add(torch.from_numpy(xb[10000:20000]).to(gpu_device))      # device path
    index.add(xb[20000:].astype(np.float16))                         # fp16 rows into an exact index
    ref_rows = np.concatenate([xb[:20000], xb[20000:].astype(np.float16).astype(np.float32)])
    for k in (1, 80, 300):
        D, I = index.search(xq, k)
        Do, Io = search_oracle.topk_ip_exact(xq, ref_rows, k)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
    assert index.last_stats()["fallback_rounds"] == 0
    # the rounded (fp16) answer differs: the exact mode is not a no-op on this data
    Dr, Ir = search_oracle.topk_ip(xq.astype(np.float16), ref_rows.astype(np.float16), 80)
    D, I = index.search(xq, 80)
    assert (Ir != I).any()
    # reset returns to fp16 mode; rounding on request gives exactly the rounded answer
    index.reset()
    assert not index.exact_f32
    index.allow_rounding(True)
    index.add(ref_rows)
    assert not index.exact_f32
    D, I = index.search(xq, 80)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)


def test_exact_float32_mode_random_data_and_query_switch(gpu_device):
    """Random float32: ids equal the float64-accumulated oracle; an fp16 index switches when a query needs it."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(6)
    xb = rng.standard_normal((120000, 128)).astype(np.float32)
    xq = rng.standard_normal((300, 128)).astype(np.float32)
    index = IndexFlatIP(128)
    index.add(xb)
    assert index.exact_f32
    D, I = index.search(xq, 80)
    Do, Io = search_oracle.topk_ip_exact(xq, xb, 80)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    # the float32-sgemm statement of the reference agrees to its own summation error
    Ds, Is = search_oracle.topk_ip(xq, xb, 80)
    assert (Is == I).mean() > 0.999
    np.testing.assert_allclose(Ds, D, rtol=2e-6, atol=2e-5)
    # fp16 rows, float32 queries that are not fp16 numbers: the index switches at search time
    index2 = IndexFlatIP(128)
    xb16 = xb[:50000].astype(np.float16)
    index2.add(torch.from_numpy(xb16).to(gpu_device))
    assert not index2.exact_f32
    D2, I2 = index2.search(xq, 10)
    assert index2.exact_f32
    Do2, Io2 = search_oracle.topk_ip_exact(xq, xb16.astype(np.float32), 10)
    np.testing.assert_array_equal(I2, Io2)
    np.testing.assert_array_equal(D2, Do2)


def test_exact_float32_mode_under_overflow_and_pages(gpu_device):
    """Rows that differ only below fp16 resolution: every row is inside the error margin, the lane lists
    overflow and the overflow-safe path must still order by the float32 values; plus k > 1024 (pages)."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(8)
    n = 150000
    xb = np.ones((n, 128), np.float32)
    # two coordinates perturbed below half an fp16 ulp: EVERY row rounds to the same fp16 vector
    xb[:, 0] = 1.0 + rng.integers(0, 2048, n).astype(np.float32) * 2.0 ** -23
    xb[:, 1] = 1.0 + rng.integers(0, 2048, n).astype(np.float32) * 2.0 ** -23
    xq = np.zeros((300, 128), np.float32)       # enough active lanes per wave to exhaust lists + spill
    xq[:, 0] = rng.choice([1.0, 2.0, 0.5, 1.0 + 2.0 ** -12, 3.0], 300)
    xq[:, 1] = rng.choice([1.0, 0.25, 1.5], 300)
    index = IndexFlatIP(128)
    index.add(xb)
    assert index.exact_f32
    D, I = index.search(xq, 100)
    Do, Io = search_oracle.topk_ip_exact(xq, xb, 100)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert index.last_stats()["fallback_rounds"] > 0
    # pages: k = 2500 over a float32 corpus fp16 cannot hold
    xb2 = _f32_corpus(rng, 9000)
    xq2 = rng.integers(-3, 4, (6, 128)).astype(np.float32)
    index.reset()
    index.add(xb2)
    D, I = index.search(xq2, 2500)
    Do, Io = search_oracle.topk_ip_exact(xq2, xb2, 2500)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_exact_float32_shards_and_npy_index(gpu_device, tmp_path):
    """Exact-float32 mode through the two ways an index arrives: row shards merged on the GPU, and a
    '<f4' .npy file through eval_retrieval.search (the reference upcasts whatever it loads to float32)."""
    import torch
    from proqa_amd import npy
    from proqa_amd.eval_retrieval import search
    from proqa_amd.index import IndexFlatIP, merge_topk_device
    rng = np.random.default_rng(12)
    xb = _f32_corpus(rng, 7000)
    xq = rng.integers(-3, 4, (40, 128)).astype(np.float32)
    Do, Io = search_oracle.topk_ip_exact(xq, xb, 80)
    tq = torch.from_numpy(xq).to(gpu_device)
    parts = []
    for lo, hi in [(0, 3000), (3000, 7000)]:
        ix = IndexFlatIP(128)
        ix.add(torch.from_numpy(xb[lo:hi]).to(gpu_device))
        assert ix.exact_f32
        parts.append(ix.search_device(tq, 80, idx_offset=lo))
    D, I = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)
    np.save(tmp_path / "index.npy", xb)
    np.save(tmp_path / "query.npy", xq)
    D, I = search(str(tmp_path / "index.npy"), str(tmp_path / "query.npy"), 80)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    Dr, Ir = search(str(tmp_path / "index.npy"), str(tmp_path / "query.npy"), 80, allow_rounding=True)
    Dh, Ih = search_oracle.topk_ip(xq.astype(np.float16), xb.astype(np.float16), 80)
    np.testing.assert_array_equal(Ir, Ih)


def test_randomised_parity_fuzz(gpu_device):
    """25 s of scripts/dev_fuzz_search.py: random sizes / batch sizes / k (pages) / tie-heavy, adversarially
    ordered and float32 corpora / host and device adds / shards, each bit-exact against the oracle.
    (This fuzz is what exposed a missing vmcnt wait before the LDS-DMA barrier of the filter kernel:
    cold-cache host uploads + small batches read a stage before it had landed.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "dev_fuzz_search.py"), "25", "7"],
                         capture_output=True, text=True, cwd=root, timeout=600)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_values_beyond_fp16_range_are_refused(gpu_device):
    from proqa_amd.index import IndexFlatIP
    xb = np.ones((300, 128), np.float32)
    xb[7, 3] = 1.0e6
    index = IndexFlatIP(128)
    with pytest.raises(RuntimeError, match="exceed the fp16 range"):
        index.add(xb)
    assert index.ntotal == 0


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
    # the headline's own path: the rounds ran on the int8 copy, no list overflowed, the scan stays on
    st = index.last_stats()
    assert st["nomination"] and st["nomination_state"] == "on" and st["fallback_rounds"] == 0 and st["nominated"] > 0
    assert torch.equal(I[:, 0], plant)
    assert (D[:, 1:] <= D[:, :-1]).all()
    assert (I >= 0).all() and (I < n).all()
    assert all(len(set(row.tolist())) == k for row in I[:64].cpu())
    # the fp16 scan of the same adopted rows: ids AND score bits of all 2032 x 80 results are equal
    index.configure_nomination("off")
    D16, I16 = index.search_device(xq, k)
    st16 = index.last_stats()
    assert not st16["nomination"] and st16["nomination_state"] == "off" and st16["fallback_rounds"] == 0
    assert torch.equal(I16, I) and torch.equal(D16.view(torch.int32), D.view(torch.int32))
    index.configure_nomination("auto")
    del D16, I16
    # re-score the reported ids in fp32 on the GPU: D must be the true inner products
    rows = xb[I[:32].reshape(-1)].float().reshape(32, k, 128)
    ref = torch.einsum("qkd,qd->qk", rows, xq[:32].float())
    assert torch.allclose(D[:32], ref, rtol=1e-5, atol=1e-3)
    # no row outside the list may beat the k-th score: exhaustively over ALL 18M rows for 8 queries, in 6M-row pieces
    # (S is a float32 GEMM with its own summation order: 1e-3 of slack on scores of ~30-45, fp32 round-off is ~1e-5)
    kth = D[:8, -1:]
    for p0 in range(0, n, 6_000_000):
        S = xq[:8].float() @ xb[p0:p0 + 6_000_000].float().T
        n_better = (S > kth + 1e-3).sum(dim=1)
        in_list = torch.stack([((I[q] >= p0) & (I[q] < p0 + 6_000_000)).sum() for q in range(8)])
        assert (n_better <= in_list).all(), p0
        # ... and every listed row of the piece really is at or above the k-th score
        assert ((S >= kth - 1e-3).sum(dim=1) >= in_list).all(), p0
        del S
    # sharded == unsharded, bit for bit
    parts = []
    for lo, hi in [(0, 7_000_000), (7_000_000, n)]:
        ix = IndexFlatIP(128)
        ix.adopt_device(xb[lo:hi])
        parts.append(ix.search_device(xq, k, idx_offset=lo))
    Dm, Im = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(Im, I) and torch.equal(Dm, D)


@pytest.mark.parametrize("n_parts,nq,k", [(2, 33, 80), (8, 257, 80), (3, 5, 1), (8, 9, 5000), (5, 21, 333)])
def test_strided_merge_of_a_gathered_buffer(gpu_device, n_parts, nq, k):
    """The receive buffer of the one all-gather -- per rank a block [ids int64 | scores float32 | pad to 16 B] -- merged
    where it lies (proqa_topk_merge_strided_device; what the RCCL paths do on more than one rank) equals the merge
    of the same parts copied into dense [n_parts, nq, k] arrays, and the oracle over the union."""
    import torch
    from proqa_amd import _lib
    from proqa_amd.index import merge_topk_device
    rng = np.random.default_rng(n_parts * 1000 + k)
    # per-part lists as a shard would report them: scores descending with ties, global ids ascending with the part
    Dp = np.sort(rng.integers(-50, 50, (n_parts, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    Ip = np.stack([np.sort(rng.permutation(100000)[:nq * k].reshape(nq, k), axis=1) + 100000 * p for p in range(n_parts)])
    for p in range(n_parts):          # (score desc, id asc) inside every list
        for q in range(nq):
            order = np.lexsort((Ip[p, q], -Dp[p, q]))
            Dp[p, q], Ip[p, q] = Dp[p, q][order], Ip[p, q][order]
    n_i, n_d = nq * k * 8, nq * k * 4
    block = (n_i + n_d + 15) // 16 * 16
    buf = np.zeros((n_parts, block), np.uint8)
    for p in range(n_parts):
        buf[p, :n_i] = Ip[p].astype(np.int64).view(np.uint8).reshape(-1)
        buf[p, n_i:n_i + n_d] = Dp[p].view(np.uint8).reshape(-1)
    g = torch.from_numpy(buf).to(gpu_device)
    D = torch.empty((nq, k), dtype=torch.float32, device=gpu_device)
    I = torch.empty((nq, k), dtype=torch.int64, device=gpu_device)
    lib = _lib.load()
    _lib.check(lib.proqa_topk_merge_strided_device(g.data_ptr() + n_i, g.data_ptr(), n_parts, nq, k, block // 4, block // 8,
                                                   D.data_ptr(), I.data_ptr(), _lib.current_stream_ptr()))
    Dd, Id = merge_topk_device(torch.from_numpy(Dp).to(gpu_device), torch.from_numpy(Ip.astype(np.int64)).to(gpu_device))
    assert torch.equal(D, Dd) and torch.equal(I, Id)
    allD = Dp.transpose(1, 0, 2).reshape(nq, -1)
    allI = Ip.transpose(1, 0, 2).reshape(nq, -1)
    for q in range(nq):
        order = np.lexsort((allI[q], -allD[q]))[:k]
        np.testing.assert_array_equal(I[q].cpu().numpy(), allI[q][order])
        np.testing.assert_array_equal(D[q].cpu().numpy(), allD[q][order])


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


@pytest.mark.parametrize("n,nq,k", [(300000, 40, 10000), (300000, 3, 5000), (200000, 300, 1500), (400000, 1, 5000),
                                    (250000, 70, 3000), (500000, 9, 11000), (200000, 200, 1000), (100000, 33, 700),
                                    (300000, 256, 1024), (300000, 700, 10000), (150000, 513, 4000)])
def test_large_k_one_pass_is_exact(gpu_device, n, nq, k):
    """1024 < k <= ~11700 (from ~670 for batches of <= 256 queries) on a shard much larger than k goes through ONE filter launch against thresholds estimated
    from a sample (search_one_pass, mips_index.cpp).  Integer data in [-8, 8]: scores are exact, a score level holds
    ~100 rows (ties across the k-th place are the rule), and ids must match the oracle bit for bit.  Batches of more than
    256 queries are dense enough here for the compact lists of 8-byte keys (mips_filter_f16<COMPACT>); the smaller batches
    log column records."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(n + k)
    xb = _int_corpus(rng, n, lo=-8, hi=8)
    xq = _int_corpus(rng, nq, lo=-8, hi=8)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, k)
    st = index.last_stats()
    assert st["fallback_rounds"] == 0 and st["rounds"] < 40, st      # the estimate held: no paging
    assert st["candidates"] / nq < (4 if k > 1024 else 8) * k
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_large_k_batches_beyond_the_store_budget_run_in_groups(gpu_device, monkeypatch):
    """Thousands of queries x a large k can need more candidate-list memory in one launch than the budget allows: the
    search runs in groups of whole query tiles instead (here: a 128 MB budget, 1300 queries at 170 KB of compact lists
    each -> groups of 512)."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(5)
    xb = _int_corpus(rng, 150000, lo=-8, hi=8)
    xq = _int_corpus(rng, 1300, lo=-8, hi=8)
    index = IndexFlatIP(128)
    index.add(xb)
    monkeypatch.setenv("PROQA_ONE_PASS_STORE_MB", "128")
    D, I = index.search(xq, 5000)
    st = index.last_stats()
    assert st["fallback_rounds"] == 0 and st["rounds"] >= 3 * 5, st
    Do, Io = search_oracle.topk_ip(xq, xb, 5000)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_large_k_one_pass_falls_back_on_an_ordered_corpus(gpu_device):
    """Rows sorted by their score against the first query: whatever the sample sees of them misjudges that query's
    threshold (too tight: fewer than k rows pass; too loose: the lists overflow).  The search notices and repeats
    page by page; the result is exact either way."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(77)
    xb = _int_corpus(rng, 200000, lo=-8, hi=8)
    xq = _int_corpus(rng, 5, lo=-8, hi=8)
    order = np.argsort(xb.astype(np.float32) @ xq[0].astype(np.float32), kind="stable")
    for name, rows in (("ascending", xb[order]), ("descending", xb[order[::-1]])):
        index = IndexFlatIP(128)
        index.add(rows)
        D, I = index.search(xq, 4000)
        Do, Io = search_oracle.topk_ip(xq, rows, 4000)
        np.testing.assert_array_equal(I, Io, err_msg=name)
        np.testing.assert_array_equal(D, Do, err_msg=name)
    assert index.last_stats()["fallback_rounds"] > 0   # descending: the head of the shard holds every good row
    # the same with a batch that takes the compact lists (more than 256 queries): in the descending order the lane lists of
    # the head chunks overflow for the first query (70 keys each), the launch reports it, the pages repeat the search
    xq2 = np.concatenate([xq[:1], _int_corpus(rng, 299, lo=-8, hi=8)])
    index = IndexFlatIP(128)
    index.add(xb[order[::-1]])
    D, I = index.search(xq2, 3000)
    Do, Io = search_oracle.topk_ip(xq2, xb[order[::-1]], 3000)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert index.last_stats()["fallback_rounds"] > 0


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


def test_reconstruct_batch_returns_the_rows_that_were_added(gpu_device):
    """proqa_index_reconstruct_batch_device: rows by id, fp16 and float32, global ids of a shard, -1 -> zero row; an
    exact-float32 index gives back its float32 values."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(3)
    xb = rng.standard_normal((3000, 128)).astype(np.float16)
    index = IndexFlatIP(128)
    index.add(xb)
    ids = torch.from_numpy(rng.integers(0, 3000, (7, 45))).to(gpu_device)
    ids[2, 3] = -1
    got16 = index.reconstruct_batch_device(ids).cpu().numpy()
    got32 = index.reconstruct_batch_device(ids, torch.float32).cpu().numpy()
    ref = xb[ids.cpu().numpy().clip(0)]
    ref[2, 3] = 0
    assert got16.dtype == np.float16 and got16.shape == (7, 45, 128)
    np.testing.assert_array_equal(got16, ref)
    np.testing.assert_array_equal(got32, ref.astype(np.float32))
    off = index.reconstruct_batch_device(ids + 1000, idx_offset=1000).cpu().numpy()    # a shard's global ids
    np.testing.assert_array_equal(off, ref)                                             # id 999 lies below the shard: zero row
    x32 = (rng.standard_normal((500, 128)) * (1 + 2.0 ** -14)).astype(np.float32)       # not representable in fp16
    exact = IndexFlatIP(128)
    exact.add(x32)
    assert exact.exact_f32
    pick = torch.tensor([0, 499, 17], device=gpu_device)
    np.testing.assert_array_equal(exact.reconstruct_batch_device(pick, torch.float32).cpu().numpy(), x32[[0, 499, 17]])
    with pytest.raises(ValueError):
        index.reconstruct_batch_device(ids.to(torch.int32))


def test_online_retriever_is_the_exact_search(gpu_device):
    """qa/online_sampler.py's per-question retrieval (k = 5000) on the exact index."""
    import torch
    from proqa_amd.online_retriever import OnlineRetriever
    rng = np.random.default_rng(21)
    xb = _int_corpus(rng, 9000)
    idmap = {str(i): f"doc-{i}" for i in range(len(xb))}
    r = OnlineRetriever(xb, idmap, device=gpu_device)
    q = _int_corpus(rng, 1)
    Do, Io = search_oracle.topk_ip(q, xb, 5000)
    for q_in in (q, torch.from_numpy(q).to(gpu_device)):
        idx, ids, emb = r.retrieve(q_in, 5000)
        np.testing.assert_array_equal(idx, Io[0])
        assert ids[:3] == [f"doc-{i}" for i in Io[0][:3]] and len(ids) == 5000
        np.testing.assert_array_equal(emb, xb[Io[0]])
    r_list = OnlineRetriever(xb, [f"doc-{i}" for i in range(len(xb))], device=gpu_device)   # row-ordered ids instead of the dict
    assert r_list.retrieve(q, 5000)[1] == ids
    r32 = OnlineRetriever(xb.astype(np.float32), None, device=gpu_device)                   # the reference's float32 array
    idx32, none, emb32 = r32.retrieve(q.astype(np.float32), 5000)
    assert none is None and emb32.dtype == np.float32
    np.testing.assert_array_equal(idx32, Io[0])
    np.testing.assert_array_equal(emb32, xb[Io[0]].astype(np.float32))
    idx, ids, emb = r.retrieve(q, 20000)                      # more than the index holds
    assert len(idx) == 9000 and emb.shape == (9000, 128)


def test_integration_md_ctypes_stub_runs(gpu_device):
    """The C-ABI binding printed in INTEGRATION.md section 3 is executed verbatim (in a fresh process, from
    the repo root) and must return the oracle's result."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"## 3\..*?```python\n(.*?)```", md, re.S).group(1)
    prog = block + '''
import numpy as np, sys
sys.path.insert(0, ".")
from oracle import search_oracle
rng = np.random.default_rng(0)
xb = rng.integers(-4, 5, (3000, 128)).astype(np.float16); xq = rng.integers(-4, 5, (17, 128)).astype(np.float16)
D, I = search(xb, xq, 80)
Do, Io = search_oracle.topk_ip(xq, xb, 80)
assert (I == Io).all() and (D == Do).all()
print("stub ok")
'''
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0 and "stub ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_exact_mode_switch_in_a_later_upload_piece(gpu_device):
    """Host uploads go in pieces of 2^20 rows: the first piece is fp16-exact, the switch to exact-float32 mode
    happens in the second piece, and the rows of the first piece must get their float32 copies too."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(31)
    n = (1 << 20) + 70000
    xb = rng.integers(-4, 5, (n, 128)).astype(np.float32)
    xb[(1 << 20) + 123:, 0] = 2049.0                       # not an fp16 number
    xb[: 1 << 20, 0] = rng.integers(-2000, 2001, 1 << 20)   # fp16-exact, large enough to compete
    xq = rng.integers(-3, 4, (9, 128)).astype(np.float32)
    xq[:, 0] = 1.0
    index = IndexFlatIP(128)
    index.add(xb)
    assert index.exact_f32 and index.ntotal == n
    D, I = index.search(xq, 50)
    Do, Io = search_oracle.topk_ip_exact(xq, xb, 50)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_exact_index_refilled_after_reset_by_a_larger_inexact_file(gpu_device, tmp_path):
    """reset() keeps the float32 buffer of the exact episode before it.  A later add_npy of a LARGER float32 file whose first
    inexact value comes late grows that buffer in the middle of the call: the float32 copies of the pieces uploaded before
    must move along (they used to be dropped: idx->n counts completed calls only), or the re-scoring reads garbage rows."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(57)
    index = IndexFlatIP(128)
    first = rng.integers(-4, 5, (20000, 128)).astype(np.float32)
    first[7, 3] = 1.0 + 2.0 ** -12                             # not an fp16 number: exact mode, a 20000-row float32 buffer
    index.add(first)
    assert index.exact_f32
    index.reset()
    n = 200000                                                 # > the old buffer; pieces of 16384 rows
    xb = rng.integers(-4, 5, (n, 128)).astype(np.float32)
    xb[:, 0] = rng.integers(-2000, 2001, n)
    xb[150000:, 1] += np.float32(2.0 ** -12)                   # the first inexact values sit in a late piece
    path = tmp_path / "rows.npy"
    np.save(path, xb)
    index.add_npy(str(path))
    assert index.exact_f32 and index.ntotal == n
    xq = rng.integers(-3, 4, (11, 128)).astype(np.float32)
    xq[:, 0] = 1.0
    D, I = index.search(xq, 40)
    Do, Io = search_oracle.topk_ip_exact(xq, xb, 40)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    # the same through add() of a host array (1M-row pieces): the late switch inside one call
    index.reset()
    index.add(xb)
    D, I = index.search(xq, 40)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_exact_mode_on_an_adopted_shard(gpu_device):
    """An adopted (caller-owned, fp16) shard searched with float32 queries fp16 cannot hold: the index builds
    its float32 copies next to the adopted rows."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(33)
    xb = _int_corpus(rng, 20000)
    xq = (rng.integers(-3, 4, (12, 128)) * (1.0 + 2.0 ** -12)).astype(np.float32)
    t = torch.from_numpy(xb).to(gpu_device)
    index = IndexFlatIP(128)
    index.adopt_device(t)
    D, I = index.search(xq, 80)
    assert index.exact_f32
    Do, Io = search_oracle.topk_ip_exact(xq, xb.astype(np.float32), 80)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    index.reset()
    assert index.ntotal == 0 and not index.exact_f32


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,k,rows", [(4096, 64, 80, 512), (20000, 300, 5, 4096), (9000, 33, 128, 1024),
                                         (40000, 2032, 80, 8192), (1024, 7, 8, 32)])
def test_bootstrap_rows_give_the_same_result(gpu_device, n, nq, k, rows):
    """The dense bootstrap (score matrix of the first rows + per-query select) replaces the first rounds:
    ids and scores are bit-identical with it on (any size) and off, and equal to the oracle."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(n + nq)
    xb = _int_corpus(rng, n)
    xq = _int_corpus(rng, nq)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    for r in (rows, 0):
        index = IndexFlatIP(128)
        index.configure_bootstrap(r)
        index.add(xb)
        D, I = index.search(xq, k)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
        assert index.last_stats()["fallback_rounds"] == 0
    # random fp16 data: a row's score has the same bits from the bootstrap and from the filter kernel
    xb = rng.standard_normal((n, 128)).astype(np.float16)
    xq = rng.standard_normal((nq, 128)).astype(np.float16)
    res = []
    for r in (rows, 0):
        index = IndexFlatIP(128)
        index.configure_bootstrap(r)
        index.add(xb)
        res.append(index.search(xq, k))
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][0], res[1][0])


@pytest.mark.gpu
def test_bootstrap_overflow_repeats_without_it(gpu_device):
    """The select keeps the keys above the k-th largest of its 256 thread maxima (thread t owns rows 4t .. 4t+3, + 1024, ...).
    If the good rows all sit in the strides of k-1 threads, that bound is low and (k-1)*rows/256 + 1 keys pass: more
    than one sort holds at 8192 rows.  The page is then repeated without the bootstrap."""
    from proqa_amd.index import IndexFlatIP
    n, nq, k, rows = 40000, 5, 80, 8192
    xb = np.zeros((n, 128), dtype=np.float16)
    xb[:, 0] = 1.0
    good = ((np.arange(n) // 4) % 256 < k - 1) & (np.arange(n) < rows)
    xb[good, 0] = 10.0
    xb[:, 1] = (np.arange(n) % 7).astype(np.float16)       # some variety below the good rows
    xq = np.zeros((nq, 128), dtype=np.float16)
    xq[:, 0] = 1.0
    xq[:, 1] = 0.125
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    index = IndexFlatIP(128)
    index.configure_bootstrap(rows)
    index.add(xb)
    D, I = index.search(xq, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert index.last_stats()["fallback_rounds"] >= 1


@pytest.mark.gpu
def test_overflow_of_a_large_slab_costs_a_bounded_rescan(gpu_device):
    """One query whose scores rise with the row number overflows its lists in the big late slabs.  The overflow-safe
    path re-scans such a slab in quarters (recursively) instead of thousands of dense 1920-row launches: the result is
    exact and the number of extra launches stays small for the queries that are not adversarial."""
    import time
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(5)
    n, nq, k = 1_000_000, 40, 80
    xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)
    xb[:, 0] = (np.arange(n) // 2000).astype(np.float16)          # 0 .. 499, exact in fp16
    xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
    xq[:, 0] = 0
    xq[0] = 0
    xq[0, 0] = 1                                                  # query 0: score = row // 2000, rising with the row
    index = IndexFlatIP(128)
    index.add(xb)
    index.search(xq, k)
    t0 = time.perf_counter()
    D, I = index.search(xq, k)
    dt = time.perf_counter() - t0
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    st = index.last_stats()
    assert st["fallback_rounds"] > 0
    assert st["fallback_rounds"] < 600 and dt < 0.5, (st, dt)


@pytest.mark.gpu
def test_begin_finish_equals_the_one_call_search(gpu_device):
    """proqa_index_search_begin_device / _finish: the search is enqueued, other work goes onto the stream behind it, the
    host checks once at the end.  Same result as the one-call form -- also when the check finds overflowed rounds and
    rewrites the result (status word 1), and for searches that cannot be deferred (large k: they complete in _begin)."""
    import ctypes
    import torch
    from proqa_amd import _lib
    from proqa_amd.index import IndexFlatIP
    lib = _lib.load()
    rng = np.random.default_rng(21)
    cases = []
    xb = _int_corpus(rng, 60000)
    cases.append(("plain", xb, _int_corpus(rng, 300), 80, None))
    cases.append(("small batch", xb, _int_corpus(rng, 5), 10, None))
    cases.append(("large k", _int_corpus(rng, 100000, lo=-8, hi=8), _int_corpus(rng, 7, lo=-8, hi=8), 3000, None))
    adv = np.zeros((40000, 128), np.float16)
    adv[:, 0] = (np.arange(40000) // 40).astype(np.float16)
    advq = np.zeros((70, 128), np.float16)
    advq[:, 0] = 1
    cases.append(("overflow", adv, advq, 80, (128, 4)))
    for name, xb_c, xq_c, k, cfg in cases:
        index = IndexFlatIP(128)
        if cfg:
            index.configure(*cfg)
        index.add(xb_c)
        xq_dev = torch.from_numpy(xq_c).cuda()
        nq = xq_c.shape[0]
        D = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
        I = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
        status = torch.full((4,), 77, dtype=torch.int32, device="cuda")
        _lib.check(lib.proqa_index_search_begin_device(index._h, xq_dev.data_ptr(), nq, 0, k, 1000, D.data_ptr(), I.data_ptr(),
                                                       status.data_ptr(), _lib.current_stream_ptr()))
        early = I.clone()                      # work enqueued behind the search sees the (optimistic) result
        status_seen = status.clone()
        rewritten = ctypes.c_int(-1)
        _lib.check(lib.proqa_index_search_finish(index._h, ctypes.byref(rewritten)))
        Do, Io = search_oracle.topk_ip(xq_c, xb_c, k)
        np.testing.assert_array_equal(I.cpu().numpy(), np.where(Io >= 0, Io + 1000, Io), err_msg=name)
        np.testing.assert_array_equal(D.cpu().numpy(), Do, err_msg=name)
        st = index.last_stats()
        assert status_seen[0].item() == rewritten.value == (1 if name == "overflow" else 0), (name, status_seen, rewritten.value)
        assert (st["fallback_rounds"] > 0) == (name == "overflow"), (name, st)
        if name != "overflow":
            assert torch.equal(early, I), name
        assert st["rounds"] > 0 and st["candidates"] > 0, (name, st)
        # a second _finish is a no-op; the one-call search still works on the handle
        _lib.check(lib.proqa_index_search_finish(index._h, None))
        D1, I1 = index.search(xq_c, k)
        np.testing.assert_array_equal(I1, Io, err_msg=name)
        index.close()


@pytest.mark.gpu
@pytest.mark.parametrize("block_rows", [100000, 10000])
def test_large_k_on_a_corpus_ordered_by_document(gpu_device, block_rows):
    """Real corpora are ordered by document: a query's top-k rows sit in a few contiguous stretches, not spread evenly over the
    shard as the sampled thresholds and the compact 64-key lists of the dense one-pass launch assume.  The launch is repeated
    once over four times the chunks when its lists wrap; where the sample itself misjudges the thresholds (a topic the sample
    slabs do not touch) the search goes page by page.  Exact either way -- that is what this test pins."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(block_rows)
    n, nq, k = 400000, 300, 1500
    n_blocks = n // block_rows
    topics = rng.integers(-2, 3, (n_blocks, 128)).astype(np.float32)
    xb = (np.repeat(topics, block_rows, axis=0) + rng.integers(-1, 2, (n, 128))).astype(np.float16)
    xq = (topics[rng.integers(0, n_blocks, nq)] + rng.integers(-1, 2, (nq, 128))).astype(np.float16)
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


@pytest.mark.gpu
def test_one_pass_store_that_cannot_be_allocated_falls_back_to_pages(gpu_device, monkeypatch):
    """The deep candidate store of the one-pass large-k search does not fit beside the caller's tensors (forced here: a
    hipMalloc that really fails, leaving HIP's sticky error behind): the search must fall back to the paged path and
    return the exact result, not report the allocation failure of a launch that never needed the memory."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(9)
    xb = _int_corpus(rng, 300000, lo=-8, hi=8)
    xq = _int_corpus(rng, 6, lo=-8, hi=8)
    Do, Io = search_oracle.topk_ip(xq, xb, 5000)
    index = IndexFlatIP(128)
    index.add(xb)
    monkeypatch.setenv("PROQA_DEBUG_STORE_LIMIT_MB", "400")
    D, I = index.search(xq, 5000)
    st = index.last_stats()
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert st["fallback_rounds"] == 1 and st["rounds"] > 20, st      # the one pass gave up (counted once), then pages
    monkeypatch.delenv("PROQA_DEBUG_STORE_LIMIT_MB")
    D, I = index.search(xq, 5000)                                     # and the one pass once the memory is there
    np.testing.assert_array_equal(I, Io)
    assert index.last_stats()["rounds"] <= 12
    index.close()


@pytest.mark.gpu
def test_candidate_store_follows_the_batch_not_the_largest_batch_seen(gpu_device):
    """A one-question large-k search after a big batch on the same handle: the deep lane lists are sized (and addressed)
    by the launch's own padded query count, so the store stays within what the one-pass plan budgeted."""
    import torch
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(12)
    xb = _int_corpus(rng, 400000, lo=-8, hi=8)
    index = IndexFlatIP(128)
    index.add(xb)
    big = _int_corpus(rng, 3000, lo=-8, hi=8)
    D, I = index.search(big, 80)
    Do, Io = search_oracle.topk_ip(big[:50], xb, 80)
    np.testing.assert_array_equal(I[:50], Io)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    one = _int_corpus(rng, 1, lo=-8, hi=8)
    D, I = index.search(one, 5000)
    Do, Io = search_oracle.topk_ip(one, xb, 5000)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    used = free0 - torch.cuda.mem_get_info()[0]
    # 256 padded queries x ~330 chunks x 2 lists x 24 records x 80 B ~ 0.33 GB; sized by the 3072-query workspace it was ~4 GB
    assert used < 1.5 * (1 << 30), used
    index.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,k", [(30000, 64, 80), (30000, 600, 80), (100, 9, 80), (70000, 40, 2000)])
def test_non_finite_scores_follow_the_heap_rule(gpu_device, n, nq, k):
    """+inf ranks first (lowest row first among several); NaN and -inf scores are never returned, as in faiss's heap --
    with fewer than k comparable rows the tail is I = -1, D = -FLT_MAX."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(n + nq)
    xb = _int_corpus(rng, n)           # integer-valued: finite scores are exact, ids must match bit for bit
    xq = _int_corpus(rng, nq)
    xq[:, 0] = np.where(np.arange(nq) % 2 == 0, 1.0, -1.0).astype(np.float16)   # sign decides +inf / -inf per query
    xq[:, 1] = 1.0
    plant = rng.choice(n, size=min(40, n // 2), replace=False)
    for i, r in enumerate(plant):
        kind = i % 4
        if kind == 0:
            xb[r, 0] = np.inf
        elif kind == 1:
            xb[r, 0] = -np.inf
        elif kind == 2:
            xb[r, 1] = np.nan
        else:
            xb[r, 0] = np.inf
            xb[r, 1] = -np.inf                      # NaN for the even queries, -inf for the odd ones
    index = IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, k)
    Do, Io = search_oracle.topk_ip_heap(xq, xb, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert not np.isnan(D).any() and np.isposinf(D[:, 0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("bounds,k", [([(0, 60), (60, 5000), (5000, 9000)], 80), ([(0, 3000), (3000, 9000)], 300)])
def test_shards_with_non_finite_scores_merge_like_the_unsharded_search(gpu_device, bounds, k):
    """The rank merge of per-shard lists with +inf scores, excluded NaN / -inf rows and a shard shorter than k (its list ends
    in I = -1 slots) equals the heap rule applied to the whole corpus."""
    import torch
    from proqa_amd.index import IndexFlatIP, merge_topk_device
    rng = np.random.default_rng(len(bounds) * 100 + k)
    xb = _int_corpus(rng, 9000)
    xq = _int_corpus(rng, 70)
    xq[:, 0] = np.where(np.arange(70) % 2 == 0, 1.0, -1.0).astype(np.float16)
    xq[:, 1] = 1.0
    for i, r in enumerate(rng.choice(9000, size=60, replace=False)):
        if i % 3 == 0:
            xb[r, 0] = np.inf
        elif i % 3 == 1:
            xb[r, 1] = np.nan
        else:
            xb[r, 0] = -np.inf
    xb[5, 0] = np.inf       # the short first shard holds one of each kind
    xb[6, 1] = np.nan
    tq = torch.from_numpy(xq).to(gpu_device)
    parts = []
    for lo, hi in bounds:
        ix = IndexFlatIP(128)
        ix.add_device(torch.from_numpy(xb[lo:hi]).to(gpu_device))
        parts.append(ix.search_device(tq, k, idx_offset=lo))
    D, I = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    Do, Io = search_oracle.topk_ip_heap(xq, xb, k)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)



def test_add_npy_streams_the_file_like_np_load_plus_add(gpu_device, tmp_path):
    """proqa_index_add_npy (reader threads -> pinned ring -> HBM) == np.load + add: fp16 files over several ring turns,
    row ranges (what a rank of a sharded index loads), appends, one reader and many, a float32 file whose switch to
    exact-float32 mode happens in a later piece, and the error cases."""
    from proqa_amd.index import IndexFlatIP
    rng = np.random.default_rng(41)
    n = 5 * 131072 + 777                                    # five 32 MiB pieces and a tail: the 4-slot ring wraps
    xb = rng.integers(-4, 5, (n, 128)).astype(np.float16)
    xq = rng.integers(-4, 5, (33, 128)).astype(np.float16)
    np.save(tmp_path / "f16.npy", xb)
    Do, Io = search_oracle.topk_ip(xq, xb, 80)
    for readers in (1, 0, 7):
        index = IndexFlatIP(128)
        index.add_npy(tmp_path / "f16.npy", readers=readers)
        assert index.ntotal == n and not index.exact_f32
        D, I = index.search(xq, 80)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
        index.close()
    # a row range, then an append of another range: rows [lo, hi) ++ [0, 1000)
    lo, hi = 200001, 600000
    index = IndexFlatIP(128)
    index.add_npy(tmp_path / "f16.npy", lo, hi - lo)
    index.add_npy(tmp_path / "f16.npy", 0, 1000)
    sub = np.concatenate([xb[lo:hi], xb[:1000]])
    D, I = index.search(xq, 80)
    Ds, Is = search_oracle.topk_ip(xq, sub, 80)
    np.testing.assert_array_equal(I, Is)
    np.testing.assert_array_equal(D, Ds)
    # errors leave the index as it was
    for bad in [(n - 5, 6), (-1, 3)]:
        with pytest.raises(Exception):
            index.add_npy(tmp_path / "f16.npy", *bad)
    with pytest.raises(Exception):
        index.add_npy(tmp_path / "missing.npy")
    np.save(tmp_path / "narrow.npy", np.zeros((10, 64), np.float16))
    with pytest.raises(Exception):
        index.add_npy(tmp_path / "narrow.npy")
    assert index.ntotal == hi - lo + 1000
    index.close()
    # float32 file: pieces of 65536 rows; the first two are fp16-exact, the switch happens in the third
    m = 2 * 65536 + 5000
    xf = rng.integers(-4, 5, (m, 128)).astype(np.float32)
    xf[: 2 * 65536, 0] = rng.integers(-2000, 2001, 2 * 65536)
    xf[2 * 65536 + 99:, 0] = 2049.0                         # not an fp16 number
    qf = rng.integers(-3, 4, (9, 128)).astype(np.float32)
    qf[:, 0] = 1.0
    np.save(tmp_path / "f32.npy", xf)
    index = IndexFlatIP(128)
    index.add_npy(tmp_path / "f32.npy")
    assert index.exact_f32 and index.ntotal == m
    D, I = index.search(qf, 50)
    De, Ie = search_oracle.topk_ip_exact(qf, xf, 50)
    np.testing.assert_array_equal(I, Ie)
    np.testing.assert_array_equal(D, De)
    index.close()


@pytest.mark.gpu
def test_pipelined_searcher_returns_the_one_call_results_in_order(gpu_device):
    """PipelinedSearcher (two handles over the same rows, two streams, the next batch enqueued before the host waits for the
    current one): every batch's result equals the oracle's / the one-call search's, in order -- batches of different sizes,
    an empty stream, a single batch, k beyond the deferred kind (it completes inside begin)."""
    import torch
    from proqa_amd.index import IndexFlatIP, PipelinedSearcher
    rng = np.random.default_rng(77)
    xb = _int_corpus(rng, 90000)
    t = torch.from_numpy(xb).cuda()
    ps = PipelinedSearcher(t)
    assert list(ps.search_batches([], 10)) == []
    sizes = [300, 1, 64, 300, 700, 5, 300]
    batches = [_int_corpus(rng, m) for m in sizes]
    got = list(ps.search_batches([torch.from_numpy(b).cuda() for b in batches], 80, idx_offset=7))
    assert len(got) == len(batches)
    for b, (D, I) in zip(batches, got):
        Do, Io = search_oracle.topk_ip(b, xb, 80)
        np.testing.assert_array_equal(I.cpu().numpy(), Io + 7)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
    assert all(s_["nomination"] for s_ in ps.last_stats())       # both handles scanned their own int8 copy
    (D, I), = list(ps.search_batches([torch.from_numpy(batches[0]).cuda()], 2000))
    Do, Io = search_oracle.topk_ip(batches[0], xb, 2000)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    ref = IndexFlatIP(128)
    ref.adopt_device(t)
    D1, I1 = ref.search_device(torch.from_numpy(batches[4]).cuda(), 80, idx_offset=7)
    assert torch.equal(I1, got[4][1]) and torch.equal(D1, got[4][0])
    # a consumer that stops after the first result: the searches still in flight are completed before their tensors go
    gen = ps.search_batches([torch.from_numpy(b).cuda() for b in batches], 80)
    D0, I0 = next(gen)
    gen.close()
    np.testing.assert_array_equal(I0.cpu().numpy(), search_oracle.topk_ip(batches[0], xb, 80)[1])
    (D2, I2), = list(ps.search_batches([torch.from_numpy(batches[1]).cuda()], 80))      # the handles are usable again
    np.testing.assert_array_equal(I2.cpu().numpy(), search_oracle.topk_ip(batches[1], xb, 80)[1])
    ps.close()
