"""Pin the NumPy search oracle: against the smallest definition and the committed digests."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import search_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n,nq,k,block", [(300, 5, 7, 64), (1000, 17, 80, 128), (50, 3, 80, 16), (4096, 8, 80, 1000)])
def test_blocked_equals_full_argsort(n, nq, k, block):
    rng = np.random.default_rng(n + k)
    xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)   # many exact ties
    xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, k, block_rows=block, query_block=4)
    D2, I2 = search_oracle.topk_ip_argsort(xq, xb, k)
    np.testing.assert_array_equal(I, I2)
    np.testing.assert_array_equal(D, D2)


def test_tie_break_is_lowest_index_and_fill_values():
    xb = np.ones((10, 128), np.float16)
    xq = np.ones((2, 128), np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 12)
    assert I[0].tolist() == list(range(10)) + [-1, -1]
    assert D[0, 10] == search_oracle.NEG_FILL and D[0, 0] == 128.0
    D, I = search_oracle.topk_ip(xq, xb[:0], 3)
    assert (I == -1).all()


def test_scores_are_float32_upcast_products():
    # eval_retrieval.py:99-100 upcasts to float32 before searching
    rng = np.random.default_rng(2)
    xb = rng.standard_normal((64, 128)).astype(np.float16)
    xq = rng.standard_normal((4, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 5)
    ref = np.sort(xq.astype(np.float64) @ xb.astype(np.float64).T, axis=1)[:, ::-1][:, :5]
    np.testing.assert_allclose(D, ref, rtol=1e-6, atol=1e-5)


def test_golden_digests():
    with open(os.path.join(GOLDEN, "search_golden.json")) as f:
        gold = json.load(f)
    rng = np.random.default_rng(0)
    xb = rng.standard_normal((4096, 128)).astype(np.float16)
    xq = rng.standard_normal((64, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 80)
    assert I[0].tolist() == gold["normal_4096x64_k80"]["I_row0"]
    assert hashlib.sha256(I.tobytes()).hexdigest() == gold["normal_4096x64_k80"]["I_sha256"]
    rng = np.random.default_rng(1)
    xb = rng.integers(-4, 5, (4096, 128)).astype(np.float16)
    xq = rng.integers(-4, 5, (64, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 80)
    assert hashlib.sha256(I.tobytes()).hexdigest() == gold["int_4096x64_k80"]["I_sha256"]
    assert hashlib.sha256(D.tobytes()).hexdigest() == gold["int_4096x64_k80"]["D_sha256"]


def test_merge_lists_matches_unsharded():
    rng = np.random.default_rng(4)
    xb = rng.integers(-3, 4, (900, 128)).astype(np.float16)
    xq = rng.integers(-3, 4, (6, 128)).astype(np.float16)
    parts = []
    for lo, hi in [(0, 300), (300, 310), (310, 900)]:
        D, I = search_oracle.topk_ip(xq, xb[lo:hi], 20)
        parts.append((D, np.where(I >= 0, I + lo, -1)))
    D, I = search_oracle.merge_lists([p[0] for p in parts], [p[1] for p in parts], 20)
    D2, I2 = search_oracle.topk_ip(xq, xb, 20)
    np.testing.assert_array_equal(I, I2)
    np.testing.assert_array_equal(D, D2)


def test_against_torch_topk_and_threaded_variant():
    """An independent exact top-k (torch.topk over the float32 score matrix) agrees with the oracle on
    tie-free data, and the thread-pooled variant bench.py times returns the oracle's result."""
    import torch
    rng = np.random.default_rng(99)
    xb = rng.standard_normal((20000, 128)).astype(np.float16)
    xq = rng.standard_normal((70, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, 80, block_rows=3000, query_block=32)
    S = torch.from_numpy(xq.astype(np.float32)) @ torch.from_numpy(xb.astype(np.float32)).T
    Dt, It = torch.topk(S, 80, dim=1, largest=True, sorted=True)
    np.testing.assert_array_equal(I, It.numpy())
    np.testing.assert_allclose(D, Dt.numpy(), rtol=1e-6, atol=1e-5)
    D2, I2 = search_oracle.topk_ip_threaded(xq, xb, 80, workers=3, query_block=16, block_rows=3000)
    np.testing.assert_array_equal(I2, I)
    np.testing.assert_array_equal(D2, D)


def test_heap_rule_for_non_finite_scores():
    rng = np.random.default_rng(11)
    xb = rng.standard_normal((300, 16)).astype(np.float16)
    xq = rng.standard_normal((5, 16)).astype(np.float16)
    D0, I0 = search_oracle.topk_ip_argsort(xq, xb, 7)
    D1, I1 = search_oracle.topk_ip_heap(xq, xb, 7)
    np.testing.assert_array_equal(I0, I1)
    np.testing.assert_array_equal(D0, D1)
    xq[:, 0] = 1.0
    xq[:, 1] = -1.0
    xb[3, 0] = np.inf        # +inf for every query
    xb[9, 0] = np.inf
    xb[4, 0] = -np.inf       # -inf
    xb[5, 0] = np.nan
    xb[6, 0] = np.inf
    xb[6, 1] = np.inf        # inf - inf = NaN
    D, I = search_oracle.topk_ip_heap(xq, xb, 300)
    for q in range(5):
        assert list(I[q, :2]) == [3, 9] and np.isposinf(D[q, :2]).all()
        assert not set(I[q].tolist()) & {4, 5, 6}
        assert (I[q, 297:] == -1).all() and (D[q, 297:] == search_oracle.NEG_FILL).all() and (I[q, :297] >= 0).all()
        assert (np.diff(D[q, 2:297]) <= 0).all()

