"""The N>1 search path on CPU: world_size-2 gloo processes, corpus row-sharded, one all-gather of
per-shard (score, id) lists, merge -> must equal the unsharded result bit for bit.

The HIP local search / merge kernels need a GPU, so the NumPy oracle is injected in their place
here (tests may use the oracle as a stand-in; the product default is the HIP path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proqa_amd.index import ShardedIndexFlatIP, shard_bounds
        rng = np.random.default_rng(42)
        xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)
        xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
        lo, hi = shard_bounds(n, world, rank)
        local = xb[lo:hi]

        def local_search(q, kk, offset):
            D, I = search_oracle.topk_ip(q.numpy(), local, kk)
            return torch.from_numpy(D), torch.from_numpy(np.where(I >= 0, I + offset, -1))

        def merge(D_all, I_all):
            D, I = search_oracle.merge_lists(list(D_all.numpy()), list(I_all.numpy()), D_all.shape[-1])
            return torch.from_numpy(D), torch.from_numpy(I)

        index = ShardedIndexFlatIP(n, local_search=local_search, merge=merge)
        assert (index.lo, index.hi, index.world_size) == (lo, hi, world)
        D, I = index.search(torch.from_numpy(xq), k)
        np.save(os.path.join(out_dir, f"D{rank}.npy"), D.numpy())
        np.save(os.path.join(out_dir, f"I{rank}.npy"), I.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,nq,k", [(1001, 9, 80), (50, 4, 80)])
def test_two_rank_sharded_search_equals_unsharded(tmp_path, n, nq, k):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, nq, k, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(42)
    xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)
    xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, k)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"I{r}.npy"), I)
        np.testing.assert_array_equal(np.load(tmp_path / f"D{r}.npy"), D)


def _query_worker(rank, world, port, n, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proqa_amd.index import QueryShardedIndexFlatIP
        rng = np.random.default_rng(43)
        xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)
        xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
        seen = []

        def local_search(q, kk):
            seen.append(q.shape[0])
            D, I = search_oracle.topk_ip(q.numpy(), xb, kk)
            return torch.from_numpy(D), torch.from_numpy(I)

        index = QueryShardedIndexFlatIP(local_search=local_search)
        D, I = index.search(torch.from_numpy(xq), k)
        per = (nq + world - 1) // world
        assert seen == ([min(per, max(nq - rank * per, 0))] if nq > rank * per else [])
        np.save(os.path.join(out_dir, f"qD{rank}.npy"), D.numpy())
        np.save(os.path.join(out_dir, f"qI{rank}.npy"), I.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,nq,k", [(1001, 9, 80), (50, 4, 80), (300, 1, 10)])
def test_two_rank_query_sharded_search_equals_unsharded(tmp_path, n, nq, k):
    """rows replicated, queries sharded (QueryShardedIndexFlatIP): 9 queries = 5 + 4 (a padded slice), 1 query = 1 + 0 (an
    empty slice); every rank ends with the whole result"""
    port = _free_port()
    mp.spawn(_query_worker, args=(2, port, n, nq, k, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(43)
    xb = rng.integers(-3, 4, (n, 128)).astype(np.float16)
    xq = rng.integers(-3, 4, (nq, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, k)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"qI{r}.npy"), I)
        np.testing.assert_array_equal(np.load(tmp_path / f"qD{r}.npy"), D)


def test_shard_bounds_cover_without_overlap():
    from proqa_amd.index import shard_bounds
    for n in (0, 1, 7, 18_000_000):
        for g in (1, 2, 4, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
