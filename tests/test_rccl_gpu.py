"""The RCCL exchange of the sharded search on real hardware.  The test box has ONE GPU and RCCL refuses two ranks
on one device, so the collective is exercised with a single rank: `ncclAllGather` / `all_gather_into_tensor` really
run (communicator init, the flat byte buffer, the strided merge), only the wire is trivial.  Two ranks with the real
kernels are covered over gloo in test_distributed_gpu.py, the rank arithmetic on CPU in test_sharded_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data(n, nq):
    rng = np.random.default_rng(11)
    return (rng.integers(-4, 5, (n, 128)).astype(np.float16), rng.integers(-4, 5, (nq, 128)).astype(np.float16))


def _nccl_worker(rank, port, n, nq, k, transport, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1)      # RCCL
    try:
        from proqa_amd.index import ShardedIndexFlatIP
        xb, xq = _data(n, nq)
        index = ShardedIndexFlatIP(n, transport=transport)
        index.add_local(xb)
        xq_dev = torch.from_numpy(xq).cuda()
        D, I = index.search(xq_dev, k, force_collective=True)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "D.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, "I.npy"), I.cpu().numpy())
        index.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["torch", "cabi"])
def test_rccl_all_gather_single_rank(gpu_device, tmp_path, transport):
    """nccl backend, world_size 1: the search goes through the RCCL all-gather and the list merge."""
    from oracle import search_oracle
    n, nq, k = 30011, 300, 80
    mp.spawn(_nccl_worker, args=(_free_port(), n, nq, k, transport, str(tmp_path)), nprocs=1, join=True)
    xb, xq = _data(n, nq)
    D, I = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(np.load(tmp_path / "I.npy"), I)
    np.testing.assert_array_equal(np.load(tmp_path / "D.npy"), D)


def test_cabi_sharded_search_without_torch_distributed(gpu_device):
    """proqa_comm_* + proqa_sharded_search_device with no process group at all (the PyTorch-free boundary): odd
    nq*k (16-byte padding of the exchanged block), a global row offset, large k through the radix merge."""
    from oracle import search_oracle
    from proqa_amd.index import ShardedIndexFlatIP
    for n, nq, k in [(5000, 7, 3), (20000, 33, 1025), (9000, 5, 9000)]:
        xb, xq = _data(n, nq)
        index = ShardedIndexFlatIP(n, transport="cabi")
        assert (index.world_size, index.lo, index.hi) == (1, 0, n)
        index.add_local(xb)
        D, I = index.search(torch.from_numpy(xq).cuda(), k)
        Do, Io = search_oracle.topk_ip(xq, xb, k)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
        # the same rows as the second half of a twice-as-large corpus: global ids are offset
        index.lo = n
        D2, I2 = index.search(torch.from_numpy(xq).cuda(), k)
        np.testing.assert_array_equal(I2.cpu().numpy(), np.where(Io >= 0, Io + n, Io))
        index.close()


def _adversarial(n, nq):
    """rows sorted by ascending score (with ties): every round's candidate lists overflow"""
    xb = np.zeros((n, 128), np.float16)
    xb[:, 0] = (np.arange(n) // 40).astype(np.float16)
    xq = np.zeros((nq, 128), np.float16)
    xq[:, 0] = 1
    return xb, xq


def test_cabi_exchange_is_repeated_after_an_overflow(gpu_device):
    """The sharded search enqueues the all-gather and the merge behind a local search whose host check is still due.
    Here that check finds overflowed rounds: the rank's status word (gathered with its block) says so, the local list
    is rewritten by the overflow-safe re-scan and the exchange runs once more -- the result is the exact one."""
    from oracle import search_oracle
    from proqa_amd.index import ShardedIndexFlatIP
    n, nq, k = 40000, 70, 80
    xb, xq = _adversarial(n, nq)
    index = ShardedIndexFlatIP(n, transport="cabi")
    index.local_index.configure(first_slab_rows=128, growth=4)
    index.add_local(xb)
    D, I = index.search(torch.from_numpy(xq).cuda(), k)
    Do, Io = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)
    assert index.local_index.last_stats()["fallback_rounds"] > 0
    # and a well-behaved search on the same handles afterwards
    rng = np.random.default_rng(3)
    xq2 = rng.integers(-4, 5, (33, 128)).astype(np.float16)
    D2, I2 = index.search(torch.from_numpy(xq2).cuda(), k)
    Do2, Io2 = search_oracle.topk_ip(xq2, xb, k)
    np.testing.assert_array_equal(I2.cpu().numpy(), Io2)
    index.close()


def _nccl_overflow_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1)
    try:
        from proqa_amd.index import ShardedIndexFlatIP
        xb, xq = _adversarial(40000, 70)
        index = ShardedIndexFlatIP(40000, transport="torch")
        index.local_index.configure(first_slab_rows=128, growth=4)
        index.add_local(xb)
        D, I = index.search(torch.from_numpy(xq).cuda(), 80, force_collective=True)
        np.save(os.path.join(out_dir, "D.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, "I.npy"), I.cpu().numpy())
        np.save(os.path.join(out_dir, "fb.npy"), np.array([index.local_index.last_stats()["fallback_rounds"]]))
        index.close()
    finally:
        dist.destroy_process_group()


def test_torch_exchange_is_repeated_after_an_overflow(gpu_device, tmp_path):
    from oracle import search_oracle
    mp.spawn(_nccl_overflow_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    xb, xq = _adversarial(40000, 70)
    Do, Io = search_oracle.topk_ip(xq, xb, 80)
    np.testing.assert_array_equal(np.load(tmp_path / "I.npy"), Io)
    np.testing.assert_array_equal(np.load(tmp_path / "D.npy"), Do)
    assert np.load(tmp_path / "fb.npy")[0] > 0


def test_a_failing_local_search_still_enters_the_collective(gpu_device):
    """A rank whose local search fails poisons its status word and goes through the all-gather all the same: the call
    returns the error (with more ranks: on every rank) instead of leaving the others waiting in the collective.  The
    handles stay usable."""
    import ctypes
    from oracle import search_oracle
    from proqa_amd import _lib
    from proqa_amd.index import ShardedIndexFlatIP
    xb, xq = _data(20000, 9)
    index = ShardedIndexFlatIP(20000, transport="cabi")
    index.add_local(xb)
    xq_dev = torch.from_numpy(xq).cuda()
    D = torch.empty((9, 80), dtype=torch.float32, device="cuda")
    I = torch.empty((9, 80), dtype=torch.int64, device="cuda")
    lib = _lib.load()
    rc = lib.proqa_sharded_search_device(index.local_index._h, index._comm, xq_dev.data_ptr(), 9, 7, 80, 0,   # dtype 7
                                         D.data_ptr(), I.data_ptr(), _lib.current_stream_ptr())
    assert rc == -1 and b"dtype" in lib.proqa_last_error()
    D2, I2 = index.search(xq_dev, 80)
    Do, Io = search_oracle.topk_ip(xq, xb, 80)
    np.testing.assert_array_equal(I2.cpu().numpy(), Io)
    np.testing.assert_array_equal(D2.cpu().numpy(), Do)
    index.close()


def _nccl_f32_worker(rank, port, transport, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1)
    try:
        from proqa_amd.index import ShardedIndexFlatIP
        rng = np.random.default_rng(4)
        xb = rng.standard_normal((30011, 128)).astype(np.float32)       # values fp16 cannot hold: exact-float32 mode
        xq = rng.standard_normal((70, 128)).astype(np.float32)
        index = ShardedIndexFlatIP(30011, transport=transport)
        index.add_local(xb)
        D, I = index.search(torch.from_numpy(xq).cuda(), 80, force_collective=True)
        np.save(os.path.join(out_dir, "D.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, "I.npy"), I.cpu().numpy())
        np.save(os.path.join(out_dir, "exact.npy"), np.array([index.local_index.exact_f32]))
        index.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["torch", "cabi"])
def test_exact_float32_search_through_the_collective(gpu_device, tmp_path, transport):
    """bench.py's float32 leg on more than one rank: float32 rows / queries (exact-float32 mode: fp16 scan, float64
    re-scoring) through the enqueued search, the RCCL all-gather and the rank merge -- bit-identical to the oracle."""
    from oracle import search_oracle
    mp.spawn(_nccl_f32_worker, args=(_free_port(), transport, str(tmp_path)), nprocs=1, join=True)
    rng = np.random.default_rng(4)
    xb = rng.standard_normal((30011, 128)).astype(np.float32)
    xq = rng.standard_normal((70, 128)).astype(np.float32)
    Do, Io = search_oracle.topk_ip_exact(xq, xb, 80)
    assert np.load(tmp_path / "exact.npy")[0]
    np.testing.assert_array_equal(np.load(tmp_path / "I.npy"), Io)
    np.testing.assert_array_equal(np.load(tmp_path / "D.npy"), Do)


def _nccl_many_queries_worker(rank, port, n, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1)
    try:
        from proqa_amd import index as index_mod
        index_mod.QUERY_BATCH = 1000                      # the slicing of the in-place exchange at a test-sized batch
        xb, xq = _data(n, nq)
        index = index_mod.ShardedIndexFlatIP(n)
        index.add_local(xb)
        xq_dev = torch.from_numpy(xq).cuda()
        D, I = index.search(xq_dev, k, force_collective=True)
        D0, I0 = index.search(xq_dev[:0], k, force_collective=True)          # no query: nobody enters the collective
        assert tuple(D0.shape) == (0, k) and tuple(I0.shape) == (0, k)
        for bad in (xq_dev[:, :64], xq_dev.cpu()):
            try:
                index.search(bad, k, force_collective=True)
            except ValueError:
                pass
            else:
                raise AssertionError("a malformed query matrix was accepted")
        np.save(os.path.join(out_dir, "D.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, "I.npy"), I.cpu().numpy())
        index.close()
    finally:
        dist.destroy_process_group()


def test_in_place_exchange_runs_in_query_slices(gpu_device, tmp_path):
    """More queries than QUERY_BATCH on the torch + nccl transport: begin / all-gather / merge / finish per slice (full
    slices and a tail), one result; empty and malformed query matrices are handled before the collective."""
    from oracle import search_oracle
    n, nq, k = 20011, 2345, 20
    mp.spawn(_nccl_many_queries_worker, args=(_free_port(), n, nq, k, str(tmp_path)), nprocs=1, join=True)
    xb, xq = _data(n, nq)
    D, I = search_oracle.topk_ip(xq, xb, k)
    np.testing.assert_array_equal(np.load(tmp_path / "I.npy"), I)
    np.testing.assert_array_equal(np.load(tmp_path / "D.npy"), D)
