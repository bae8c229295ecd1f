"""The N>1 paths with the REAL HIP kernels: two ranks share the one GPU of the test box (gloo for
the exchange, because RCCL refuses two ranks on one device).  Sharded search must equal the
unsharded search bit for bit; sharded encode must write the same .npy as a single rank."""
import json
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _search_worker(rank, world, port, n, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proqa_amd.index import ShardedIndexFlatIP
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(7)
        xb = rng.integers(-4, 5, (n, 128)).astype(np.float16)
        xq = rng.integers(-4, 5, (nq, 128)).astype(np.float16)
        index = ShardedIndexFlatIP(n)
        index.add_local(xb[index.lo:index.hi])
        D, I = index.search(torch.from_numpy(xq).to(dev), k)
        np.save(os.path.join(out_dir, f"D{rank}.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, f"I{rank}.npy"), I.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_search(gpu_device, tmp_path):
    from oracle import search_oracle
    n, nq, k = 30011, 300, 80
    mp.spawn(_search_worker, args=(2, _free_port(), n, nq, k, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(7)
    xb = rng.integers(-4, 5, (n, 128)).astype(np.float16)
    xq = rng.integers(-4, 5, (nq, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, k)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"I{r}.npy"), I)
        np.testing.assert_array_equal(np.load(tmp_path / f"D{r}.npy"), D)


def _query_shard_worker(rank, world, port, n, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proqa_amd.index import QueryShardedIndexFlatIP
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(8)
        xb = rng.integers(-4, 5, (n, 128)).astype(np.float16)
        xq = rng.integers(-4, 5, (nq, 128)).astype(np.float16)
        index = QueryShardedIndexFlatIP()
        index.add(xb)                       # every rank holds all rows
        index.prepare()
        D, I = index.search(torch.from_numpy(xq).to(dev), k)
        assert D.is_cuda and tuple(D.shape) == (nq, k)
        np.save(os.path.join(out_dir, f"qD{rank}.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, f"qI{rank}.npy"), I.cpu().numpy())
        index.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_query_sharded_search(gpu_device, tmp_path):
    """rows replicated, queries sharded (QueryShardedIndexFlatIP): 301 queries = 151 + 150, the HIP search on both ranks"""
    from oracle import search_oracle
    n, nq, k = 70001, 301, 80
    mp.spawn(_query_shard_worker, args=(2, _free_port(), n, nq, k, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(8)
    xb = rng.integers(-4, 5, (n, 128)).astype(np.float16)
    xq = rng.integers(-4, 5, (nq, 128)).astype(np.float16)
    D, I = search_oracle.topk_ip(xq, xb, k)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"qI{r}.npy"), I)
        np.testing.assert_array_equal(np.load(tmp_path / f"qD{r}.npy"), D)


def test_two_ranks_one_gpu_sharded_encode(gpu_device, tmp_path):
    """torchrun-style launch of get_embed.py (WORLD_SIZE=2): each rank encodes a contiguous row
    range and writes its slice of one pre-sized .npy; result == the single-process file."""
    model_dir = tmp_path / "small-bert"
    model_dir.mkdir()
    shutil.copy(os.path.join(GOLDEN, "vocab_small.txt"), model_dir / "vocab.txt")
    cfg = json.load(open(os.path.join(GOLDEN, "encoder_config.json")))
    cfg["model_type"] = "bert"
    (model_dir / "config.json").write_text(json.dumps(cfg))
    z = np.load(os.path.join(GOLDEN, "encoder_golden.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}
    torch.save(sd, tmp_path / "ckpt.pt")
    gold = json.load(open(os.path.join(GOLDEN, "recall_golden.json")))
    with open(tmp_path / "paras.txt", "w") as f:
        for i in range(3):
            for doc_id, text in gold["docs"]:
                f.write(json.dumps({"id": f"{doc_id}-{i}", "text": text}) + "\n")
    common = [sys.executable, os.path.join(ROOT, "get_embed.py"), "--do_predict", "--predict_batch_size", "16",
              "--bert_model_name", str(model_dir), "--fp16", "--predict_file", str(tmp_path / "paras.txt"),
              "--init_checkpoint", str(tmp_path / "ckpt.pt"), "--eval-workers", "0"]
    env = dict(os.environ, PROQA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    single = subprocess.run(common + ["--embed_save_path", str(tmp_path / "single.npy")], env=os.environ.copy(),
                            capture_output=True, text=True)
    assert single.returncode == 0, single.stderr[-2000:]
    procs = [subprocess.Popen(common + ["--embed_save_path", str(tmp_path / "sharded.npy")],
                              env=dict(env, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    a, b = np.load(tmp_path / "single.npy"), np.load(tmp_path / "sharded.npy")
    assert a.shape == (60, 128) and a.dtype == np.float16
    # batch composition differs (padding lengths), values agree to fp16 round-off
    assert np.abs(a.astype(np.float32) - b.astype(np.float32)).max() < 2e-3
