"""Randomised parity fuzz of the HIP search against the NumPy oracles (dev; run on the MI355X).
usage: python scripts/dev_fuzz_search.py [seconds] [seed]"""
import os
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import search_oracle
from proqa_amd.index import IndexFlatIP, merge_topk_device

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
large = len(sys.argv) > 3 and sys.argv[3] == "large"     # fewer, bigger cases (multi-tile batches, millions of rows)
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
t_end = time.time() + budget
n_cases = 0
while time.time() < t_end:
    if large:
        n = int(rng.choice([300000, 1000000, 2500000]))
        nq = int(rng.choice([5, 250, 513, 1100, 2100]))
    else:
        n = int(rng.choice([1, 7, 100, 129, 1000, 5000, 20000, 70000, 200000]))
        nq = int(rng.choice([1, 3, 31, 32, 33, 200, 256, 257, 600]))
        if n <= 5000 and rng.random() < 0.05:
            nq = 17000                                  # more than one library call per search (QUERY_BATCH)
    k = int(rng.choice([1, 2, 5, 80, 100, 640, 1024, 1025, 1500, 3000, 5000, 10000, 11500]))
    kind = rng.choice(["int", "int_wide", "int_wide", "int_narrow", "sorted", "const", "f32_int", "f32_dense_band"])
    if k > 3000 and nq > 300:
        nq = int(rng.choice([1, 40, 300]))
    shards = int(rng.choice([1, 1, 2, 3]))
    if kind == "int":
        xb = rng.integers(-4, 5, (n, 128)).astype(np.float16); xq = rng.integers(-4, 5, (nq, 128)).astype(np.float16)
    elif kind == "int_wide":                        # few rows per score level: the one-pass large-k search holds its estimate
        xb = rng.integers(-8, 9, (n, 128)).astype(np.float16); xq = rng.integers(-8, 9, (nq, 128)).astype(np.float16)
    elif kind == "int_narrow":                      # massive ties
        xb = rng.integers(0, 2, (n, 128)).astype(np.float16); xq = rng.integers(0, 2, (nq, 128)).astype(np.float16)
    elif kind == "sorted":                          # adversarial order: scores increase with the row index
        xb = rng.integers(-1, 2, (n, 128)).astype(np.float16); xb[:, 0] = np.minimum(np.arange(n) // 7, 2000)
        xq = rng.integers(0, 2, (nq, 128)).astype(np.float16); xq[:, 0] = 1
    elif kind == "const":
        xb = np.ones((n, 128), np.float16); xq = np.ones((nq, 128), np.float16)
    elif kind == "f32_int":                         # exact-float32 mode
        xb = rng.integers(-2500, 2501, (n, 128)).astype(np.float32); xq = rng.integers(-3, 4, (nq, 128)).astype(np.float32)
    else:                                           # every row inside the fp16 error band
        xb = np.ones((n, 128), np.float32); xb[:, 0] = 1 + rng.integers(0, 2048, n) * 2.0 ** -23
        xb[:, 1] = 1 + rng.integers(0, 2048, n) * 2.0 ** -23
        xq = np.zeros((nq, 128), np.float32); xq[:, 0] = rng.choice([1.0, 2.0, 0.5], nq); xq[:, 1] = 1
    exact = xb.dtype == np.float32
    if exact and n * nq > (3e8 if large else 3e7):
        nq = max(1, int((3e8 if large else 3e7) // n))
        xq = xq[:nq]
    if not large and n * nq * 8 > 2.5e9:
        continue
    oracle = search_oracle.topk_ip_exact if exact else search_oracle.topk_ip
    Do, Io = oracle(xq, xb, k)
    tq = torch.from_numpy(xq).to(dev)
    always_nominate = bool(rng.random() < 0.5)
    # leaping rounds (thresholds at a rank j < k): the planner's rank, a rank far too high (PROQA_LEAP_RANK is read at every
    # search: most leaps fall short and are re-scanned), or none
    leap = rng.choice(["auto", "auto", "rank", "off"])
    os.environ.pop("PROQA_LEAP_RANK", None)
    if leap == "rank":
        os.environ["PROQA_LEAP_RANK"] = str(int(rng.integers(1, max(2, k))))
    bounds = np.linspace(0, n, shards + 1).astype(int)
    handles = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ix = IndexFlatIP(128)
        if leap == "off":
            ix.configure_leap("off")
        if always_nominate:
            ix.configure_nomination("always")       # the int8 nomination rounds on every shard of >= 512 rows (k <= 128)
        if hi > lo:
            if rng.random() < 0.5:
                ix.add(xb[lo:hi])
            else:
                ix.add(torch.from_numpy(xb[lo:hi]).to(dev))
        handles.append((ix, int(lo)))

    def search_all(tq_, xq_, Do_, Io_, what):
        global n_cases
        parts = [ix.search_device(tq_, k, idx_offset=lo) for ix, lo in handles]
        if shards == 1:
            D, I = parts[0]
        else:
            D, I = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        D, I = D.cpu().numpy(), I.cpu().numpy()
        n_cases += 1
        if not ((I == Io_).all() and (D == Do_).all()):
            bad = np.argwhere(I != Io_)
            print(f"MISMATCH ({what}) n={n} nq={xq_.shape[0]} k={k} kind={kind} shards={shards} seed={seed} case={n_cases}: first at {bad[:3].tolist()}")
            print(I[bad[0][0]][:12], Io_[bad[0][0]][:12], D[bad[0][0]][:6], Do_[bad[0][0]][:6])
            sys.exit(1)

    search_all(tq, xq, Do, Io, "first search")
    # the same handles again: another batch size (another launch shape, merge and schedule), the state the first search left
    # behind (an int8 copy that exists, a suspended scan, a merge size the index was moved to)
    for _ in range(int(rng.choice([0, 0, 1, 2]))):
        nq2 = int(rng.choice([1, 20, 32, 100, 256, 300]))
        if exact and n * nq2 > 3e7:
            break
        xq2 = xq[rng.integers(0, xq.shape[0], nq2)] if rng.random() < 0.5 else np.ascontiguousarray(xb[rng.integers(0, n, nq2)]).astype(xq.dtype)
        Do2, Io2 = oracle(xq2, xb, k)
        search_all(torch.from_numpy(xq2).to(dev), xq2, Do2, Io2, "repeated search")
print(f"fuzz ok: {n_cases} cases in {budget:.0f} s (seed {seed})")
