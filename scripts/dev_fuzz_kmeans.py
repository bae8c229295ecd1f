"""Randomised parity fuzz of the k-means assign/update kernels against the NumPy oracle (dev; MI355X).
usage: python scripts/dev_fuzz_kmeans.py [seconds] [seed]"""
import ctypes, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import kmeans_oracle
from proqa_amd import _lib
from proqa_amd.group_paras import KMeans

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
lib = _lib.load()
t_end = time.time() + budget
n_cases = 0
while time.time() < t_end:
    n = int(rng.choice([1, 5, 63, 64, 65, 1000, 4097, 30000]))
    k = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 1000]))
    l2 = bool(rng.integers(0, 2))
    # integer points/centroids: distances and inner products are exact -> ids and values bit-identical
    x = rng.integers(-3, 4, (n, 128)).astype(np.float16)
    cent = rng.integers(-3, 4, (k, 128)).astype(np.float32)
    if k > 3:
        cent[k - 1] = cent[1]                         # duplicate centroid: lowest index wins
    km = KMeans(128, k, spherical_metric=not l2)
    km.centroids = torch.from_numpy(cent).to(dev)
    D, I = km.assign(torch.from_numpy(x).to(dev))
    Do, Io = kmeans_oracle.assign(x, cent, l2)
    if not ((I.cpu().numpy() == Io).all() and (D.cpu().numpy() == Do).all()):
        print(f"ASSIGN MISMATCH n={n} k={k} l2={l2} seed={seed}")
        sys.exit(1)
    # update: random assignment with empty clusters, float points -> point-order fp32 sums, bit-identical
    xf = rng.standard_normal((n, 128)).astype(np.float16)
    a = rng.integers(0, max(1, k - rng.integers(0, min(k, 3))), n).astype(np.int32)
    h = ctypes.c_void_p()
    _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h)))
    tx, ta = torch.from_numpy(xf).to(dev), torch.from_numpy(a).to(dev)
    cg = torch.full((k, 128), 7.0, dtype=torch.float32, device=dev)
    cnt = torch.zeros(k, dtype=torch.int32, device=dev)
    _lib.check(lib.proqa_kmeans_update_device(h, tx.data_ptr(), n, ta.data_ptr(), cg.data_ptr(), cnt.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    lib.proqa_kmeans_free(h)
    ref = np.zeros((k, 128), np.float32); cn = np.zeros(k, np.int64)
    for i in range(n):
        ref[a[i]] += xf[i].astype(np.float32); cn[a[i]] += 1
    ref[cn > 0] /= cn[cn > 0].astype(np.float32)[:, None]
    ref[cn == 0] = 7.0
    if not ((cnt.cpu().numpy() == cn).all() and (cg.cpu().numpy() == ref).all()):
        print(f"UPDATE MISMATCH n={n} k={k} seed={seed}")
        sys.exit(1)
    n_cases += 1
print(f"kmeans fuzz ok: {n_cases} cases in {budget:.0f} s (seed {seed})")
