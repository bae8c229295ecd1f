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
    # float data through the two-pass assignment (hi-only nomination + full precision for the undecided): labels may differ from
    # the float64 oracle only where its two best centroids are closer than fp32 round-off; then a Lloyd step (update, hinted
    # assignment in the update's sorted order) against an unhinted assignment on a fresh handle: identical labels
    if n >= 64 and k >= 2:
        centers = rng.standard_normal((k, 128)).astype(np.float32) * np.float32(rng.choice([0.05, 1.0, 8.0]))
        if k > 4:
            centers[1::2] = centers[0::2][: len(centers[1::2])] * np.float32(1 + 1e-4)      # twins a few fp16 ulps apart
        xs = (centers[rng.integers(0, k, n)] + np.float32(rng.choice([0.01, 0.3])) * rng.standard_normal((n, 128))).astype(np.float16)
        tx = torch.from_numpy(xs).to(dev)
        tc = torch.from_numpy(centers).to(dev)
        def assign(handle, cen, hint=None):
            lab = torch.empty(n, dtype=torch.int32, device=dev); dist = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(lib.proqa_kmeans_assign_hinted_device(handle, tx.data_ptr(), n, cen.data_ptr(), 1 if l2 else 0,
                                                             hint.data_ptr() if hint is not None else None, lab.data_ptr(),
                                                             dist.data_ptr(), torch.cuda.current_stream().cuda_stream))
            return lab, dist
        h1, h2 = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h1)))
        _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h2)))
        lab, dist = assign(h1, tc)
        x64, c64 = xs.astype(np.float64), centers.astype(np.float64)
        sc = x64 @ c64.T - (0.5 * (c64 * c64).sum(1) if l2 else 0.0)
        got = lab.cpu().numpy()
        best = sc.max(1)
        lost = best - sc[np.arange(n), got]
        if (lost > 1e-3 * (1 + np.abs(best))).any():
            print(f"FLOAT ASSIGN MISMATCH n={n} k={k} l2={l2} seed={seed} worst {lost.max()}")
            sys.exit(1)
        c1 = tc.clone(); cnt = torch.zeros(k, dtype=torch.int32, device=dev)
        _lib.check(lib.proqa_kmeans_update_device(h1, tx.data_ptr(), n, lab.data_ptr(), c1.data_ptr(), cnt.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream))
        a1, _ = assign(h1, c1, lab)
        a2, _ = assign(h2, c1)
        if not torch.equal(a1, a2):
            print(f"HINTED ASSIGN MISMATCH n={n} k={k} l2={l2} seed={seed}: {(a1 != a2).sum().item()} labels differ")
            sys.exit(1)
        lib.proqa_kmeans_free(h1); lib.proqa_kmeans_free(h2)
    n_cases += 1
print(f"kmeans fuzz ok: {n_cases} cases in {budget:.0f} s (seed {seed})")
