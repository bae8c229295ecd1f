"""NumPy restatement of the k-means behind group_paras.py — TEST INFRASTRUCTURE ONLY.

/root/reference/retrieval/group_paras.py:20-53 builds `faiss.Clustering(d, ncentroids)` (niter,
max_points_per_centroid, verbose set; everything else default), trains it with an IndexFlatL2
(IndexFlatIP when --spherical; NB `clus.spherical` itself is never set, so centroids are not
re-normalised), then assigns every row with `index.search(data, 1)`.

The arithmetic lives in faiss-cpu==1.6.3 (requirements.txt:2), not vendored and not installable
here, so its published algorithm (faiss/Clustering.cpp, faiss/utils/random.cpp at v1.6.3) is
restated: rand_perm (Fisher-Yates on std::mt19937), sub-sampling to k*max_points_per_centroid with
seed 1234, initial centroids = first k rows of rand_perm(seed+1), Lloyd iterations with
km_update_centroids (fp32 sums in point order, empty clusters split from a size-weighted random
cluster with the +-1/1024 perturbation, RandomGenerator(1234)).  Parity status: UNPINNED by the
reference (no tests, faiss cannot run here); ties in the nearest-centroid search (unspecified by
FAISS) go to the lowest centroid index.
"""
import numpy as np

EPS = np.float32(1.0 / 1024.0)


class MT19937:
    """std::mt19937 stream (what faiss::RandomGenerator wraps)."""

    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(int(seed) & 0xFFFFFFFF)

    def raw(self):
        return int(self.bg.random_raw())

    def rand_int(self, mx):
        return self.raw() % mx

    def rand_float(self):
        return np.float32(self.raw()) / np.float32(4294967295.0)


def rand_perm(n, seed):
    perm = np.arange(n, dtype=np.int64)
    rng = MT19937(seed)
    for i in range(n - 1):
        i2 = i + rng.rand_int(n - i)
        perm[i], perm[i2] = perm[i2], perm[i]
    return perm


def assign(x, centroids, l2=True):
    """index.search(x, 1): (D [n], I [n]); squared L2 (clipped at 0) or inner product."""
    x32 = np.asarray(x, np.float32)
    c32 = np.asarray(centroids, np.float32)
    ip = x32.astype(np.float64) @ c32.astype(np.float64).T
    if l2:
        d = (x32.astype(np.float64) ** 2).sum(1)[:, None] + (c32.astype(np.float64) ** 2).sum(1)[None] - 2 * ip
        I = np.argmin(d, axis=1)           # first minimum = lowest centroid index
        D = np.maximum(d[np.arange(len(I)), I], 0)
    else:
        I = np.argmax(ip, axis=1)
        D = ip[np.arange(len(I)), I]
    return D.astype(np.float32), I.astype(np.int64)


def update_centroids(x, centroids, assignment, k):
    """km_update_centroids: returns (centroids, hassign, nsplit)."""
    x32 = np.asarray(x, np.float32)
    n, d = x32.shape
    c = np.zeros((k, d), np.float32)
    hassign = np.zeros(k, np.int64)
    for i in range(n):                      # fp32 accumulation in point order
        ci = assignment[i]
        hassign[ci] += 1
        c[ci] += x32[i]
    nz = hassign > 0
    c[nz] /= hassign[nz].astype(np.float32)[:, None]
    nsplit = 0
    rng = MT19937(1234)
    for ci in range(k):
        if hassign[ci] == 0:
            cj = 0
            while True:
                p = np.float32(hassign[cj] - 1.0) / np.float32(n - k)
                if rng.rand_float() < p:
                    break
                cj = (cj + 1) % k
            c[ci] = c[cj]
            sign = np.where(np.arange(d) % 2 == 0, np.float32(1), np.float32(-1))
            c[ci] *= (np.float32(1) + sign * EPS)
            c[cj] *= (np.float32(1) - sign * EPS)
            hassign[ci] = hassign[cj] // 2
            hassign[cj] -= hassign[ci]
            nsplit += 1
    return c, hassign, nsplit


def train(x, k, niter, max_points_per_centroid=256, l2=True, seed=1234):
    """Clustering::train for nredo=1: returns (centroids [k,d] f32, objective per iteration)."""
    x32 = np.asarray(x, np.float32)
    nx = x32.shape[0]
    assert nx >= k
    if nx > k * max_points_per_centroid:
        perm = rand_perm(nx, seed)
        nx = k * max_points_per_centroid
        x32 = x32[perm[:nx]]
    if nx == k:
        return x32.copy(), []
    perm = rand_perm(nx, seed + 1)
    centroids = x32[perm[:k]].copy()
    obj = []
    for _ in range(niter):
        D, I = assign(x32, centroids, l2)
        obj.append(float(D.astype(np.float64).sum()))
        centroids, _, _ = update_centroids(x32, centroids, I, k)
    return centroids, obj


def clustering(data, niter, ncentroids, max_points_per_centroid, spherical=False):
    """group_paras.clusering: (D [n,1], I [n,1]) of the final index.search(data, 1)."""
    l2 = not spherical
    centroids, _ = train(data, ncentroids, niter, max_points_per_centroid, l2)
    D, I = assign(data, centroids, l2)
    return D[:, None], I[:, None], centroids
