"""k-means grouping of passage embeddings into training splits, on MI355X.

Drop-in for /root/reference/retrieval/group_paras.py: `clusering()` (:20-53; the reference's
spelling) trains `ncentroids` centroids with faiss.Clustering semantics and assigns every row;
`group_paras()` (:12-18) writes `split_<i>.txt` files.  The Lloyd loop follows faiss v1.6.3
Clustering.cpp (sub-sampling to k*max_points_per_centroid with rand_perm(seed 1234), initial
centroids from rand_perm(seed+1), km_update_centroids with empty-cluster splitting); the nearest-
centroid search and the centroid means run on the GPU through libproqa_hip.so
(proqa_kmeans_assign_device / proqa_kmeans_update_device).  No CPU path.

    python -m proqa_amd.group_paras --ncentroids 10000 --niter 250 --max_points_per_centroid 1000 [--spherical]
"""
import argparse
import ctypes
import os

import numpy as np
import torch

from . import _lib

EPS = np.float32(1.0 / 1024.0)


def rand_perm(n, seed):
    perm = np.empty(n, dtype=np.int32)
    _lib.check(_lib.load().proqa_rand_perm(n, seed, perm.ctypes.data))
    return perm


class _StdMT:
    """std::mt19937 (faiss::RandomGenerator) for the empty-cluster split."""

    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(int(seed) & 0xFFFFFFFF)

    def rand_float(self):
        return np.float32(int(self.bg.random_raw())) / np.float32(4294967295.0)


class KMeans:
    """faiss.Clustering-shaped trainer over fp16 points resident in HBM."""

    def __init__(self, d, k, niter=25, max_points_per_centroid=256, seed=1234, verbose=False, spherical_metric=False):
        self.d, self.k, self.niter = d, k, niter
        self.max_points_per_centroid = max_points_per_centroid
        self.seed, self.verbose = seed, verbose
        self.l2 = not spherical_metric
        self.centroids = None
        self.obj = []
        self.iter_seconds = []      # wall time of every Lloyd iteration (assign + objective + update + void split)
        self.assign_ms = []         # HIP-event time of its nearest-centroid search alone
        self._lib = _lib.load()
        _lib.require_gpu()

    def _handle(self, n):
        h = ctypes.c_void_p()
        _lib.check(self._lib.proqa_kmeans_create(self.d, int(n), self.k, ctypes.byref(h)))
        return h

    def _assign(self, h, x, centroids, hint=None):
        n = x.shape[0]
        I = torch.empty(n, dtype=torch.int32, device=x.device)
        D = torch.empty(n, dtype=torch.float32, device=x.device)
        # `hint` (the labels of the previous Lloyd iteration): same result, less bookkeeping in the nominating pass
        _lib.check(self._lib.proqa_kmeans_assign_hinted_device(h, x.data_ptr(), n, centroids.data_ptr(), 1 if self.l2 else 0,
                                                               hint.data_ptr() if hint is not None else None, I.data_ptr(),
                                                               D.data_ptr(), _lib.current_stream_ptr()))
        return D, I

    def _split_empty(self, centroids, counts, n):
        """faiss km_update_centroids' treatment of void clusters (host: k is small, splits are rare)."""
        hassign = counts.cpu().numpy().astype(np.int64)
        empty = np.nonzero(hassign == 0)[0]
        if len(empty) == 0:
            return 0
        c = centroids.cpu().numpy()
        rng = _StdMT(1234)
        sign = np.where(np.arange(self.d) % 2 == 0, np.float32(1), np.float32(-1))
        for ci in empty:
            cj = 0
            while True:
                p = np.float32(hassign[cj] - 1.0) / np.float32(n - self.k)
                if rng.rand_float() < p:
                    break
                cj = (cj + 1) % self.k
            c[ci] = c[cj]
            c[ci] *= (np.float32(1) + sign * EPS)
            c[cj] *= (np.float32(1) - sign * EPS)
            hassign[ci] = hassign[cj] // 2
            hassign[cj] -= hassign[ci]
        centroids.copy_(torch.from_numpy(c))
        return len(empty)

    def train(self, x):
        """x: CUDA fp16 [n, d].  Leaves float32 centroids [k, d] in self.centroids (on the GPU)."""
        nx = x.shape[0]
        if nx < self.k:
            raise RuntimeError(f"Number of training points ({nx}) should be at least as large as number of clusters ({self.k})")
        if nx > self.k * self.max_points_per_centroid:
            if self.verbose:
                print(f"Sampling a subset of {self.k * self.max_points_per_centroid} / {nx} for training")
            perm = rand_perm(nx, self.seed)
            nx = self.k * self.max_points_per_centroid
            x = x[torch.from_numpy(perm[:nx].astype(np.int64)).to(x.device)].contiguous()
        if nx == self.k:
            self.centroids = x.float().contiguous()
            return self
        perm = rand_perm(nx, self.seed + 1)
        centroids = x[torch.from_numpy(perm[:self.k].astype(np.int64)).to(x.device)].float().contiguous()
        counts = torch.empty(self.k, dtype=torch.int32, device=x.device)
        h = self._handle(nx)
        try:
            import time
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            I = None
            for it in range(self.niter):
                t_it = time.perf_counter()
                ev[0].record()
                D, I = self._assign(h, x, centroids, hint=I)
                ev[1].record()
                err = float(D.double().sum())
                self.obj.append(err)
                _lib.check(self._lib.proqa_kmeans_update_device(h, x.data_ptr(), nx, I.data_ptr(), centroids.data_ptr(),
                                                                counts.data_ptr(), _lib.current_stream_ptr()))
                nsplit = self._split_empty(centroids, counts, nx)
                self.assign_ms.append(ev[0].elapsed_time(ev[1]))
                self.iter_seconds.append(time.perf_counter() - t_it)
                if self.verbose:
                    print(f"  Iteration {it} objective={err:g} nsplit={nsplit}")
        finally:
            self._lib.proqa_kmeans_free(h)
        self.centroids = centroids
        return self

    def assign(self, x):
        h = self._handle(x.shape[0])
        try:
            return self._assign(h, x, self.centroids)
        finally:
            self._lib.proqa_kmeans_free(h)


def clusering(data, niter=1000, verbose=True, ncentroids=1024, max_points_per_centroid=10000000, gpu_id=0,
              spherical=False, allow_fp16_rounding=False):
    """(D [n,1] float32, I [n,1] int64) like the reference's clusering() (its spelling).

    The reference clusters `np.float32(x)` (retrieval/group_paras.py:72-73); the points live in HBM as fp16 here.
    For the embeddings get_embed.py writes under --fp16 ('<f2' files) the float32 upcast holds exactly the fp16
    values, so nothing is lost.  float32 values fp16 cannot hold are never rounded silently: they are refused
    unless allow_fp16_rounding is set (the search index has an exact-float32 mode for such data; the k-means
    does not)."""
    device = torch.device("cuda", gpu_id)
    data = np.ascontiguousarray(data)
    if data.dtype not in (np.float16, np.float32):
        raise TypeError(f"embeddings must be float16 or float32, got {data.dtype}")
    x = torch.empty(data.shape, dtype=torch.float16, device=device)
    step = 1 << 20
    inexact = 0
    for r0 in range(0, data.shape[0], step):
        piece = torch.from_numpy(data[r0:r0 + step]).to(device)
        x[r0:r0 + step] = piece.to(torch.float16)
        if data.dtype == np.float32 and not allow_fp16_rounding:
            inexact += int((x[r0:r0 + step].float() != piece).sum())     # NaN counts as inexact: refused as well
    if inexact:
        raise ValueError(f"{inexact} float32 values are not representable in fp16; clustering them would round the "
                         "points silently.  Either pass allow_fp16_rounding=True (group_paras.py --allow-fp16-rounding) to "
                         "accept the rounding, or cluster an fp16 index: get_embed.py --fp16 (or --embed_dtype float16) "
                         "writes '<f2' embeddings, whose float32 upcast is exact.")
    km = KMeans(x.shape[1], ncentroids, niter=niter, max_points_per_centroid=max_points_per_centroid, verbose=verbose,
                spherical_metric=spherical)
    with torch.cuda.device(device):
        km.train(x)
        D, I = km.assign(x)
    return D.cpu().numpy()[:, None], I.cpu().numpy().astype(np.int64)[:, None]


def write_file(file_name, samples):
    with open(file_name, "w") as f_out:
        for line in samples:
            f_out.write(line)


def group_paras(I, ncentroids, split_path, train_file="../data/retrieve_train.txt"):
    samples = [[] for _ in range(ncentroids)]
    with open(train_file) as f_in:
        for i, line in enumerate(f_in):
            samples[I[i][0]].append(line)
    for i, group in enumerate(samples):
        write_file(split_path + "split_" + str(i) + ".txt", group)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--ncentroids", type=int, default=10000)
    parser.add_argument("--niter", type=int, default=250)
    parser.add_argument("--max_points_per_centroid", type=int, default=1000)
    parser.add_argument("--indexpath", type=str, default=None)
    parser.add_argument("--spherical", action="store_true")
    parser.add_argument("--allow-fp16-rounding", action="store_true",
                        help="cluster float32 embeddings that fp16 cannot hold after rounding them to fp16")
    # the reference hard-codes these three paths (:61-62, :14)
    parser.add_argument("--train_para_embed_path", type=str, default="encodings/train_para_embed.npy")
    parser.add_argument("--split_save_path", type=str, default="../data/data_splits/")
    parser.add_argument("--train_file", type=str, default="../data/retrieve_train.txt")
    args = parser.parse_args(argv)

    split_save_path = args.split_save_path
    if os.path.exists(split_save_path) and os.listdir(split_save_path):
        print(f"output directory {split_save_path} already exists and is not empty.")
    if not os.path.exists(split_save_path):
        os.makedirs(split_save_path, exist_ok=True)

    from . import npy
    x = npy.load(args.train_para_embed_path)
    D, I = clusering(x, niter=args.niter, ncentroids=args.ncentroids,
                     max_points_per_centroid=args.max_points_per_centroid, spherical=args.spherical,
                     allow_fp16_rounding=args.allow_fp16_rounding)
    group_paras(I, args.ncentroids, split_path=split_save_path, train_file=args.train_file)
    return D, I


if __name__ == "__main__":
    main()
