"""k-means (group_paras.py row of SURVEY section 8f) on the GPU against the NumPy restatement of
faiss.Clustering (oracle/kmeans_oracle.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import kmeans_oracle

pytestmark = pytest.mark.gpu


def blobs(rng, n, k, spread=0.15):
    centers = rng.standard_normal((k, 128)).astype(np.float32)
    lab = rng.integers(0, k, n)
    return (centers[lab] + spread * rng.standard_normal((n, 128))).astype(np.float16), lab


@pytest.mark.parametrize("l2", [True, False])
@pytest.mark.parametrize("n,k", [(3000, 37), (700, 64), (20000, 300)])
def test_assign_matches_oracle(gpu_device, n, k, l2):
    from proqa_amd.group_paras import KMeans
    rng = np.random.default_rng(n + k)
    x, _ = blobs(rng, n, max(k // 2, 2))
    cent = rng.standard_normal((k, 128)).astype(np.float32) * (1 + rng.random((k, 1)).astype(np.float32))
    km = KMeans(128, k, spherical_metric=not l2)
    km.centroids = torch.from_numpy(cent).to(gpu_device)
    D, I = km.assign(torch.from_numpy(x).to(gpu_device))
    Do, Io = kmeans_oracle.assign(x, cent, l2)
    I = I.cpu().numpy()
    # fp32-grade dot products: assignments may differ only where two centroids tie to round-off
    mism = np.nonzero(I != Io)[0]
    assert len(mism) <= max(1, n // 2000)
    np.testing.assert_allclose(D.cpu().numpy(), Do, rtol=2e-4, atol=2e-3)


def test_assign_integer_points_exact_and_tie_break(gpu_device):
    from proqa_amd.group_paras import KMeans
    rng = np.random.default_rng(1)
    x = rng.integers(-3, 4, (2000, 128)).astype(np.float16)
    cent = rng.integers(-3, 4, (100, 128)).astype(np.float32)
    cent[57] = cent[3]                      # duplicate centroid: the lower index must win
    for l2 in (True, False):
        km = KMeans(128, 100, spherical_metric=not l2)
        km.centroids = torch.from_numpy(cent).to(gpu_device)
        D, I = km.assign(torch.from_numpy(x).to(gpu_device))
        Do, Io = kmeans_oracle.assign(x, cent, l2)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy(), Do)
        assert not (I.cpu().numpy() == 57).any()


def test_update_is_bit_identical_to_point_order_sums(gpu_device):
    import ctypes
    from proqa_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2)
    n, k = 5000, 41
    x = rng.standard_normal((n, 128)).astype(np.float16)
    a = rng.integers(0, k - 1, n).astype(np.int32)       # cluster k-1 stays empty
    h = ctypes.c_void_p()
    _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h)))
    tx, ta = torch.from_numpy(x).to(gpu_device), torch.from_numpy(a).to(gpu_device)
    cent = torch.full((k, 128), 7.0, dtype=torch.float32, device=gpu_device)
    cnt = torch.zeros(k, dtype=torch.int32, device=gpu_device)
    _lib.check(lib.proqa_kmeans_update_device(h, tx.data_ptr(), n, ta.data_ptr(), cent.data_ptr(), cnt.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    lib.proqa_kmeans_free(h)
    ref = np.zeros((k, 128), np.float32)
    cn = np.zeros(k, np.int64)
    for i in range(n):
        ref[a[i]] += x[i].astype(np.float32)
        cn[a[i]] += 1
    ref[cn > 0] /= cn[cn > 0].astype(np.float32)[:, None]
    got = cent.cpu().numpy()
    np.testing.assert_array_equal(cnt.cpu().numpy(), cn)
    np.testing.assert_array_equal(got[:k - 1], ref[:k - 1])
    assert (got[k - 1] == 7.0).all()                       # empty cluster untouched


def test_rand_perm_is_faiss_rand_perm():
    from proqa_amd.group_paras import rand_perm
    for n, seed in [(1, 5), (10, 1234), (1000, 1235)]:
        np.testing.assert_array_equal(rand_perm(n, seed), kmeans_oracle.rand_perm(n, seed))


def test_clusering_matches_oracle_trajectory(gpu_device):
    """Whole group_paras.clusering on separated blobs (incl. sub-sampling): same partition as the
    oracle, objective per iteration within 1e-4 relative."""
    from proqa_amd.group_paras import KMeans, clusering
    rng = np.random.default_rng(3)
    x, lab = blobs(rng, 4000, 16, spread=0.05)
    km = KMeans(128, 16, niter=6, max_points_per_centroid=100)       # 4000 > 16*100: sub-samples
    tx = torch.from_numpy(x).to(gpu_device)
    km.train(tx)
    cent_o, obj_o = kmeans_oracle.train(x, 16, 6, 100, True)
    np.testing.assert_allclose(km.obj, obj_o, rtol=1e-4)
    np.testing.assert_allclose(km.centroids.cpu().numpy(), cent_o, rtol=1e-4, atol=1e-4)
    D, I = clusering(x, niter=6, verbose=False, ncentroids=16, max_points_per_centroid=100)
    Do, Io, _ = kmeans_oracle.clustering(x, 6, 16, 100)
    assert D.shape == (4000, 1) and I.shape == (4000, 1) and I.dtype == np.int64
    assert (I == Io).mean() > 0.999


def test_clusering_never_rounds_float32_silently(gpu_device):
    """The reference clusters np.float32(x) (group_paras.py:72-73).  float32 embeddings whose values are fp16
    numbers (what '<f2' files upcast to) cluster exactly like the fp16 array; values fp16 cannot hold are refused
    unless the caller accepts the rounding."""
    from proqa_amd.group_paras import clusering
    rng = np.random.default_rng(9)
    x16, _ = blobs(rng, 2000, 8, spread=0.05)
    x16 = x16.astype(np.float16)
    D16, I16 = clusering(x16, niter=4, verbose=False, ncentroids=8, max_points_per_centroid=1000)
    D32, I32 = clusering(x16.astype(np.float32), niter=4, verbose=False, ncentroids=8, max_points_per_centroid=1000)
    np.testing.assert_array_equal(I16, I32)
    np.testing.assert_array_equal(D16, D32)
    bad = (rng.integers(-2500, 2501, (2000, 128))).astype(np.float32)     # odd integers above 2048 are not fp16 numbers
    with pytest.raises(ValueError, match="not representable in fp16"):
        clusering(bad, niter=2, verbose=False, ncentroids=8, max_points_per_centroid=1000)
    D, I = clusering(bad, niter=2, verbose=False, ncentroids=8, max_points_per_centroid=1000, allow_fp16_rounding=True)
    Dr, Ir = clusering(bad.astype(np.float16), niter=2, verbose=False, ncentroids=8, max_points_per_centroid=1000)
    np.testing.assert_array_equal(I, Ir)


def test_empty_cluster_split_like_faiss(gpu_device):
    """Update step with void clusters: means + faiss' re-seeding rule (size-weighted pick driven by
    RandomGenerator(1234), +-1/1024 perturbation) must equal the oracle's km_update_centroids."""
    import ctypes
    from proqa_amd import _lib
    from proqa_amd.group_paras import KMeans
    lib = _lib.load()
    rng = np.random.default_rng(4)
    n, k = 3000, 23
    x = rng.standard_normal((n, 128)).astype(np.float16)
    a = rng.integers(0, k, n).astype(np.int32)
    a[np.isin(a, [2, 11, 22])] = 5                      # clusters 2, 11, 22 end up empty
    km = KMeans(128, k)
    h = ctypes.c_void_p()
    _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h)))
    tx, ta = torch.from_numpy(x).to(gpu_device), torch.from_numpy(a).to(gpu_device)
    cent = torch.zeros((k, 128), dtype=torch.float32, device=gpu_device)
    cnt = torch.zeros(k, dtype=torch.int32, device=gpu_device)
    _lib.check(lib.proqa_kmeans_update_device(h, tx.data_ptr(), n, ta.data_ptr(), cent.data_ptr(), cnt.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    lib.proqa_kmeans_free(h)
    nsplit = km._split_empty(cent, cnt, n)
    ref, hassign, nsplit_o = kmeans_oracle.update_centroids(x, None, a, k)
    assert nsplit == nsplit_o == 3
    np.testing.assert_array_equal(cent.cpu().numpy(), ref)


def test_group_paras_cli_writes_splits(gpu_device, tmp_path):
    from proqa_amd import group_paras
    rng = np.random.default_rng(5)
    x, _ = blobs(rng, 300, 4, spread=0.05)
    np.save(tmp_path / "train_para_embed.npy", x)
    with open(tmp_path / "retrieve_train.txt", "w") as f:
        for i in range(300):
            f.write(f'{{"q": {i}}}\n')
    out = str(tmp_path / "splits") + os.sep
    D, I = group_paras.main(["--ncentroids", "4", "--niter", "5", "--max_points_per_centroid", "1000",
                             "--train_para_embed_path", str(tmp_path / "train_para_embed.npy"),
                             "--split_save_path", out, "--train_file", str(tmp_path / "retrieve_train.txt")])
    files = sorted(os.listdir(out))
    assert files == [f"split_{i}.txt" for i in range(4)]
    lines = sum(len(open(os.path.join(out, f)).readlines()) for f in files)
    assert lines == 300
    for i in range(4):
        want = [f'{{"q": {j}}}\n' for j in range(300) if I[j][0] == i]
        assert open(os.path.join(out, f"split_{i}.txt")).readlines() == want


_TWO_PASS_CHILD = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from proqa_amd.group_paras import KMeans
dev = torch.device("cuda:0")
rng = np.random.default_rng(9)
out = []
for name, n, k, l2 in (("clustered", 50000, 700, True), ("clustered-ip", 30000, 257, False), ("near-ties", 20000, 128, True),
                      ("ragged", 1537, 65, True)):
    centers = rng.standard_normal((k, 128)).astype(np.float32)
    if name == "near-ties":                       # every centroid has a twin a few fp16 ulps away, and an exact duplicate
        centers[1::2] = centers[0::2] * (1 + 2e-4 * rng.standard_normal((k // 2, 1)).astype(np.float32))
        centers[5] = centers[2]
    x = (centers[rng.integers(0, k, n)] + 0.3 * rng.standard_normal((n, 128))).astype(np.float16)
    km = KMeans(128, k, spherical_metric=not l2)
    km.centroids = torch.from_numpy(centers).to(dev)
    D, I = km.assign(torch.from_numpy(x).to(dev))
    out.append(name + ":" + hashlib.sha256(I.cpu().numpy().tobytes()).hexdigest()[:16] + ":%.9e" % float(D.double().sum()))
print("RESULT " + " ".join(out))
"""


def test_two_pass_assignment_equals_the_full_precision_pass(gpu_device, tmp_path):
    """The default assignment (hi-only nomination of every point + the full-precision kernel over the undecided ones) gives
    the labels of the full-precision kernel over all points (PROQA_KMEANS_TWO_PASS=0): clustered data, inner-product metric,
    centroids with twins a few ulps apart and exact duplicates, ragged sizes; the summed distances agree to 1e-6."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", _TWO_PASS_CHILD, root], env=dict(os.environ, PROQA_KMEANS_TWO_PASS=mode),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][0]
        res[mode] = [t.split(":") for t in line.split()[1:]]
    for a, b in zip(res["1"], res["0"]):
        assert a[0] == b[0] and a[1] == b[1], (a, b)                      # identical labels
        assert abs(float(a[2]) - float(b[2])) <= 1e-6 * abs(float(b[2])), (a, b)


def test_hinted_assignment_in_sorted_order_gives_the_same_labels(gpu_device):
    """A Lloyd loop's second assignment -- hinted with the previous labels and walking the points in the sorted order the
    update in between left behind -- returns exactly what an unhinted assignment on a fresh handle returns, for right,
    wrong and out-of-range hints."""
    import ctypes
    from proqa_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(21)
    n, k = 40000, 300
    x_np, _ = blobs(rng, n, k, spread=0.4)
    x = torch.from_numpy(x_np).to(gpu_device)
    cent = torch.from_numpy(rng.standard_normal((k, 128)).astype(np.float32)).to(gpu_device)

    def handle():
        h = ctypes.c_void_p()
        _lib.check(lib.proqa_kmeans_create(128, n, k, ctypes.byref(h)))
        return h

    def assign(h, c, hint=None):
        lab = torch.empty(n, dtype=torch.int32, device=gpu_device)
        dist = torch.empty(n, dtype=torch.float32, device=gpu_device)
        _lib.check(lib.proqa_kmeans_assign_hinted_device(h, x.data_ptr(), n, c.data_ptr(), 1, hint.data_ptr() if hint is not None else None,
                                                         lab.data_ptr(), dist.data_ptr(), _lib.current_stream_ptr()))
        return lab, dist

    h = handle()
    lab0, _ = assign(h, cent)
    cent1 = cent.clone()
    counts = torch.empty(k, dtype=torch.int32, device=gpu_device)
    _lib.check(lib.proqa_kmeans_update_device(h, x.data_ptr(), n, lab0.data_ptr(), cent1.data_ptr(), counts.data_ptr(),
                                              _lib.current_stream_ptr()))
    fresh = handle()
    want, want_d = assign(fresh, cent1)
    hints = {"previous labels": lab0, "random": torch.randint(0, k, (n,), device=gpu_device, dtype=torch.int32),
             "out of range": torch.full((n,), k + 7, device=gpu_device, dtype=torch.int32),
             "negative": torch.full((n,), -1, device=gpu_device, dtype=torch.int32)}
    for name, hint in hints.items():
        got, got_d = assign(h, cent1, hint)           # (the handle still holds the update's sorted order)
        assert torch.equal(got, want), name
        torch.testing.assert_close(got_d, want_d, rtol=1e-5, atol=1e-4)
    got, _ = assign(h, cent1, lab0.clone())            # in place: the hint buffer is the output buffer
    lab_io = lab0.clone()
    dist = torch.empty(n, dtype=torch.float32, device=gpu_device)
    _lib.check(lib.proqa_kmeans_assign_hinted_device(h, x.data_ptr(), n, cent1.data_ptr(), 1, lab_io.data_ptr(), lab_io.data_ptr(),
                                                     dist.data_ptr(), _lib.current_stream_ptr()))
    assert torch.equal(lab_io, want)
    for hh in (h, fresh):
        lib.proqa_kmeans_free(hh)
