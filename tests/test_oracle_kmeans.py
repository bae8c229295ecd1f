"""Pin the NumPy k-means oracle (restated faiss.Clustering) against brute-force definitions, and
the C-ABI rand_perm against it (no GPU needed)."""
import numpy as np

from oracle import kmeans_oracle as ko


def test_rand_perm_is_a_seeded_permutation_and_matches_c_abi():
    from proqa_amd.group_paras import rand_perm
    for n, seed in [(0, 1), (1, 1), (7, 1234), (500, 1235), (500, 99)]:
        p = ko.rand_perm(n, seed)
        assert sorted(p.tolist()) == list(range(n))
        np.testing.assert_array_equal(rand_perm(n, seed), p)
    assert not np.array_equal(ko.rand_perm(500, 1234), ko.rand_perm(500, 1235))
    # first draws of std::mt19937(1234): 822569775, 2137449171, 2671936806 (C++11 reference values)
    rng = ko.MT19937(1234)
    assert [rng.raw() for _ in range(3)] == [822569775, 2137449171, 2671936806]


def test_assign_is_exact_nearest_with_lowest_index_ties():
    rng = np.random.default_rng(0)
    x = rng.integers(-2, 3, (60, 128)).astype(np.float16)
    c = rng.integers(-2, 3, (9, 128)).astype(np.float32)
    c[6] = c[1]
    D, I = ko.assign(x, c, True)
    for i in range(60):
        d = ((x[i].astype(np.float64) - c.astype(np.float64)) ** 2).sum(1)
        assert I[i] == int(np.argmin(d)) and I[i] != 6
        assert D[i] == np.float32(d.min())
    D, I = ko.assign(x, c, False)
    for i in range(60):
        s = c.astype(np.float64) @ x[i].astype(np.float64)
        assert I[i] == int(np.argmax(s)) and D[i] == np.float32(s.max())


def test_update_means_and_void_cluster_rule():
    rng = np.random.default_rng(1)
    n, k = 400, 7
    x = rng.standard_normal((n, 128)).astype(np.float16)
    a = rng.integers(0, k - 1, n)                       # cluster 6 empty
    c, h, nsplit = ko.update_centroids(x, None, a, k)
    assert nsplit == 1 and h.sum() == n and (h > 0).all()
    for ci in range(k - 1):
        m = x[a == ci].astype(np.float64).mean(0)
        # the split partner was perturbed by (1 +- 1/1024); everyone else is the plain mean
        assert np.allclose(c[ci], m, rtol=2e-3, atol=1e-4)
    ratios = c[6] / c[np.argmin(np.abs(c[:6] - c[6]).sum(1))]
    assert np.allclose(np.abs(ratios - 1), 2.0 / 1024, atol=1e-4)


def test_train_decreases_objective_and_subsamples():
    rng = np.random.default_rng(2)
    centers = rng.standard_normal((5, 128))
    x = (centers[rng.integers(0, 5, 600)] + 0.05 * rng.standard_normal((600, 128))).astype(np.float16)
    c, obj = ko.train(x, 5, 5, 1000, True)
    assert all(b <= a * (1 + 1e-6) for a, b in zip(obj, obj[1:]))
    c2, obj2 = ko.train(x, 5, 3, 20, True)              # 600 > 5*20: trains on 100 sampled rows
    assert c2.shape == (5, 128) and len(obj2) == 3
    D, I, _ = ko.clustering(x, 3, 5, 1000)
    assert D.shape == (600, 1) and I.shape == (600, 1)


def test_lloyd_iterations_match_scikit_learn():
    """Independent implementation of the same Lloyd iteration (no void clusters, same initial
    centroids): scikit-learn's KMeans lands on the centroids the restated faiss loop produces."""
    import warnings
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(3)
    k, n, niter = 6, 900, 4
    centers = 3 * rng.standard_normal((k, 128))
    x = (centers[rng.integers(0, k, n)] + 0.7 * rng.standard_normal((n, 128))).astype(np.float16)
    init = (centers + 0.5 * rng.standard_normal((k, 128))).astype(np.float32)   # one start per true cluster
    c = init.copy()
    for _ in range(niter):
        _, a = ko.assign(x, c, True)
        c, h, nsplit = ko.update_centroids(x, c, a, k)
        assert nsplit == 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=k, init=init.astype(np.float64), n_init=1, max_iter=niter, tol=0.0,
                    algorithm="lloyd").fit(x.astype(np.float64))
    np.testing.assert_allclose(c, km.cluster_centers_, rtol=1e-4, atol=1e-4)
    _, a = ko.assign(x, c, True)
    np.testing.assert_array_equal(a, km.predict(x.astype(np.float64)))
