"""Parity cases pinned by tests/golden/round2.npz (outputs of the reference itself): range_search with
max_results, pairwise_cosine with a caller's eps, the hamming / angular functions and an angular IVF
index, the batch functions of distance/simd.py, the dtype of a zero cosine query.

Bars: bit-exact where the reference's result is reproducible (Manhattan, direct-form Euclidean, Hamming
counts, kept row sets and their order); 1e-4 relative where it goes through BLAS."""

import numpy as np
import pytest

from oracle import ref_numpy as R
from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def ids_of(n):
    return [f"v{i}" for i in range(n)]


def rows_of(results):
    return np.array([int(r.id[1:]) for r in results], dtype=np.int64)


@pytest.mark.parametrize("metric", ["euclidean", "manhattan", "cosine", "dot"])
def test_range_search_reference_semantics(metric):
    """The first max_results rows in insertion order inside the radius, then sorted (index/flat.py:545-570)."""
    from vectordb_amd.index import FlatIndex
    g = load_golden("round2.npz")
    x, q = g["x"], g["q"]
    idx = FlatIndex(dimension=24, metric=metric)
    idx.add_batch(ids_of(len(x)), x, [{"cat": int(i % 4)} for i in range(len(x))])
    radius = float(g[f"range_{metric}_radius"])
    qq = q[0] if metric != "dot" else q[1]
    for tag, kw in (("all", {}), ("max5", {"max_results": 5}), ("max40", {"max_results": 40}),
                    ("filter_max10", {"max_results": 10, "filter_fn": lambda _id, m: m["cat"] != 2})):
        res = idx.range_search(qq, radius=radius, **kw)
        want_rows, want_dist = g[f"range_{metric}_{tag}_rows"], g[f"range_{metric}_{tag}_dist"]
        got_rows, got_dist = rows_of(res), np.array([r.distance for r in res])
        assert np.all(np.diff(got_dist) >= 0)
        if metric == "manhattan":
            np.testing.assert_array_equal(got_rows, want_rows)
            np.testing.assert_array_equal(got_dist, want_dist)
        else:
            # the kept set is decided by row order and the radius; rows whose reference distance is within rounding
            # of the radius may fall on either side
            full = R.query_distances(qq, x, metric).astype(np.float64)
            for r in set(got_rows.tolist()) ^ set(want_rows.tolist()):
                assert tag == "all" and abs(full[r] - radius) <= RTOL * max(1.0, abs(radius)), (tag, r, full[r])
            if set(got_rows.tolist()) == set(want_rows.tolist()):
                np.testing.assert_allclose(np.sort(got_dist), np.sort(want_dist), rtol=RTOL, atol=2e-5)
        for r in res:
            assert r.score == pytest.approx(1.0 / (1.0 + r.distance))
    # exact ties keep insertion order: rows 7, 40, 1200 are the same vector
    if metric in ("euclidean", "manhattan", "cosine"):
        res = idx.range_search(q[0], radius=radius)
        assert rows_of(res)[:3].tolist() == [7, 40, 1200]


def test_range_search_large_result_set(nat):
    """More results than one sort chunk: 200k rows inside the radius, every one returned once, in (distance, row) order."""
    rng = np.random.RandomState(3)
    x = rng.randn(300_000, 16).astype(np.float32)
    q = rng.randn(16).astype(np.float32)
    store = nat.RowStore(16)
    store.append(x)
    full = R.query_distances(q, x, "manhattan")
    radius = float(np.sort(full)[200_000])
    dist, rows = store.range_search(q, radius, nat.MANHATTAN)
    ref_rows, ref_dist = R.flat_range_search(q, x, radius, "manhattan")
    np.testing.assert_array_equal(rows, ref_rows)
    np.testing.assert_array_equal(dist, ref_dist)
    # with a cap: the first 70 000 in row order, sorted
    dist, rows = store.range_search(q, radius, nat.MANHATTAN, cap=70_000)
    ref_rows, ref_dist = R.flat_range_search(q, x, radius, "manhattan", max_results=70_000)
    np.testing.assert_array_equal(rows, ref_rows)
    np.testing.assert_array_equal(dist, ref_dist)
    # nothing inside the radius
    dist, rows = store.range_search(q, -1.0, nat.MANHATTAN)
    assert len(rows) == 0
    store.close()


def test_pairwise_cosine_with_eps():
    from vectordb_amd.distance import pairwise_cosine
    g = load_golden("round2.npz")
    for tag, eps in (("default", 1e-8), ("1e-3", 1e-3), ("1e-2", 1e-2)):
        got = pairwise_cosine(g["eps_x"], g["eps_y"], eps=eps)
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, g[f"pairwise_cosine_eps_{tag}"], rtol=RTOL, atol=2e-6)
    # fewer than four rows on the left: the streaming kernel serves the matrix
    got = pairwise_cosine(g["eps_x"][4:7], g["eps_y"], eps=1e-2)
    np.testing.assert_allclose(got, g["pairwise_cosine_eps_1e-2"][4:7], rtol=RTOL, atol=2e-6)


def test_hamming_and_angular(nat):
    from vectordb_amd.distance import get_metric_fn, hamming, angular_distance, list_metrics
    g = load_golden("round2.npz")
    assert "hamming" in list_metrics() and "angular" in list_metrics()
    store = nat.RowStore(24)
    store.append(g["ham_x"])
    np.testing.assert_array_equal(store.distances(g["ham_q"], nat.HAMMING)[0], g["hamming"])       # counts: exact
    dist, rows = store.search(g["ham_q"][None, :], 5, nat.HAMMING)
    ref = R.topk_stable(g["hamming"], 5)
    np.testing.assert_array_equal(rows[0], ref)
    store.close()
    store = nat.RowStore(24)
    store.append(g["x"][:400])
    got = store.distances(g["q"][2], nat.ANGULAR)[0]
    np.testing.assert_allclose(got, g["angular"], rtol=RTOL, atol=2e-6)
    store.close()
    assert get_metric_fn("hamming")(g["ham_q"], g["ham_x"][3]) == g["hamming"][3] == hamming(g["ham_q"], g["ham_x"][3])
    assert angular_distance(g["q"][2], g["x"][9]) == pytest.approx(g["angular"][9], rel=RTOL)


def test_angular_ivf_index():
    """IVFIndex(metric="angular") reaches the metric through get_metric_fn in the reference (index/ivf.py:611-613)."""
    from vectordb_amd.index import IVFIndex
    g = load_golden("round2.npz")
    x, q = g["x"], g["q"]
    idx = IVFIndex(dimension=24, metric="angular", n_clusters=16, n_probe=4, seed=3)
    idx._set_centroids(g["angular_ivf_centroids"])
    idx.add_batch(ids_of(len(x)), x)
    got = np.array([idx._id_to_cluster[i] for i in ids_of(len(x))])
    ref = g["angular_ivf_assign"]
    moved = np.flatnonzero(got != ref)
    assert len(moved) <= 2
    if len(moved):
        cent = g["angular_ivf_centroids"]
        for i in moved:
            d = R.rows_scalar_metric(x[i], cent, "angular")
            assert abs(d[got[i]] - d[ref[i]]) <= 1e-5
        idx = IVFIndex(dimension=24, metric="angular", n_clusters=16, n_probe=4, seed=3)
        idx._set_centroids(cent)
        for i in range(len(x)):
            idx._append(f"v{i}", x[i], {}, int(ref[i]))
    for i in range(len(q)):
        res = idx.search(q[i], k=10, n_probe=4)
        want_rows, want_dist = g["angular_ivf_rows"][i], g["angular_ivf_dist"][i]
        n = int((want_rows >= 0).sum())
        assert sorted(rows_of(res)) == sorted(want_rows[:n].tolist())
        np.testing.assert_allclose([r.distance for r in res], want_dist[:n], rtol=RTOL, atol=2e-6)


def test_simd_seam():
    from vectordb_amd.distance.simd import get_optimized_distance_fn, simd_euclidean_batch, numpy_dot_batch
    g = load_golden("round2.npz")
    q, x = g["q"][3], g["x"][:500]
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        fn = get_optimized_distance_fn(metric)
        got = fn(q, x)
        assert got.shape == (500,) and got.dtype == np.float32
        if metric in ("euclidean", "manhattan"):
            np.testing.assert_array_equal(got, g[f"simd_{metric}"])          # direct forms: NumPy's bits
        else:
            np.testing.assert_allclose(got, g[f"simd_{metric}"], rtol=RTOL, atol=2e-5)
    assert get_optimized_distance_fn("euclidean", use_numba=False) is simd_euclidean_batch
    assert (numpy_dot_batch(q, x) > 0).any()                                 # the positive dot product
    with pytest.raises(ValueError, match="Unknown metric"):
        get_optimized_distance_fn("chebyshev")


def test_zero_cosine_query_dtype():
    from vectordb_amd.distance import query_distances
    g = load_golden("round2.npz")
    z = query_distances(np.zeros(24, dtype=np.float32), g["x"][:10], "cosine")
    assert str(z.dtype) == str(g["zero_query_cosine_dtype"])
    np.testing.assert_array_equal(z, g["zero_query_cosine"])
    assert query_distances(g["q"][0], g["x"][:10], "cosine").dtype == np.float32
