"""Pin the CPU oracle (oracle/ref_numpy.py) against golden vectors that were
produced by the real reference (tests/golden/make_golden.py).

Bit-exact where the reference's own result is reproducible (element-wise fp32
ops + NumPy pairwise sums); rounding tolerance where the reference goes
through BLAS, whose bits depend on the host CPU's kernels.
"""

import numpy as np
import pytest

from oracle import ref_numpy as R
from conftest import load_golden, same_ids_or_near_ties

BLAS_RTOL, BLAS_ATOL = 2e-5, 2e-5
METRICS6 = ("euclidean", "euclidean_squared", "cosine", "dot", "manhattan", "chebyshev")
EXACT = ("manhattan", "chebyshev")


def close(a, b, rtol=BLAS_RTOL, atol=BLAS_ATOL):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- known answers
# restated from the reference's tests/unit/test_metrics.py:46-50,80-92,116-132,
# 153-157,174-178,255-279

def test_known_answers():
    z2, p34 = np.array([0.0, 0.0]), np.array([3.0, 4.0])
    assert R.rows_scalar_metric(z2, p34[None], "euclidean")[0] == pytest.approx(5.0)
    assert R.rows_scalar_metric(z2, p34[None], "manhattan")[0] == pytest.approx(7.0)
    assert R.rows_scalar_metric(z2, p34[None], "chebyshev")[0] == pytest.approx(4.0)
    assert R.rows_scalar_metric(z2, p34[None], "euclidean_squared")[0] == pytest.approx(25.0)
    e0, e1 = np.array([1.0, 0.0]), np.array([0.0, 1.0])
    assert R.rows_scalar_metric(e0, np.array([e0, e1, -e0]), "cosine") == pytest.approx([0.0, 1.0, 2.0])
    assert R.rows_scalar_metric(np.array([1.0, 2.0]), np.array([[3.0, 4.0]]), "dot")[0] == pytest.approx(-11.0)
    q = np.zeros(3)
    c = np.diag([1.0, 2.0, 3.0])
    close(R.query_distances(q, c, "euclidean"), [1.0, 2.0, 3.0])
    close(R.query_distances(e0, np.array([e0, e1, -e0]), "cosine"), [0.0, 1.0, 2.0])
    with pytest.raises(ValueError):
        R.query_distances(q, c, "nope")
    with pytest.raises(KeyError):
        R.rows_scalar_metric(q, c, "nope")


# ---------------------------------------------------------------- metrics.npz

@pytest.mark.parametrize("seed", [0, 42, 123])
def test_query_distances_golden(seed):
    g = load_golden("metrics.npz")
    x, q = g[f"s{seed}_x"], g[f"s{seed}_q"]
    for metric in METRICS6:
        got, ref = R.query_distances(q, x, metric), g[f"s{seed}_qd_{metric}"]
        assert got.dtype == ref.dtype and got.shape == ref.shape
        if metric in EXACT:
            np.testing.assert_array_equal(got, ref)
        else:
            close(got, ref)
    z = R.query_distances(np.zeros(32, np.float32), x, "cosine")
    np.testing.assert_array_equal(z, g[f"s{seed}_qd_cosine_zeroq"])
    assert z.dtype == np.float64


@pytest.mark.parametrize("seed", [0, 42, 123])
def test_pairwise_golden(seed):
    g = load_golden("metrics.npz")
    x = g[f"s{seed}_x"]
    a, b = x[:20], x[40:75]
    close(R.pairwise_euclidean(a, b), g[f"s{seed}_pw_euclidean"])
    close(R.pairwise_euclidean_squared(a, b), g[f"s{seed}_pw_euclidean_squared"])
    close(R.pairwise_cosine(a, b), g[f"s{seed}_pw_cosine"])
    close(R.pairwise_dot(a, b), g[f"s{seed}_pw_dot"])
    close(R.pairwise_euclidean(a), g[f"s{seed}_pw_euclidean_self"], atol=5e-3)  # cancellation at 0
    np.testing.assert_array_equal(R.pairwise_manhattan(a, b), g[f"s{seed}_pw_manhattan_small"])
    np.testing.assert_array_equal(R.pairwise_manhattan(x[:100], x[100:200]), g[f"s{seed}_pw_manhattan_large"])


@pytest.mark.parametrize("seed", [0, 42, 123])
def test_scalar_functions_golden(seed):
    g = load_golden("metrics.npz")
    x, q = g[f"s{seed}_x"][:64], g[f"s{seed}_q"]
    for metric in METRICS6:
        got, ref = R.rows_scalar_metric(q, x, metric), g[f"s{seed}_scalar_{metric}"]
        if metric in ("euclidean", "manhattan", "chebyshev"):
            np.testing.assert_array_equal(got, ref)          # bit-identical
        else:
            close(got, ref)


@pytest.mark.parametrize("d", [3, 8, 13, 100, 130, 384, 768])
def test_dims_and_pairwise_sum_order(d):
    g = load_golden("metrics.npz")
    x, q = g[f"d{d}_x"], g[f"d{d}_q"]
    np.testing.assert_array_equal(np.sum(x ** 2, axis=1), g[f"d{d}_sqnorm"])
    emu = np.array([R.pairwise_sum_f32(r * r) for r in x[:6]])
    np.testing.assert_array_equal(emu, g[f"d{d}_sqnorm"][:6])
    np.testing.assert_array_equal(R.rows_scalar_metric(q, x, "euclidean"), g[f"d{d}_scalar_euclidean"])
    for metric in ("euclidean", "cosine", "dot", "manhattan", "chebyshev"):
        got, ref = R.query_distances(q, x, metric), g[f"d{d}_qd_{metric}"]
        if metric in EXACT:
            np.testing.assert_array_equal(got, ref)
        else:
            close(got, ref)


# ---------------------------------------------------------------- flat

def _config1_data():
    np.random.seed(42)
    x = np.random.randn(10000, 128).astype(np.float32)
    np.random.seed(123)
    q = np.random.randn(100, 128).astype(np.float32)
    return x, q


def test_flat_config1_golden():
    g = load_golden("flat_config1.npz")
    x, q = _config1_data()
    assert x.sum(dtype=np.float64) == g["x_checksum"][0]
    assert q.sum(dtype=np.float64) == g["q_checksum"][0]
    for i in range(100):
        rows, dist = R.flat_search(q[i], x, 10, "euclidean")
        full = R.query_distances(q[i], x, "euclidean")
        assert same_ids_or_near_ties(rows, g["flat_rows"][i], lambda r: full[r], g["flat_dist"][i][-1])
        close(dist, g["flat_dist"][i])
        close(R.flat_scores(dist), g["flat_score"][i])
        # Collection.search: stable sort, earliest row wins ties
        assert same_ids_or_near_ties(rows, g["coll_rows"][i], lambda r: full[r], g["coll_dist"][i][-1])
        close(dist, g["coll_dist"][i])
        close(R.collection_scores(dist), g["coll_score"][i], atol=1e-5)
    mask = (np.arange(len(x)) % 4) == 1
    for i in range(20):
        rows, dist = R.flat_search(q[i], x, 10, "euclidean", mask=mask)
        # the reference gathers the candidate rows first (core/collection.py:740-741),
        # so its gemv sees a different matrix: values equal to rounding
        assert set(rows) == set(g["coll_filter_rows"][i])
        close(dist, g["coll_filter_dist"][i])


def test_flat_small_golden():
    g = load_golden("flat_small.npz")
    x, q, mask = g["x"], g["q"], g["mask"]
    for metric in METRICS6:
        for k in (1, 10, 37):
            for i in range(len(q)):
                rows, dist = R.flat_search(q[i], x, k, metric)
                full = R.query_distances(q[i], x, metric)
                ref_rows, ref_dist = g[f"{metric}_k{k}_rows"][i], g[f"{metric}_k{k}_dist"][i]
                assert same_ids_or_near_ties(rows, ref_rows, lambda r: full[r], ref_dist[-1], atol=1e-3)
                if metric in EXACT:
                    np.testing.assert_array_equal(dist, ref_dist.astype(dist.dtype))
                else:
                    close(dist, ref_dist, atol=2e-3 if metric == "euclidean" else BLAS_ATOL)
        for i in range(len(q)):
            rows, dist = R.flat_search(q[i], x, 10, metric, mask=mask)
            full = R.query_distances(q[i], x, metric)
            assert same_ids_or_near_ties(rows, g[f"{metric}_filter_rows"][i], lambda r: full[r],
                                         g[f"{metric}_filter_dist"][i][-1], atol=1e-3)
            close(dist, g[f"{metric}_filter_dist"][i], atol=2e-3 if metric == "euclidean" else BLAS_ATOL)
            assert mask[rows].all()
    # exact duplicates of row 3 sit at rows 100 and 900; query 0 equals row 3
    rows, _ = R.flat_search(q[0], x, 3, "manhattan")
    assert list(rows) == [3, 100, 900]
    # normalize=True (index/flat.py:243-246, 314-317)
    xn = x / np.where(np.linalg.norm(x, axis=1, keepdims=True) < 1e-8, 1.0, np.linalg.norm(x, axis=1, keepdims=True))
    for i in range(len(q)):
        nq = np.linalg.norm(q[i])
        qq = q[i] / nq if nq > 1e-8 else q[i]
        rows, dist = R.flat_search(qq, xn, 10, "euclidean")
        assert set(rows) == set(g["normalize_rows"][i])
        close(dist, g["normalize_dist"][i], atol=2e-3)
    for k in (30, 60, 65):
        rows, dist = R.flat_search(q[1], x[:60], k, "euclidean")
        assert len(rows) == min(k, 60)
        np.testing.assert_array_equal(np.sort(rows), np.sort(g[f"small_k{k}_rows"]))
        close(dist, g[f"small_k{k}_dist"])
    d = R.query_distances(q[2], x, "euclidean")
    inside = np.flatnonzero(d <= 8.5)
    assert set(inside) == set(g["range_rows"])


def test_batch_topk_golden():
    g = load_golden("topk.npz")
    x, q = g["x"], g["q"]
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        for n in (50, 9000, 12000):
            ks = (1, 10, n, n + 5) if n == 50 else (1, 10)
            for k in ks:
                idx, dist = R.batch_top_k(q, x[:n], k, metric)
                ref_idx, ref_dist = g[f"{metric}_n{n}_k{k}_idx"], g[f"{metric}_n{n}_k{k}_dist"]
                assert idx.dtype == np.int64 and len(idx) == min(k, n)
                np.testing.assert_array_equal(idx, ref_idx)
                close(dist, ref_dist)


# ---------------------------------------------------------------- k-means / IVF

def test_kmeans_golden():
    g = load_golden("kmeans.npz")
    x = g["x"]
    rng = np.random.RandomState(42)
    close(R.kmeans_pp_init(x, 12, rng), g["pp_init"], rtol=0, atol=0)
    cent = R.kmeans_fit(x, 12, 20, seed=42)
    close(cent, g["centroids"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(R.kmeans_assign(x, g["centroids"]), g["labels"])
    np.testing.assert_array_equal(R.kmeans_fit(x[:20], 50, seed=1), g["few_centroids"])


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan", "dot"])
def test_ivf_golden(metric):
    g = load_golden("ivf.npz")
    x, q, mask = g["x"], g["q"], g["mask"]
    cent = g[f"{metric}_centroids"]
    assign = R.ivf_assign_batch(x, cent, metric)
    np.testing.assert_array_equal(assign, g[f"{metric}_assign_batch"])
    single = np.array([R.ivf_assign_single(v, cent, metric) for v in x[:300]])
    np.testing.assert_array_equal(single, g[f"{metric}_assign_single"])
    lists = R.ivf_build_lists(assign, len(cent))
    exact = metric in ("euclidean", "manhattan")
    for npb in (1, 8, 32):
        for i in range(len(q)):
            rows, dist = R.ivf_search(q[i], x, cent, lists, 10, npb, metric)
            ref_rows, ref_dist = g[f"{metric}_np{npb}_rows"][i], g[f"{metric}_np{npb}_dist"][i]
            n = int((ref_rows >= 0).sum())
            assert len(rows) == n
            if exact:
                np.testing.assert_array_equal(rows, ref_rows[:n])
                np.testing.assert_array_equal(dist, ref_dist[:n])
            else:
                np.testing.assert_array_equal(np.sort(rows), np.sort(ref_rows[:n]))
                close(dist, ref_dist[:n])
    for i in range(len(q)):
        rows, dist = R.ivf_search(q[i], x, cent, lists, 10, 8, metric, mask=mask)
        ref_rows, ref_dist = g[f"{metric}_filter_rows"][i], g[f"{metric}_filter_dist"][i]
        n = int((ref_rows >= 0).sum())
        if exact:
            np.testing.assert_array_equal(rows, ref_rows[:n])
            np.testing.assert_array_equal(dist, ref_dist[:n])
        else:
            np.testing.assert_array_equal(np.sort(rows), np.sort(ref_rows[:n]))
            close(dist, ref_dist[:n])


def test_ivf_train_golden():
    g = load_golden("ivf.npz")
    cent = R.kmeans_fit(g["x"], 32, 20, seed=42)
    close(cent, g["euclidean_centroids"], rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- PQ

def test_pq_golden():
    g = load_golden("pq.npz")
    x, q, mask, book = g["x"], g["q"], g["mask"], g["codebook"]
    np.testing.assert_array_equal(R.pq_encode(x, book), g["codes"])
    for i in range(len(q)):
        lut = R.pq_lut(q[i], book)
        full = R.pq_adc(g["codes"], lut)
        assert full.dtype == np.float32
        np.testing.assert_array_equal(full, g["full_dist"][i])
        rows, dist = R.pq_search(q[i], book, g["codes"], 10)
        np.testing.assert_array_equal(rows, g["rows"][i])
        np.testing.assert_array_equal(dist, g["dist"][i])
        rows, dist = R.pq_search(q[i], book, g["codes"], 10, mask=mask)
        np.testing.assert_array_equal(rows, g["filter_rows"][i])
        np.testing.assert_array_equal(dist, g["filter_dist"][i])
    sbook = g["small_codebook"]
    np.testing.assert_array_equal(R.pq_encode(x[:300], sbook), g["small_codes"])
    rows, dist = R.pq_search(q[0], sbook, g["small_codes"], 10)
    np.testing.assert_array_equal(rows, g["small_rows"])
    np.testing.assert_array_equal(dist, g["small_dist"])


def test_pq_train_golden():
    g = load_golden("pq.npz")
    book = R.pq_train(g["x"], 4, 8, 20, seed=42)
    assert book.dtype == np.float32 and book.shape == g["codebook"].shape
    close(book, g["codebook"], rtol=1e-5, atol=1e-5)
    small = R.pq_train(g["x"][:100], 8, 8, 20, seed=1)
    close(small, g["small_codebook"], rtol=1e-5, atol=1e-5)
    assert not small[:, 100:].any()          # zero padding beyond the 100 trained codewords


# ------------------------------------------------------------------ HybridIndex (IVF-PQ)
def _hybrid_members(lists, n_lists, removed=()):
    keep = np.ones(len(lists), dtype=bool)
    keep[list(removed)] = False
    return [np.flatnonzero((lists == l) & keep).astype(np.int64) for l in range(n_lists)]


def test_hybrid_train_encode_match_reference():
    g = load_golden("hybrid.npz")
    x = g["x"]
    cent, book = R.hybrid_train(x, 16, 4, n_bits=8, n_iter=int(g["n_iter"]), seed=11)
    np.testing.assert_allclose(cent, g["centroids"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(book, g["codebook"], rtol=1e-4, atol=1e-5)
    lists, codes = R.hybrid_encode(x, g["centroids"], g["codebook"])
    assert np.array_equal(lists, g["lists"])
    assert np.array_equal(codes, g["codes"])


def test_hybrid_search_matches_reference():
    g = load_golden("hybrid.npz")
    members = _hybrid_members(g["lists"], len(g["centroids"]))
    for tag, n_probe in (("p4", 4), ("p16", 16)):
        for i, q in enumerate(g["q"]):
            rows, dist = R.hybrid_search(q, g["centroids"], g["codebook"], g["codes"], members, 10, n_probe)
            assert np.array_equal(rows, g[f"rows_{tag}"][i][:len(rows)])
            assert np.array_equal(dist, g[f"dist_{tag}"][i][:len(rows)])
            rows, dist = R.hybrid_search(q, g["centroids"], g["codebook"], g["codes"], members, 10, n_probe, mask=g["mask"])
            assert np.array_equal(rows, g[f"filter_rows_{tag}"][i][:len(rows)])
            assert np.array_equal(dist, g[f"filter_dist_{tag}"][i][:len(rows)])
    members = _hybrid_members(g["lists"], len(g["centroids"]), removed=g["removed"])
    rows, dist = R.hybrid_search(g["q"][0], g["centroids"], g["codebook"], g["codes"], members, 10, 16)
    assert np.array_equal(rows, g["rows_after_remove"]) and np.array_equal(dist, g["dist_after_remove"])


# ---------------------------------------------------------------- cases added with round2.npz
def test_range_search_max_results_golden():
    """FlatIndex.range_search keeps the FIRST max_results rows in insertion order inside the radius, then sorts
    them (index/flat.py:545-570)."""
    g = load_golden("round2.npz")
    x, q, mask = g["x"], g["q"], g["mask"]
    for metric in ("euclidean", "manhattan", "cosine", "dot"):
        radius = float(g[f"range_{metric}_radius"])
        qq = q[0] if metric != "dot" else q[1]
        for tag, kw in (("all", {}), ("max5", {"max_results": 5}), ("max40", {"max_results": 40}),
                        ("filter_max10", {"max_results": 10, "mask": mask})):
            rows, dist = R.flat_range_search(qq, x, radius, metric, **kw)
            want_rows, want_dist = g[f"range_{metric}_{tag}_rows"], g[f"range_{metric}_{tag}_dist"]
            if metric == "manhattan" or tag != "all":
                # the kept SET is fixed by insertion order; the order inside it by (distance, row)
                np.testing.assert_array_equal(rows, want_rows)
            else:
                assert set(rows.tolist()) == set(want_rows.tolist())
            np.testing.assert_allclose(np.sort(dist.astype(np.float64)), np.sort(want_dist), rtol=2e-5, atol=2e-6)


def test_pairwise_cosine_eps_golden():
    g = load_golden("round2.npz")
    for tag, eps in (("default", 1e-8), ("1e-3", 1e-3), ("1e-2", 1e-2)):
        np.testing.assert_allclose(R.pairwise_cosine(g["eps_x"], g["eps_y"], eps=eps), g[f"pairwise_cosine_eps_{tag}"],
                                   rtol=2e-5, atol=2e-6)
    # the eps matters: the three matrices differ
    assert not np.allclose(g["pairwise_cosine_eps_default"], g["pairwise_cosine_eps_1e-2"], atol=1e-3)


def test_hamming_angular_simd_zero_query_golden():
    g = load_golden("round2.npz")
    np.testing.assert_array_equal(R.rows_scalar_metric(g["ham_q"], g["ham_x"], "hamming"), g["hamming"])
    np.testing.assert_allclose(R.rows_scalar_metric(g["q"][2], g["x"][:400], "angular"), g["angular"], rtol=1e-5, atol=1e-6)
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        got = R.simd_batch(metric, g["q"][3], g["x"][:500])
        if metric in ("euclidean", "manhattan"):
            np.testing.assert_array_equal(got, g[f"simd_{metric}"])
        else:
            np.testing.assert_allclose(got, g[f"simd_{metric}"], rtol=2e-5, atol=2e-6)
    z = R.query_distances(np.zeros(24, dtype=np.float32), g["x"][:10], "cosine")
    assert str(z.dtype) == str(g["zero_query_cosine_dtype"]) == "float64"
    np.testing.assert_array_equal(z, g["zero_query_cosine"])


def test_angular_ivf_golden():
    g = load_golden("round2.npz")
    x, q, cent, assign = g["x"], g["q"], g["angular_ivf_centroids"], g["angular_ivf_assign"]
    got_assign = R.ivf_assign_batch(x, cent, "angular")
    assert np.mean(got_assign == assign) >= 0.999
    lists = R.ivf_build_lists(assign, len(cent))
    for i in range(len(q)):
        rows, dist = R.ivf_search(q[i], x, cent, lists, 10, 4, "angular")
        np.testing.assert_array_equal(rows, g["angular_ivf_rows"][i][:len(rows)])
        np.testing.assert_allclose(dist, g["angular_ivf_dist"][i][:len(rows)], rtol=1e-5, atol=1e-6)


def test_training_steps_golden():
    """k-means one step at a time: KMeans (ivf.py:165-202), the PQ sub-quantizer (pq.py:144-180, float64 from the
    second step on) and the IVF-PQ sub-quantizer (hybrid.py:174-203)."""
    g = load_golden("round2.npz")
    x = g["km_x"]
    np.testing.assert_array_equal(R.kmeans_pp_init(x, 12, np.random.RandomState(42)), g["km_pp_init"])
    np.testing.assert_array_equal(R.kmeans_assign(x, g["km_pp_init"]), g["km_assign_of_pp_init"])
    for steps in (1, 2):
        np.testing.assert_allclose(R.kmeans_fit(x, 12, n_iter=steps, seed=42), g[f"km_centroids_after_{steps}"],
                                   rtol=1e-6, atol=1e-6)
    sub = g["pq_sub"]
    np.testing.assert_array_equal(np.random.RandomState(9).choice(len(sub), 64, replace=False), g["pq_init_rows"])
    for steps in (1, 2, 3, 20):
        got = R.pq_train_subquantizer(sub, 64, steps, np.random.RandomState(9))
        assert got.dtype == g[f"pq_centroids_after_{steps}"].dtype == np.float64
        np.testing.assert_allclose(got, g[f"pq_centroids_after_{steps}"], rtol=1e-9, atol=1e-9)
    for steps in (1, 2, 5):
        got = R.hybrid_train_subquantizer(sub, 64, steps, np.random.RandomState(9))
        np.testing.assert_allclose(got, g[f"hy_centroids_after_{steps}"], rtol=1e-9, atol=1e-9)
