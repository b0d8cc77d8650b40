"""Parity of the HIP kernels, called through the C ABI, with the CPU oracle and
with the golden vectors produced by the reference.

Bars (written next to each check):
  * bit-exact: row norms, Manhattan, Chebyshev, direct-form Euclidean, PQ codes /
    tables / ADC distances, and every returned row number on those paths;
  * fp32 dot-product metrics (BLAS in the reference, MFMA/fma here): distances
    within 1e-4 relative (plus the cancellation floor of the identity form near
    zero), id sets identical except rows whose reference distance ties the k-th
    reference distance to rounding.
"""

import numpy as np
import pytest

from oracle import ref_numpy as R
from conftest import load_golden, same_ids_or_near_ties

pytestmark = pytest.mark.gpu

RTOL = 1e-4
METRICS6 = ("euclidean", "euclidean_squared", "cosine", "dot", "manhattan", "chebyshev")
EXACT = ("manhattan", "chebyshev")


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def identity_atol(metric, q, x):
    """Rounding floor of the dot-product forms: a few ulps of the large terms
    that cancel in |q|^2+|c|^2-2c.q (squared domain for both Euclidean
    metrics), of sum|c_i q_i| for dot, of 1.0 for cosine."""
    if metric in ("euclidean", "euclidean_squared"):
        return 4e-6 * float(np.max(np.sum(x * x, axis=1)) + np.sum(q * q))
    if metric == "dot":
        return 1e-5 * float(np.max(np.abs(x).sum(axis=1)) * np.max(np.abs(q)))
    return 2e-6


def assert_close(metric, got, ref, q, x):
    """1e-4 relative, plus the cancellation floor.  Euclidean is compared in the
    squared domain, where the identity form's error is additive."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    atol = identity_atol(metric, q, x)
    if metric == "euclidean":
        np.testing.assert_allclose(got ** 2, ref ** 2, rtol=2 * RTOL, atol=atol)
    else:
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=atol)


def ids_atol(metric, q, x, kth):
    """Distance slack within which two rows count as tied at the k-th place."""
    a = identity_atol(metric, q, x)
    if metric == "euclidean":
        return a / max(2.0 * abs(kth), np.sqrt(a))
    return a


def check_search(nat, x, queries, k, metric, mask=None):
    store = nat.RowStore(x.shape[1])
    store.append(x)
    words = nat.pack_mask(mask)
    dist, rows = store.search(queries, k, nat.METRIC_CODES[metric], words)
    for i, q in enumerate(np.atleast_2d(queries)):
        ref_rows, ref_dist = R.flat_search(q, x, k, metric, mask=mask)
        full = R.query_distances(q, x, metric)
        n_ref = len(ref_rows)
        assert (rows[i, n_ref:] == -1).all() and np.isinf(dist[i, n_ref:]).all()
        got_rows, got_dist = rows[i, :n_ref], dist[i, :n_ref]
        assert len(set(got_rows)) == n_ref
        if mask is not None:
            assert mask[got_rows].all()
        if n_ref == 0:                  # a filter that lets no row pass: nothing but the sentinels checked above
            continue
        if metric in EXACT:
            np.testing.assert_array_equal(got_rows, ref_rows)
            np.testing.assert_array_equal(got_dist, ref_dist.astype(np.float32))
        else:
            atol = ids_atol(metric, q, x, float(ref_dist[-1]))
            assert same_ids_or_near_ties(got_rows, ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
            assert_close(metric, got_dist, ref_dist, q, x)
            assert (np.diff(got_dist) >= 0).all()
    store.close()
    return dist, rows


# ------------------------------------------------------------------ norms

@pytest.mark.parametrize("d", [1, 3, 7, 8, 13, 32, 100, 128, 130, 384, 768, 1000])
def test_row_norms_bit_exact(nat, d):
    rng = np.random.RandomState(d)
    x = rng.randn(700, d).astype(np.float32)
    store = nat.RowStore(d)
    store.append(x[:300])
    store.append(x[300:])           # second append grows the store
    sq, nr = store.norms()
    np.testing.assert_array_equal(sq, np.sum(x ** 2, axis=1))                 # metrics.py:456
    np.testing.assert_array_equal(nr, np.linalg.norm(x, axis=1))              # metrics.py:470
    store.close()


def test_row_norms_golden(nat):
    g = load_golden("metrics.npz")
    for d in (3, 8, 13, 100, 130, 384, 768):
        store = nat.RowStore(d)
        store.append(g[f"d{d}_x"])
        sq, _ = store.norms()
        np.testing.assert_array_equal(sq, g[f"d{d}_sqnorm"])
        store.close()


# ------------------------------------------------------------------ flat search

@pytest.mark.parametrize("metric", METRICS6)
@pytest.mark.parametrize("nq", [1, 3, 12])        # 1,3: streaming scan; 12: matrix-core path (dot metrics)
def test_flat_search_small_golden_inputs(nat, metric, nq):
    g = load_golden("flat_small.npz")
    x, q, mask = g["x"], g["q"][:nq], g["mask"]
    for k in (1, 10, 37):                          # 37 > 32: two selection passes
        check_search(nat, x, q, k, metric)
    check_search(nat, x, q, 10, metric, mask=mask)


@pytest.mark.parametrize("metric", METRICS6)
def test_flat_search_vs_reference_golden(nat, metric):
    """Against the reference's own FlatIndex.search outputs."""
    g = load_golden("flat_small.npz")
    x, q, mask = g["x"], g["q"], g["mask"]
    store = nat.RowStore(x.shape[1])
    store.append(x)
    code = nat.METRIC_CODES[metric]
    for k in (1, 10, 37):
        dist, rows = store.search(q, k, code)
        for i in range(len(q)):
            full = R.query_distances(q[i], x, metric)
            ref_rows, ref_dist = g[f"{metric}_k{k}_rows"][i], g[f"{metric}_k{k}_dist"][i]
            atol = ids_atol(metric, q[i], x, float(ref_dist[-1]))
            assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
            if metric in EXACT:
                np.testing.assert_array_equal(dist[i], ref_dist.astype(np.float32))
            else:
                assert_close(metric, dist[i], ref_dist, q[i], x)
    dist, rows = store.search(q, 10, code, nat.pack_mask(mask))
    for i in range(len(q)):
        full = R.query_distances(q[i], x, metric)
        ref_rows, ref_dist = g[f"{metric}_filter_rows"][i], g[f"{metric}_filter_dist"][i]
        atol = ids_atol(metric, q[i], x, float(ref_dist[-1]))
        assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
        assert_close(metric, dist[i], ref_dist, q[i], x)
    store.close()


def test_flat_ties_go_to_lowest_row(nat):
    g = load_golden("flat_small.npz")
    x, q = g["x"], g["q"]
    store = nat.RowStore(x.shape[1])
    store.append(x)
    # rows 3, 100, 900 are identical and query 0 equals them
    for metric in ("manhattan", "chebyshev", "dot", "cosine"):
        for batch in (q[:1], q[:8]):
            _, rows = store.search(batch, 3, nat.METRIC_CODES[metric])
            if metric in ("manhattan", "chebyshev"):
                assert list(rows[0]) == [3, 100, 900]
            else:
                # the three copies have bit-identical distances on every path
                d3 = R.query_distances(q[0], x, metric)
                if np.argsort(d3, kind="stable")[0] == 3:
                    assert list(rows[0]) == [3, 100, 900]
    store.close()


def test_flat_config1_golden(nat):
    """BASELINE config 1 (10k x 128 euclidean, 100 queries, k=10) against the
    reference's FlatIndex.search and Collection.search results."""
    g = load_golden("flat_config1.npz")
    np.random.seed(42)
    x = np.random.randn(10000, 128).astype(np.float32)
    np.random.seed(123)
    q = np.random.randn(100, 128).astype(np.float32)
    store = nat.RowStore(128)
    store.append(x)
    for batch in (True, False):
        if batch:
            dist, rows = store.search(q, 10, nat.EUCLIDEAN)
        else:
            out = [store.search(q[i], 10, nat.EUCLIDEAN) for i in range(100)]
            dist = np.concatenate([o[0] for o in out]); rows = np.concatenate([o[1] for o in out])
        for i in range(100):
            full = R.query_distances(q[i], x, "euclidean")
            for name in ("flat", "coll"):
                ref_rows, ref_dist = g[f"{name}_rows"][i], g[f"{name}_dist"][i]
                assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL)
                np.testing.assert_allclose(dist[i], ref_dist, rtol=RTOL)
    mask = (np.arange(len(x)) % 4) == 1
    dist, rows = store.search(q[:20], 10, nat.EUCLIDEAN, nat.pack_mask(mask))
    for i in range(20):
        assert set(rows[i]) == set(g["coll_filter_rows"][i])
        np.testing.assert_allclose(dist[i], g["coll_filter_dist"][i], rtol=RTOL)
    store.close()


@pytest.mark.parametrize("n,d,nq", [(1, 4, 1), (5, 2, 2), (127, 33, 5), (129, 64, 130), (1000, 100, 17), (4097, 20, 4)])
def test_flat_ragged_shapes(nat, n, d, nq):
    rng = np.random.RandomState(n + d)
    x = rng.randn(n, d).astype(np.float32)
    q = rng.randn(nq, d).astype(np.float32)
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        for k in (1, 5, n + 3):                   # k > n is capped; the tail reads -1 / inf
            check_search(nat, x, q, k, metric)


def test_flat_zero_query_and_zero_rows_cosine(nat):
    rng = np.random.RandomState(9)
    x = rng.randn(500, 16).astype(np.float32)
    x[5] = 0.0
    q = np.zeros((6, 16), np.float32)
    q[1:] = rng.randn(5, 16)
    for nq in (1, 6):
        store = nat.RowStore(16)
        store.append(x)
        dist, rows = store.search(q[:nq], 4, nat.COSINE)
        np.testing.assert_array_equal(rows[0], [0, 1, 2, 3])            # all ones -> first rows (metrics.py:472-473)
        np.testing.assert_array_equal(dist[0], np.ones(4, np.float32))
        full = store.distances(q[:nq], nat.COSINE)
        assert full[0].tolist() == [1.0] * 500
        if nq > 1:
            np.testing.assert_allclose(full[1], R.query_distances(q[1], x, "cosine"), rtol=RTOL, atol=2e-6)
            assert full[1][5] == 1.0                                    # zero row: norm replaced by 1 -> similarity 0
        store.close()


def test_flat_empty_and_errors(nat):
    store = nat.RowStore(8)
    dist, rows = store.search(np.zeros((2, 8), np.float32), 3, nat.EUCLIDEAN)
    assert (rows == -1).all() and np.isinf(dist).all()
    with pytest.raises(ValueError):
        store.append(np.zeros((3, 7), np.float32))
    with pytest.raises(ValueError):
        store.search(np.zeros((1, 8), np.float32), 3, 99)
    with pytest.raises(ValueError):
        store.search(np.zeros((1, 8), np.float32), 0, nat.EUCLIDEAN)
    store.close()


def test_flat_update_and_swap_remove(nat):
    """index/flat.py:596-609: the last row is copied over the removed one and
    the matrix shrinks by one."""
    rng = np.random.RandomState(2)
    x = rng.randn(300, 24).astype(np.float32)
    q = rng.randn(5, 24).astype(np.float32)
    store = nat.RowStore(24)
    store.append(x)
    y = x.copy()
    y[17] = y[-1]
    y = y[:-1]
    store.set_row(17, x[-1])
    store.truncate(299)
    for metric in ("euclidean", "manhattan"):
        dist, rows = store.search(q, 7, nat.METRIC_CODES[metric])
        for i in range(5):
            ref_rows, ref_dist = R.flat_search(q[i], y, 7, metric)
            np.testing.assert_array_equal(rows[i], ref_rows)
            np.testing.assert_allclose(dist[i], ref_dist, rtol=RTOL)
    sq, _ = store.norms()
    np.testing.assert_array_equal(sq, np.sum(y ** 2, axis=1))
    store.close()


# ------------------------------------------------------------------ full distance matrices

@pytest.mark.parametrize("nq", [1, 2, 40])
def test_distances_golden(nat, nq):
    g = load_golden("metrics.npz")
    for seed in (0, 42, 123):
        x, q = g[f"s{seed}_x"], g[f"s{seed}_q"]
        qs = np.stack([q] + [x[i] for i in range(nq - 1)])
        store = nat.RowStore(x.shape[1])
        store.append(x)
        for metric in METRICS6:
            out = store.distances(qs, nat.METRIC_CODES[metric])
            ref = g[f"s{seed}_qd_{metric}"]
            if metric in EXACT:
                np.testing.assert_array_equal(out[0], ref)
            else:
                assert_close(metric, out[0], ref, q, x)
            for j in range(1, nq):
                assert_close(metric, out[j], R.query_distances(qs[j], x, metric), qs[j], x)
        # direct forms (the per-pair scalar functions, metrics.py:43,190,214)
        out = store.distances(q, nat.EUCLIDEAN_DIRECT)
        np.testing.assert_array_equal(out[0][:64].astype(np.float64), g[f"s{seed}_scalar_euclidean"])
        out = store.distances(q, nat.COSINE_PAIR)
        np.testing.assert_allclose(out[0][:64], g[f"s{seed}_scalar_cosine"], rtol=RTOL, atol=2e-6)
        store.close()


@pytest.mark.parametrize("d", [3, 8, 13, 100, 130, 384, 768])
def test_distances_dims_golden(nat, d):
    g = load_golden("metrics.npz")
    x, q = g[f"d{d}_x"], g[f"d{d}_q"]
    store = nat.RowStore(d)
    store.append(x)
    for metric in ("euclidean", "cosine", "dot", "manhattan", "chebyshev"):
        out = store.distances(q, nat.METRIC_CODES[metric])[0]
        ref = g[f"d{d}_qd_{metric}"]
        if metric in EXACT:
            np.testing.assert_array_equal(out, ref)
        else:
            assert_close(metric, out, ref, q, x)
    out = store.distances(q, nat.EUCLIDEAN_DIRECT)[0]
    np.testing.assert_array_equal(out.astype(np.float64), g[f"d{d}_scalar_euclidean"])
    store.close()


def test_range_search_golden(nat):
    g = load_golden("flat_small.npz")
    x, q = g["x"], g["q"]
    store = nat.RowStore(x.shape[1])
    store.append(x)
    dist, rows = store.range_search(q[2], 8.5, nat.EUCLIDEAN)
    full = R.query_distances(q[2], x, "euclidean")
    assert same_ids_or_near_ties(rows, g["range_rows"], lambda r: full[r], 8.5, rtol=RTOL)
    assert (np.diff(dist) >= 0).all() and (dist <= 8.5).all()
    store.close()


# ------------------------------------------------------------------ assignment / IVF

@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan", "dot"])
def test_ivf_golden(nat, metric):
    g = load_golden("ivf.npz")
    x, q, mask = g["x"], g["q"], g["mask"]
    cent = g[f"{metric}_centroids"]
    code = nat.METRIC_CODES[metric]
    # batch assignment (ivf.py:553-565): identity form for euclidean/cosine, scalar functions otherwise
    idx, _ = nat.assign(x, cent, code if metric in ("euclidean", "cosine") else {"manhattan": nat.MANHATTAN, "dot": nat.DOT}[metric])
    ref_assign = g[f"{metric}_assign_batch"]
    diff = np.flatnonzero(idx != ref_assign)
    if metric == "manhattan":
        assert len(diff) == 0
    else:
        # a differing assignment must be a rounding-level tie between two centroids
        for r in diff:
            d = R.rows_scalar_metric(x[r], cent, metric)
            assert abs(d[idx[r]] - d[ref_assign[r]]) <= 1e-5 * max(1.0, abs(d[ref_assign[r]]))
        assert len(diff) <= 2
    lists = nat.IvfLists(32, code)
    lists.set_centroids(cent)
    lists.build(x, ref_assign)
    exact = metric in ("euclidean", "manhattan")
    for npb in (1, 8, 32):
        dist, rows = lists.search(q, 10, npb)
        for i in range(len(q)):
            ref_rows, ref_dist = g[f"{metric}_np{npb}_rows"][i], g[f"{metric}_np{npb}_dist"][i]
            n = int((ref_rows >= 0).sum())
            assert (rows[i, n:] == -1).all()
            if exact:
                np.testing.assert_array_equal(rows[i, :n], ref_rows[:n])          # same rows, same order
                np.testing.assert_array_equal(dist[i, :n].astype(np.float64), ref_dist[:n])
            else:
                np.testing.assert_array_equal(np.sort(rows[i, :n]), np.sort(ref_rows[:n]))
                np.testing.assert_allclose(dist[i, :n], ref_dist[:n], rtol=RTOL, atol=2e-6)
    dist, rows = lists.search(q, 10, 8, nat.pack_mask(mask))
    for i in range(len(q)):
        ref_rows, ref_dist = g[f"{metric}_filter_rows"][i], g[f"{metric}_filter_dist"][i]
        n = int((ref_rows >= 0).sum())
        if exact:
            np.testing.assert_array_equal(rows[i, :n], ref_rows[:n])
            np.testing.assert_array_equal(dist[i, :n].astype(np.float64), ref_dist[:n])
        else:
            np.testing.assert_array_equal(np.sort(rows[i, :n]), np.sort(ref_rows[:n]))
    # single-vector assignment (ivf.py:546-551): scalar functions
    single_code = {"euclidean": nat.EUCLIDEAN_DIRECT, "cosine": nat.COSINE_PAIR, "manhattan": nat.MANHATTAN, "dot": nat.DOT}[metric]
    idx1, _ = nat.assign(x[:300], cent, single_code)
    ref1 = g[f"{metric}_assign_single"]
    if exact:
        np.testing.assert_array_equal(idx1, ref1)
    else:
        assert (idx1 != ref1).sum() <= 1
    # probe order
    probes, pd = lists.probe(q, 8)
    for i in range(len(q)):
        ref_p = R.ivf_probe_order(q[i], cent, metric, 8)
        if exact:
            np.testing.assert_array_equal(probes[i], ref_p)
    lists.close()


def test_ivf_large_nprobe_and_k(nat):
    rng = np.random.RandomState(8)
    x = rng.randn(3000, 16).astype(np.float32)
    q = rng.randn(7, 16).astype(np.float32)
    cent = R.kmeans_fit(x, 64, 5, seed=0)
    assign = R.ivf_assign_batch(x, cent, "euclidean")
    lists_ref = R.ivf_build_lists(assign, 64)
    lists = nat.IvfLists(16, nat.EUCLIDEAN)
    lists.set_centroids(cent)
    lists.build(x, assign)
    for npb, k in ((40, 10), (64, 50), (5, 100)):
        dist, rows = lists.search(q, k, npb)
        for i in range(len(q)):
            ref_rows, ref_dist = R.ivf_search(q[i], x, cent, lists_ref, k, npb, "euclidean")
            n = len(ref_rows)
            np.testing.assert_array_equal(rows[i, :n], ref_rows)
            np.testing.assert_array_equal(dist[i, :n].astype(np.float64), ref_dist)
            assert (rows[i, n:] == -1).all()
    assert lists.last_scan_rows() > 0
    lists.close()


def test_kmeans_accumulate(nat):
    rng = np.random.RandomState(4)
    x = rng.randn(5000, 24).astype(np.float32)
    a = rng.randint(0, 37, size=5000).astype(np.int32)
    sums, counts = nat.kmeans_accumulate(x, a, 40)
    ref = np.zeros((40, 24), np.float64)
    np.add.at(ref, a, x.astype(np.float64))
    np.testing.assert_allclose(sums, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(counts, np.bincount(a, minlength=40))


# ------------------------------------------------------------------ PQ

def test_pq_golden(nat):
    g = load_golden("pq.npz")
    x, q, mask, book = g["x"], g["q"], g["mask"], g["codebook"]
    np.testing.assert_array_equal(nat.pq_encode(x, book), g["codes"])              # pq.py:188-200, bit-exact
    pq = nat.PqCodes(32, 4, 256)
    pq.set_codebook(book)
    codes = pq.append_rows(x[:700])
    pq.append_codes(g["codes"][700:])
    np.testing.assert_array_equal(codes, g["codes"][:700])
    for i in range(len(q)):
        np.testing.assert_array_equal(pq.lut(q[i]), R.pq_lut(q[i], book))           # pq.py:270-283
    dist, rows = pq.search(q, 10)
    np.testing.assert_array_equal(rows, g["rows"])
    np.testing.assert_array_equal(dist, g["dist"])
    dist, rows = pq.search(q, 10, nat.pack_mask(mask))
    np.testing.assert_array_equal(rows, g["filter_rows"])
    np.testing.assert_array_equal(dist, g["filter_dist"])
    # full ordering (k = n): 38 selection passes
    dist, rows = pq.search(q[:2], len(x))
    for i in range(2):
        ref_rows, ref_dist = R.pq_search(q[i], book, g["codes"], len(x))
        np.testing.assert_array_equal(rows[i], ref_rows)
        np.testing.assert_array_equal(dist[i], ref_dist)
    pq.close()
    small = nat.PqCodes(32, 8, 256)
    small.set_codebook(g["small_codebook"])
    np.testing.assert_array_equal(small.append_rows(x[:300]), g["small_codes"])
    dist, rows = small.search(q[0], 10)
    np.testing.assert_array_equal(rows[0], g["small_rows"])
    np.testing.assert_array_equal(dist[0], g["small_dist"])
    small.close()


@pytest.mark.parametrize("m,dsub", [(16, 8), (8, 16), (3, 5), (2, 136)])
def test_pq_shapes(nat, m, dsub):
    rng = np.random.RandomState(m * dsub)
    d = m * dsub
    x = rng.randn(2500, d).astype(np.float32)
    q = rng.randn(3, d).astype(np.float32)
    book = rng.randn(m, 256, dsub).astype(np.float32)
    codes = R.pq_encode(x, book)
    np.testing.assert_array_equal(nat.pq_encode(x, book), codes)
    pq = nat.PqCodes(d, m, 256)
    pq.set_codebook(book)
    pq.append_codes(codes)
    mask = rng.rand(2500) < 0.3
    dist, rows = pq.search(q, 10, nat.pack_mask(mask))
    for i in range(3):
        ref_rows, ref_dist = R.pq_search(q[i], book, codes, 10, mask=mask)
        np.testing.assert_array_equal(rows[i], ref_rows)
        np.testing.assert_array_equal(dist[i], ref_dist)
    pq.close()


# ------------------------------------------------------------------ cross-shard merge

@pytest.mark.parametrize("shards", [2, 8])
def test_sharded_merge_equals_single_store(nat, shards):
    """G row shards searched separately on one GPU, merged by the same kernel an
    all-gather feeds on a multi-GPU node: result equals the unsharded search."""
    import torch
    from vectordb_amd.sharded import HipFlatBackend, ShardedFlat, shard_range
    rng = np.random.RandomState(shards)
    x = rng.randn(5003, 40).astype(np.float32)
    x[4000] = x[12]
    x[2500] = x[12]
    q = rng.randn(33, 40).astype(np.float32)
    q[0] = x[12]
    k = 10
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy(q).to(dev)
    for metric, code in (("manhattan", nat.MANHATTAN), ("cosine", nat.COSINE)):
        gd = torch.empty((shards, len(q), k), device=dev, dtype=torch.float32)
        gr = torch.empty((shards, len(q), k), device=dev, dtype=torch.int64)
        blocks = []
        for r in range(shards):
            lo, hi = shard_range(len(x), shards, r)
            be = HipFlatBackend(40, code)
            be.add_device(torch.from_numpy(x[lo:hi]).to(dev))
            d, rows = ShardedFlat(be, hi - lo, 0, 1).search(tq, k)         # local search, local row numbers
            # the packed block rank r of a `shards`-rank group would hand to the all-gather
            as_rank = ShardedFlat(be, len(x), r, shards)
            local, gathered, ld, lr, pgd, pgr, base, pod, porow = as_rank._buffers(len(q), k, dev)
            be.local_search(tq, k, ld, lr)
            blocks.append(local.clone())
            gd[r] = d
            gr[r] = rows + (rows >= 0).to(torch.int64) * lo              # what rank r contributes to the all-gather
            ref_r, ref_d = R.flat_search(q[1], x[lo:hi], k, metric)
            if metric == "manhattan":
                np.testing.assert_array_equal(rows[1].cpu().numpy(), ref_r)
            torch.cuda.synchronize()
            be.store.close()
        od = torch.empty((len(q), k), device=dev, dtype=torch.float32)
        orow = torch.empty((len(q), k), device=dev, dtype=torch.int64)
        nat.topk_merge_device(gd.data_ptr(), gr.data_ptr(), shards, len(q), k, od.data_ptr(), orow.data_ptr(),
                              stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        # the same merge over the packed blocks in place (rank stride > nq*k, local rows + per-rank base)
        gathered.copy_(torch.cat(blocks))
        HipFlatBackend.merge(be, pgd, pgr, k, pod, porow, base)
        torch.cuda.synchronize()
        assert torch.equal(pod, od) and torch.equal(porow, orow)
        od, orow = od.cpu().numpy(), orow.cpu().numpy()
        single = nat.RowStore(40)
        single.append(x)
        sd, sr = single.search(q, k, code)
        single.close()
        if metric == "manhattan":
            np.testing.assert_array_equal(orow, sr)
            np.testing.assert_array_equal(od, sd)
            assert list(orow[0][:3]) == [12, 2500, 4000]
        else:
            for i in range(len(q)):
                assert set(orow[i]) == set(sr[i])
            np.testing.assert_allclose(od, sd, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("shards", [2, 5])
def test_sharded_ivf_lists_equal_single_index(nat, shards):
    """Inverted lists spread over G logical shards (balanced greedy), every shard
    searched through the C ABI with all centroids but only its own lists, merged by
    (distance, visiting order): rows, order and distances equal the unsharded index."""
    import torch
    from vectordb_amd.sharded import HipIvfBackend, ShardedIVF, assign_lists_balanced, local_ivf_layout
    rng = np.random.RandomState(31 + shards)
    x = rng.randn(6000, 24).astype(np.float32)
    x[4100] = x[77]                        # equal distances
    x[5000] = x[77]
    cent = x[rng.choice(len(x), 40, replace=False)].copy()
    assign = R.ivf_assign_batch(x, cent, "euclidean").astype(np.int32)
    q = rng.randn(19, 24).astype(np.float32)
    q[0] = x[77]
    k, n_probe = 12, 9
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy(q).to(dev)
    owner = assign_lists_balanced(np.bincount(assign, minlength=len(cent)), shards)
    gd = torch.empty((shards, len(q), k), device=dev, dtype=torch.float32)
    gr = torch.empty((shards, len(q), k), device=dev, dtype=torch.int64)
    go = torch.empty((shards, len(q), k), device=dev, dtype=torch.int64)
    backends = []
    for r in range(shards):
        rows_g, list_rows_l, offs = local_ivf_layout(assign, len(cent), owner, r)
        be = HipIvfBackend(24, nat.EUCLIDEAN, cent, x[rows_g], list_rows_l, offs, rows_g)
        be.local_search(tq, k, n_probe, gd[r], gr[r], go[r])
        backends.append(be)
    od = torch.empty((len(q), k), device=dev, dtype=torch.float32)
    orow = torch.empty((len(q), k), device=dev, dtype=torch.int64)
    backends[0].merge(gd, go, gr, k, od, orow)
    torch.cuda.synchronize()
    single = nat.IvfLists(24, nat.EUCLIDEAN)
    single.set_centroids(cent)
    single.build(x, assign)
    sd, sr = single.search(q, k, n_probe)
    np.testing.assert_array_equal(orow.cpu().numpy(), sr)
    np.testing.assert_array_equal(od.cpu().numpy(), sd)
    members = [np.flatnonzero(assign == l).astype(np.int64) for l in range(len(cent))]
    ref_r, ref_d = R.ivf_search(q[0], x, cent, members, k, n_probe, "euclidean")
    np.testing.assert_array_equal(sr[0], ref_r)
    # the packed form of one all-gather: every shard's (distances | visiting orders | rows) block side by side,
    # merged where the blocks lie
    sh = ShardedIVF(backends[0], 0, shards)
    _local, gathered, _ld, _lo, _lr, pd, po, pr, pod, porow = sh._buffers(len(q), k, dev)
    gathered.fill_(0xEE)
    for r, be in enumerate(backends):
        be.local_search(tq, k, n_probe, pd[r], pr[r], po[r])
    backends[0].merge(pd, po, pr, k, pod, porow)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(porow.cpu().numpy(), sr)
    np.testing.assert_array_equal(pod.cpu().numpy(), sd)
    assert sh._buffers(len(q), k, dev)[1] is gathered                     # made once, reused
    # ShardedIVF with one shard is the plain index
    d1, r1 = ShardedIVF(backends[0], 0, 1).search(tq, k, n_probe)
    assert d1.shape == (len(q), k)
    single.close()
    for be in backends:
        be.lists.close()


def test_sharded_pq_blocks_equal_single_index(nat):
    """PQ codes in contiguous row blocks, one block per logical shard, merged by (distance, global row)."""
    import torch
    from vectordb_amd.sharded import HipPqBackend, ShardedFlat, shard_range
    rng = np.random.RandomState(8)
    m, d, n, k, shards = 8, 32, 9001, 10, 3
    book = rng.randn(m, 256, d // m).astype(np.float32)
    codes = rng.randint(0, 256, size=(n, m)).astype(np.uint8)
    codes[7000] = codes[5]                 # equal distances across shards
    q = rng.randn(6, d).astype(np.float32)
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy(q).to(dev)
    gd = torch.empty((shards, len(q), k), device=dev, dtype=torch.float32)
    gr = torch.empty((shards, len(q), k), device=dev, dtype=torch.int64)
    bes = []
    for r in range(shards):
        lo, hi = shard_range(n, shards, r)
        be = HipPqBackend(d, m, 256, book)
        be.add_device(torch.from_numpy(codes[lo:hi]).to(dev))
        dl, rl = ShardedFlat(be, hi - lo, 0, 1).search(tq, k)
        gd[r] = dl
        gr[r] = rl + (rl >= 0).to(torch.int64) * lo
        bes.append(be)
    od = torch.empty((len(q), k), device=dev, dtype=torch.float32)
    orow = torch.empty((len(q), k), device=dev, dtype=torch.int64)
    bes[0].merge(gd, gr, k, od, orow)
    torch.cuda.synchronize()
    for i in range(len(q)):
        ref_r, ref_d = R.pq_search(q[i], book, codes, k)
        np.testing.assert_array_equal(orow[i].cpu().numpy(), ref_r)
        np.testing.assert_array_equal(od[i].cpu().numpy(), ref_d)
    for be in bes:
        be.codes.close()


# ------------------------------------------------------------------ bf16 storage
def _bf16_round(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("metric", ["euclidean", "euclidean_squared", "cosine", "dot"])
def test_flat_bf16_storage(nat, metric):
    """bf16 storage = the fp32 path on inputs rounded to bf16 (products of bf16 values are exact
    in fp32, accumulation is fp32): against the oracle on ROUNDED inputs the usual 1e-4 bar holds.
    Against the unrounded fp32 reference the stated tolerance is 2e-2 relative on the distance
    (bf16 keeps 8 significant bits) and recall@10 is reported (>= 0.8 on Gaussian data here)."""
    rng = np.random.RandomState(12)
    n, d, nq, k = 20011, 100, 70, 10
    x = rng.randn(n, d).astype(np.float32)
    q = rng.randn(nq, d).astype(np.float32)
    code = {"euclidean": nat.EUCLIDEAN, "euclidean_squared": nat.EUCLIDEAN_SQ, "cosine": nat.COSINE, "dot": nat.DOT}[metric]
    store = nat.RowStore(d, storage="bf16")
    store.append(x[:15000])
    store.append(x[15000:])
    assert len(store) == n
    xr, qr = _bf16_round(x), _bf16_round(q)
    sq, nrm = store.norms()
    assert np.array_equal(sq, np.sum(xr * xr, axis=1))              # norm terms describe the rounded rows
    dist, rows = store.search(q, k, code)
    recall = 0.0
    for i in range(nq):
        ref_all = R.query_distances(qr[i], xr, metric)
        ref_rows = R.topk_stable(ref_all, k)
        tol = ids_atol(metric, qr[i], xr, float(ref_all[ref_rows[-1]]))
        assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: ref_all[r], float(ref_all[ref_rows[-1]]), rtol=RTOL, atol=tol)
        assert_close(metric, dist[i], ref_all[rows[i]], qr[i], xr)
        # against the unrounded reference
        full = R.query_distances(q[i], x, metric)
        recall += len(set(rows[i].tolist()) & set(R.topk_stable(full, k).tolist())) / k
        scale = np.abs(full[rows[i]]) + (np.sqrt(np.sum(q[i] ** 2) * np.max(sq)) if metric != "cosine" else 1.0) * 1e-2
        assert np.all(np.abs(dist[i] - full[rows[i]]) <= 2e-2 * scale)
    recall /= nq
    print(f"bf16 {metric}: recall@10 vs fp32 reference = {recall:.3f}")
    assert recall >= 0.8
    # one query, a filter, k beyond one selection pass, a full distance row
    keep = (np.arange(n) % 3) == 1
    d1, r1 = store.search(q[:1], 40, code, mask_words=nat.pack_mask(keep))
    ref_all = R.query_distances(qr[0], xr, metric)
    kept = np.flatnonzero(keep)
    ref_rows = kept[R.topk_stable(ref_all[kept], 40)]
    assert np.all(keep[r1[0]])
    assert same_ids_or_near_ties(r1[0], ref_rows, lambda r: ref_all[r], float(ref_all[ref_rows[-1]]), rtol=RTOL,
                                 atol=ids_atol(metric, qr[0], xr, float(ref_all[ref_rows[-1]])))
    full_rows = store.distances(q[:5], code)
    for i in range(5):
        assert_close(metric, full_rows[i], R.query_distances(qr[i], xr, metric), qr[i], xr)
    with pytest.raises(ValueError):
        store.search(q[:1], 5, nat.MANHATTAN)
    store.close()


# ------------------------------------------------------------------ IVF-PQ (HybridIndex)
def test_ivfpq_golden(nat):
    """The reference's own HybridIndex outputs: list and code of every row, searches at two probe
    counts with and without a filter, bit-exact distances and order; coarse assignment goes through
    BLAS in the reference, so a row may sit in another list only on a rounding-level tie."""
    g = load_golden("hybrid.npz")
    x, q, cent, book = g["x"], g["q"], g["centroids"], g["codebook"]
    h = nat.IvfPqLists(32, 4, 256)
    h.set_quantizers(cent, book)
    lists, codes = h.encode(x)
    moved = np.flatnonzero(lists != g["lists"])
    assert len(moved) <= 2
    for r in moved:
        d = np.sum((x[r] - cent) ** 2, axis=1)
        assert abs(d[lists[r]] - d[g["lists"][r]]) <= 1e-4 * d[g["lists"][r]]
    same = lists == g["lists"]
    assert np.array_equal(codes[same], g["codes"][same])           # direct-form residual codes: exact
    n_lists = len(cent)
    order = np.argsort(g["lists"], kind="stable")
    offs = np.searchsorted(g["lists"][order], np.arange(n_lists + 1))
    h.build(g["codes"], order, offs)
    words = nat.pack_mask(g["mask"])
    for tag, n_probe in (("p4", 4), ("p16", 16)):
        dist, rows = h.search(q, 10, n_probe)
        np.testing.assert_array_equal(rows, g[f"rows_{tag}"])
        np.testing.assert_array_equal(dist, g[f"dist_{tag}"])
        dist, rows = h.search(q, 10, n_probe, mask_words=words)
        np.testing.assert_array_equal(rows, g[f"filter_rows_{tag}"])
        np.testing.assert_array_equal(dist, g[f"filter_dist_{tag}"])
    # removal keeps list order (hybrid.py:405-409): rebuild the lists without the removed rows
    keep = np.ones(len(x), dtype=bool)
    keep[g["removed"]] = False
    order2 = order[keep[order]]
    offs2 = np.searchsorted(g["lists"][order2], np.arange(n_lists + 1))
    compact = np.full(len(x), -1, dtype=np.int64)
    compact[np.flatnonzero(keep)] = np.arange(int(keep.sum()))
    h.build(g["codes"][keep], compact[order2], offs2)
    dist, rows = h.search(q[:1], 10, 16)
    np.testing.assert_array_equal(np.flatnonzero(keep)[rows[0]], g["rows_after_remove"])
    np.testing.assert_array_equal(dist[0], g["dist_after_remove"])
    h.close()


def test_ivfpq_vs_oracle_batch(nat):
    """A batch large enough for the matrix-core coarse ranking, k beyond one selection pass, m = 8."""
    rng = np.random.RandomState(21)
    n, d, m, n_lists = 20000, 64, 8, 64
    x = rng.randn(n, d).astype(np.float32)
    q = rng.randn(40, d).astype(np.float32)
    cent = x[rng.choice(n, n_lists, replace=False)].copy()
    book = (0.5 * rng.randn(m, 256, d // m)).astype(np.float32)
    h = nat.IvfPqLists(d, m, 256)
    h.set_quantizers(cent, book)
    lists, codes = h.encode(x)
    ref_lists, ref_codes = R.hybrid_encode(x, cent, book)
    same = lists == ref_lists
    assert same.mean() > 0.999
    assert np.array_equal(codes[same], ref_codes[same])
    order = np.argsort(lists, kind="stable")
    offs = np.searchsorted(lists[order], np.arange(n_lists + 1))
    h.build(codes, order, offs)
    members = [order[offs[l]:offs[l + 1]] for l in range(n_lists)]
    dist, rows = h.search(q, 50, 6)
    for i in range(len(q)):
        ref_rows, ref_dist = R.hybrid_search(q[i], cent, book, codes, members, 50, 6)
        # the coarse ranking is a BLAS product in the reference: compare only when the probed sets agree
        cs = np.sum(cent ** 2, axis=1)
        coarse = np.sort(np.sum(q[i] ** 2) + cs - 2 * (q[i] @ cent.T))
        if coarse[6] - coarse[5] > 1e-3:
            np.testing.assert_array_equal(rows[i], ref_rows)
            np.testing.assert_array_equal(dist[i], ref_dist)
    # 200 queries: the lists ranked from the whole distance matrix, 1200 (query, list) units scanned with the codebooks
    # in LDS and the candidates ranked by one block per query; 20 queries: one block per unit, then the slot selection
    q2 = rng.randn(200, d).astype(np.float32)
    dist2, rows2 = h.search(q2, 10, 6)
    dist3, rows3 = h.search(q2[:20], 10, 6)
    np.testing.assert_array_equal(rows2[:20], rows3)
    np.testing.assert_array_equal(dist2[:20].view(np.uint32), dist3.view(np.uint32))
    for i in range(150, 170):
        ref_rows, ref_dist = R.hybrid_search(q2[i], cent, book, codes, members, 10, 6)
        coarse = np.sort(np.sum(q2[i] ** 2) + cs - 2 * (q2[i] @ cent.T))
        if coarse[6] - coarse[5] > 1e-3:
            np.testing.assert_array_equal(rows2[i], ref_rows)
            np.testing.assert_array_equal(dist2[i], ref_dist)
    h.close()


@pytest.mark.parametrize("m,dsub", [(4, 4), (4, 8), (4, 16), (8, 4), (8, 8), (8, 16), (16, 4), (16, 8)])
def test_ivfpq_two_table_scan_every_shape(nat, m, dsub):
    """The batch scan with the codebooks and two tables in LDS, for every (sub-spaces, sub-space width) it is compiled
    for: 1200 (query, list) units over lists of a few thousand rows, some past the 4096 whose codes are read a unit
    ahead, one list empty; distances and rows equal to the oracle's wherever the probed lists are not a near-tie."""
    rng = np.random.RandomState(100 * m + dsub)
    d = m * dsub
    n, n_lists, n_probe, k = 30000, 9, 4, 10
    x = rng.randn(n, d).astype(np.float32)
    cent = x[rng.choice(n, n_lists, replace=False)].copy()
    cent[8] = 50.0                                   # far from every row: an empty list
    cent[0] *= 0.25                                  # near the origin: a long list
    book = (0.5 * rng.randn(m, 256, dsub)).astype(np.float32)
    h = nat.IvfPqLists(d, m, 256)
    h.set_quantizers(cent, book)
    lists, codes = h.encode(x)
    order = np.argsort(lists, kind="stable")
    offs = np.searchsorted(lists[order], np.arange(n_lists + 1))
    sizes = np.diff(offs)
    assert sizes.max() > 4096 and sizes.min() == 0
    h.build(codes, order, offs)
    members = [order[offs[l]:offs[l + 1]] for l in range(n_lists)]
    q = rng.randn(300, d).astype(np.float32)
    dist, rows = h.search(q, k, n_probe)
    cs = np.sum(cent ** 2, axis=1)
    checked = 0
    for i in list(range(0, 300, 25)) + [299]:
        ref_rows, ref_dist = R.hybrid_search(q[i], cent, book, codes, members, k, n_probe)
        coarse = np.sort(np.sum(q[i] ** 2) + cs - 2 * (q[i] @ cent.T))
        if coarse[n_probe] - coarse[n_probe - 1] > 1e-3:
            np.testing.assert_array_equal(rows[i], ref_rows)
            np.testing.assert_array_equal(dist[i], ref_dist)
            checked += 1
    assert checked >= 8
    h.close()


def test_sharded_merge_with_short_shards(nat):
    """Shards holding fewer rows than k: their lists end in empty entries (row -1), which the
    packed merge skips before it adds the shard's first global row."""
    import torch
    from vectordb_amd.sharded import HipFlatBackend, ShardedFlat, shard_range
    rng = np.random.RandomState(3)
    shards, k, d = 8, 10, 16
    x = rng.randn(25, d).astype(np.float32)
    x[24] = x[2]                                    # a tie between the first and the last shard
    q = rng.randn(5, d).astype(np.float32)
    q[0] = x[2]
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy(q).to(dev)
    blocks = []
    for r in range(shards):
        lo, hi = shard_range(len(x), shards, r)
        be = HipFlatBackend(d, nat.MANHATTAN)
        be.add_device(torch.from_numpy(x[lo:hi]).to(dev))
        as_rank = ShardedFlat(be, len(x), r, shards)
        local, gathered, ld, lr, gd, gr, base, od, orow = as_rank._buffers(len(q), k, dev)
        be.local_search(tq, k, ld, lr)
        torch.cuda.synchronize()
        assert int((lr >= 0).sum()) == len(q) * (hi - lo)
        blocks.append(local.clone())
        be.store.close()
    gathered.copy_(torch.cat(blocks))
    HipFlatBackend.merge(be, gd, gr, k, od, orow, base)
    torch.cuda.synchronize()
    single = nat.RowStore(d)
    single.append(x)
    sd, sr = single.search(q, k, nat.MANHATTAN)
    single.close()
    np.testing.assert_array_equal(orow.cpu().numpy(), sr)
    np.testing.assert_array_equal(od.cpu().numpy(), sd)
    assert list(orow[0, :2].cpu().numpy()) == [2, 24]
