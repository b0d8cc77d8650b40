"""The candidate-queue form of the batched Flat search (vdb_flat_search on row sets large
enough for a sampled bound): parity with the CPU oracle to the same bars as the list form
(distances 1e-4 relative, ids equal up to rows tied at the k-th place), on both kernels
(row pitch a multiple of 128 bytes or not), with a filter mask, ragged ends, k = 1 and
k = 32, bf16 storage, and on data built so that the queues overflow and the list kernel
has to repeat the search."""

import numpy as np
import pytest

from oracle import ref_numpy as R
from conftest import same_ids_or_near_ties
from test_gpu_cabi import check_search, assert_close, ids_atol, RTOL, _bf16_round

pytestmark = pytest.mark.gpu

N_MIN = 512 * 128           # smallest row count that gets a sample pass


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def search_and_stats(nat, x, q, k, metric, mask=None, storage=None):
    store = nat.RowStore(x.shape[1], storage=storage or "f32")
    store.append(x)
    dist, rows = store.search(q, k, nat.METRIC_CODES[metric], nat.pack_mask(mask))
    stats = store.search_stats()
    store.close()
    return dist, rows, stats


@pytest.mark.parametrize("metric", ["euclidean", "euclidean_squared", "cosine", "dot"])
@pytest.mark.parametrize("d,n_extra", [(32, 0), (32, 77), (100, 5), (48, 128)])
def test_queue_search_matches_oracle(nat, metric, d, n_extra):
    """d = 32: 128-byte pitch (branch-free kernel); d = 100, 48: other pitches (generic kernel)."""
    rng = np.random.default_rng(d * 7 + n_extra)
    n = N_MIN + n_extra
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((130, d), dtype=np.float32)         # two query tiles, the second ragged
    _, _, stats = search_and_stats(nat, x, q[:5], 10, metric)
    assert stats == (True, False)
    check_search(nat, x, q[:40], 10, metric)
    dist, rows = check_search(nat, x, q[:3], 1, metric)
    check_search(nat, x, q[100:130], 32, metric)                # rows of the second tile only through a full batch:
    d_all, r_all, _ = search_and_stats(nat, x, q, 10, metric)
    d_part, r_part, _ = search_and_stats(nat, x, q[124:130], 10, metric)     # the same dot products in another tiling
    np.testing.assert_array_equal(r_all[124:130], r_part)
    np.testing.assert_array_equal(d_all[124:130], d_part)


def test_queue_search_with_mask_and_few_survivors(nat):
    rng = np.random.default_rng(3)
    n, d = N_MIN + 1000, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((16, d), dtype=np.float32)
    for keep in (0.5, 0.01, 20.0 / n):
        mask = rng.random(n) < keep
        mask[-3:] = True                                         # rows of the ragged last tile stay eligible
        check_search(nat, x, q, 10, "cosine", mask=mask)
        check_search(nat, x, q[:4], 10, "euclidean", mask=mask)
    mask = np.zeros(n, dtype=bool)
    mask[[5, n - 1]] = True                                      # fewer passing rows than k
    dist, rows, stats = search_and_stats(nat, x, q[:4], 10, "dot", mask=mask)
    assert stats[0]
    assert (np.sort(rows[:, :2], axis=1) == [5, n - 1]).all() and (rows[:, 2:] == -1).all()


def test_queue_ties_go_to_lowest_row(nat):
    """Duplicated rows: equal distances come out in ascending row order (the stable order of
    core/collection.py:757 and the argsort tie rule of index/flat.py:354-362)."""
    rng = np.random.default_rng(11)
    n, d = N_MIN, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((4, d), dtype=np.float32)
    for i in range(4):
        near = (q[i] + 0.01 * rng.standard_normal(d)).astype(np.float32)
        x[[1000 + i, 20000 + i, 50000 + i, n - 1 - i]] = near   # four copies, four different slabs
    dist, rows, stats = search_and_stats(nat, x, q, 6, "euclidean")
    assert stats == (True, False)
    for i in range(4):
        assert list(rows[i, :4]) == [1000 + i, 20000 + i, 50000 + i, n - 1 - i]
        assert dist[i, 0] == dist[i, 1] == dist[i, 2] == dist[i, 3]


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "dot"])
def test_queue_overflow_falls_back_to_lists(nat, metric):
    """The sampled tiles (every 16th) hold only far rows, all others hold rows close to the
    queries: the sampled bound lets almost every row through, the queues overflow, the flag
    is raised on the device and the list kernel repeats the search.  Results stay exact."""
    rng = np.random.default_rng(5)
    n, d = N_MIN, 32
    centre = rng.standard_normal(d).astype(np.float32)
    x = (centre + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    tiles = np.arange(n) // 128
    far = tiles % 16 == 0
    x[far] = (-centre + 3.0 * rng.standard_normal((int(far.sum()), d))).astype(np.float32)
    q = (centre + 0.05 * rng.standard_normal((12, d))).astype(np.float32)
    dist, rows, stats = search_and_stats(nat, x, q, 10, metric)
    assert stats == (True, True)
    for i in range(len(q)):
        ref_rows, ref_dist = R.flat_search(q[i], x, 10, metric)
        full = R.query_distances(q[i], x, metric)
        atol = ids_atol(metric, q[i], x, float(ref_dist[-1]))
        assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
        assert_close(metric, dist[i], ref_dist, q[i], x)
    # the next search on the same handle starts from a clean flag
    store = nat.RowStore(d)
    store.append(x)
    store.search(q, 10, nat.METRIC_CODES[metric])
    assert store.search_stats() == (True, True)
    y = rng.standard_normal((n, d), dtype=np.float32)
    store.truncate(0)
    store.append(y)
    store.search(q, 10, nat.METRIC_CODES[metric])
    assert store.search_stats() == (True, False)
    store.close()


def test_one_overflowing_query_is_searched_again_alone(nat):
    """4000 rows sit next to one point, none of them in a sampled tile: the query at that point gets a bound from
    ordinary rows, its queues overflow, and it alone goes through the list kernel again.  The other 1023 queries keep
    the results they have without it (bit for bit), the unlucky one matches the oracle, and the batch costs little
    more than a clean one (a repeat of the whole batch through the list kernel costs more than twice)."""
    import time
    rng = np.random.default_rng(21)
    n, d, nq, k = 1024 * 1024, 64, 1024, 10
    x = rng.standard_normal((n, d), dtype=np.float32)
    point = rng.standard_normal(d).astype(np.float32)
    tiles = np.arange(n) // 128
    free = np.flatnonzero(tiles % 4 != 0)                      # whatever the sample stride (a multiple of 4 here), not sampled
    spot = rng.choice(free, 4000, replace=False)
    x[spot] = (point + 0.05 * rng.standard_normal((len(spot), d))).astype(np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    bad = 517
    q_bad = q.copy()
    q_bad[bad] = point
    store = nat.RowStore(d)
    store.append(x)
    for mode in (None, "bf16"):
        store.set_prefilter(mode)
        code = nat.METRIC_CODES["euclidean"]
        d_clean, r_clean = store.search(q, k, code)
        assert store.search_stats() == (True, False)
        d_bad, r_bad = store.search(q_bad, k, code)
        assert store.search_stats() == (True, True)
        others = np.arange(nq) != bad
        np.testing.assert_array_equal(r_bad[others], r_clean[others])
        np.testing.assert_array_equal(d_bad[others].view(np.uint32), d_clean[others].view(np.uint32))
        ref_rows, ref_dist = R.flat_search(point, x, k, "euclidean")
        full = R.query_distances(point, x, "euclidean")
        assert same_ids_or_near_ties(r_bad[bad], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL,
                                     atol=ids_atol("euclidean", point, x, float(ref_dist[-1])))
        assert_close("euclidean", d_bad[bad], ref_dist, point, x)
        if mode is None:
            def best_of(queries):
                t = []
                for _ in range(7):
                    t0 = time.perf_counter()
                    store.search(queries, k, code)
                    t.append(time.perf_counter() - t0)
                return min(t)
            t_clean, t_bad = best_of(q), best_of(q_bad)
            print(f"clean batch {t_clean * 1e3:.3f} ms, with one overflowing query {t_bad * 1e3:.3f} ms ({t_bad / t_clean:.2f}x)")
            assert t_bad < 1.35 * t_clean
    store.close()


def test_queue_search_bf16_storage(nat):
    """bf16 rows: the fp32 bars hold against the oracle on the rounded inputs."""
    rng = np.random.default_rng(9)
    n, d = N_MIN + 50, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((20, d), dtype=np.float32)
    xr, qr = _bf16_round(x), _bf16_round(q)
    for metric in ("cosine", "dot", "euclidean"):
        dist, rows, stats = search_and_stats(nat, x, q, 10, metric, storage="bf16")
        assert stats == (True, False)
        for i in range(len(q)):
            ref_rows, ref_dist = R.flat_search(qr[i], xr, 10, metric)
            full = R.query_distances(qr[i], xr, metric)
            atol = ids_atol(metric, qr[i], xr, float(ref_dist[-1]))
            assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
            assert_close(metric, dist[i], ref_dist, qr[i], xr)


def test_queue_search_more_queries_than_one_block(nat):
    """1030 queries: the library walks them in blocks of 1024 (bound pass per block, then queue pass
    per block, one overflow flag for all).  The ragged second block returns what a small batch of
    the same queries returns, bit for bit, and the first block matches the oracle."""
    rng = np.random.default_rng(21)
    n, d = N_MIN + 300, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((1030, d), dtype=np.float32)
    for metric in ("cosine", "euclidean"):
        d_all, r_all, stats = search_and_stats(nat, x, q, 10, metric)
        assert stats == (True, False)
        d_tail, r_tail, _ = search_and_stats(nat, x, q[1020:1030], 10, metric)
        np.testing.assert_array_equal(r_all[1020:1030], r_tail)
        np.testing.assert_array_equal(d_all[1020:1030], d_tail)
        d_mid, r_mid, _ = search_and_stats(nat, x, q[500:520], 10, metric)
        np.testing.assert_array_equal(r_all[500:520], r_mid)
        np.testing.assert_array_equal(d_all[500:520], d_mid)
        for i in (0, 777, 1023, 1024, 1029):
            ref_rows, ref_dist = R.flat_search(q[i], x, 10, metric)
            full = R.query_distances(q[i], x, metric)
            atol = ids_atol(metric, q[i], x, float(ref_dist[-1]))
            assert same_ids_or_near_ties(r_all[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
            assert_close(metric, d_all[i], ref_dist, q[i], x)


def test_queue_search_bf16_two_query_tiles(nat):
    """bf16 rows, 300 queries: the 256 x 256 kernel with a ragged second query tile and a ragged
    last row tile; queries 256.. return what a batch of their own returns."""
    rng = np.random.default_rng(22)
    n, d = N_MIN + 131, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((300, d), dtype=np.float32)
    xr, qr = _bf16_round(x), _bf16_round(q)
    for metric in ("dot", "cosine"):
        d_all, r_all, stats = search_and_stats(nat, x, q, 10, metric, storage="bf16")
        assert stats == (True, False)
        d_tail, r_tail, _ = search_and_stats(nat, x, q[256:300], 10, metric, storage="bf16")
        np.testing.assert_array_equal(r_all[256:300], r_tail)
        np.testing.assert_array_equal(d_all[256:300], d_tail)
        for i in (0, 255, 256, 299):
            ref_rows, ref_dist = R.flat_search(qr[i], xr, 10, metric)
            full = R.query_distances(qr[i], xr, metric)
            atol = ids_atol(metric, qr[i], xr, float(ref_dist[-1]))
            assert same_ids_or_near_ties(r_all[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
            assert_close(metric, d_all[i], ref_dist, qr[i], xr)


@pytest.mark.parametrize("metric", ["cosine", "euclidean", "dot"])
def test_queue_search_zero_and_tiny_vectors(nat, metric):
    """Zero and near-zero rows and queries in a row set large enough for the queue path: the
    norm guards of distance/metrics.py:472-476 (norms under 1e-8 count as 1) decide the cosine
    distances there, and the conservative filter must not lose such rows."""
    rng = np.random.default_rng(31)
    n, d = N_MIN + 17, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[[3, 40000, n - 1]] = 0.0
    x[[7, 50001]] = (1e-10 * rng.standard_normal((2, d))).astype(np.float32)
    x[[9, 60000]] *= 1e3
    q = rng.standard_normal((8, d), dtype=np.float32)
    q[1] = 0.0
    q[2] = (1e-10 * rng.standard_normal(d)).astype(np.float32)
    q[3] *= 1e3
    q[4] = x[9]
    dist, rows, stats = search_and_stats(nat, x, q, 10, metric)
    assert stats[0]
    for i in range(len(q)):
        ref_rows, ref_dist = R.flat_search(q[i], x, 10, metric)
        full = R.query_distances(q[i], x, metric)
        atol = ids_atol(metric, q[i], x, float(ref_dist[-1]))
        assert same_ids_or_near_ties(rows[i], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol), (metric, i)
        assert_close(metric, dist[i], ref_dist, q[i], x)
