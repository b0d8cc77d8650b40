"""Batched IVF searches through the matrix-core scan (vdb_ivf_set_prefilter) against the oracle's scalar scan.

The reference visits the probed lists in ranking order and computes sqrt(sum((q - c)^2)) per member
(index/ivf.py:611-632, distance/metrics.py:43).  The matrix cores only pick the pairs that get that computation, so
every case here demands the reference's rows in the reference's order with identical distance bits: dimensions that
are and are not multiples of the bf16 column steps, rows longer than one column chunk, duplicated rows (ties in
visiting order), filters, lists too small to give a bound, a queue overflow (the streaming scan repeats the batch),
row magnitudes from 1e-12 to 1e12, and the bf16 copy kept in step with appends, list moves and removals."""

import numpy as np
import pytest

from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def build(nat, x, cent, metric="euclidean"):
    assign = R.ivf_assign_batch(x, cent, "euclidean").astype(np.int32)
    lists = nat.IvfLists(x.shape[1], nat.EUCLIDEAN if metric == "euclidean" else nat.EUCLIDEAN_SQ)
    lists.set_centroids(cent)
    lists.build(x, assign)
    members = [np.flatnonzero(assign == l).astype(np.int64) for l in range(len(cent))]
    return lists, members


def check(nat, lists, x, cent, members, q, k, n_probe, mask=None, metric="euclidean", expect_overflow=False):
    words = nat.pack_mask(mask) if mask is not None else None
    dist, rows = lists.search(q, k, n_probe, words)
    used, over = lists.search_stats()
    assert used, "the batch did not take the matrix-core scan"
    assert over == expect_overflow
    for i in range(len(q)):
        ref_rows, ref_dist = R.ivf_search(q[i], x, cent, members, k, n_probe, metric, mask)
        n = len(ref_rows)
        if metric == "euclidean":
            np.testing.assert_array_equal(rows[i, :n], ref_rows, err_msg=f"query {i}")
            np.testing.assert_array_equal(dist[i, :n].astype(np.float64), ref_dist.astype(np.float32).astype(np.float64), err_msg=f"query {i}")
        else:
            # the reference's squared form goes through BLAS (np.dot(diff, diff), distance/metrics.py:68-69): not
            # reproducible to the bit, so the values are compared to 1e-5 here and to the bit with the streaming scan below
            np.testing.assert_allclose(dist[i, :n], ref_dist, rtol=1e-5, err_msg=f"query {i}")
        assert (rows[i, n:] == -1).all() and np.isinf(dist[i, n:]).all()
    # the streaming scan of the same handle returns the same bits
    lists.set_prefilter(None)
    d2, r2 = lists.search(q, k, n_probe, words)
    assert lists.search_stats() == (False, False)
    lists.set_prefilter("bf16")
    np.testing.assert_array_equal(rows, r2)
    np.testing.assert_array_equal(dist.view(np.uint32), d2.view(np.uint32))


@pytest.mark.parametrize("d", [8, 20, 32, 64, 100, 128, 200, 384])
def test_matches_scalar_scan(nat, d):
    rng = np.random.RandomState(100 + d)
    n = 20000
    x = rng.randn(n, d).astype(np.float32)
    x[5000] = x[17]; x[12000] = x[17]; x[19999] = x[17]          # equal distances: visiting order decides
    cent = x[rng.choice(n, 64, replace=False)].copy()
    q = rng.randn(70, d).astype(np.float32)
    q[0] = x[17]
    q[1] = x[17] + 1e-4
    lists, members = build(nat, x, cent)
    lists.set_prefilter("bf16")
    check(nat, lists, x, cent, members, q, 10, 8)
    check(nat, lists, x, cent, members, q[:3], 32, 64)           # every list probed, the largest k of the path
    check(nat, lists, x, cent, members, q[:40], 1, 1)
    lists.close()


def test_squared_metric_filters_and_small_lists(nat):
    rng = np.random.RandomState(7)
    n, d = 30000, 48
    x = rng.randn(n, d).astype(np.float32)
    # a few large lists and many tiny ones: queries near tiny lists get their bound from several lists or none
    cent = np.concatenate([x[rng.choice(n, 6, replace=False)], 6.0 + 0.01 * rng.randn(60, d).astype(np.float32)]).astype(np.float32)
    x[:300] = cent[6 + rng.randint(0, 60, 300)] + 0.001 * rng.randn(300, d).astype(np.float32)
    q = np.concatenate([rng.randn(30, d), cent[6:26] + 0.01 * rng.randn(20, d)]).astype(np.float32)
    for metric in ("euclidean", "euclidean_squared"):
        lists, members = build(nat, x, cent, metric)
        lists.set_prefilter("bf16")
        check(nat, lists, x, cent, members, q, 10, 12, metric=metric)
        for keep in (0.5, 0.02, 0.0005):
            mask = rng.rand(n) < keep
            check(nat, lists, x, cent, members, q, 10, 12, mask=mask, metric=metric)
        check(nat, lists, x, cent, members, q, 10, 66, mask=np.zeros(n, dtype=bool), metric=metric)
        lists.close()


def test_queue_overflow_falls_back_to_the_streaming_scan(nat):
    """A filter that leaves a query's nearest list without k rows gives that query no bound: all of its other
    candidates pass, its queue fills, and the streaming scan repeats the batch with the same results."""
    rng = np.random.RandomState(8)
    n, d = 60000, 32
    x = rng.randn(n, d).astype(np.float32)
    cent = x[rng.choice(n, 8, replace=False)].copy()
    q = rng.randn(12, d).astype(np.float32)
    lists, members = build(nat, x, cent)
    lists.set_prefilter("bf16")
    probes, _ = lists.probe(q[:1], 1)
    mask = np.ones(n, dtype=bool)
    mask[members[int(probes[0, 0])]] = False
    assert mask.sum() > 8 * 4096
    check(nat, lists, x, cent, members, q, 10, 8, mask=mask, expect_overflow=True)
    check(nat, lists, x, cent, members, q, 10, 8)                 # and the flag does not stick
    lists.close()


def test_magnitudes_and_degenerate_rows(nat):
    rng = np.random.RandomState(9)
    n, d = 12000, 40
    for scale in (1e-12, 1e-3, 1.0, 1e6, 1e12):
        x = (rng.randn(n, d) * scale).astype(np.float32)
        x[100:140] = 0.0                                          # zero rows
        x[200:260] = x[199]                                       # sixty copies of one row
        cent = x[rng.choice(n, 24, replace=False)].copy()
        q = (rng.randn(20, d) * scale).astype(np.float32)
        q[0] = 0.0
        q[1] = x[199]
        lists, members = build(nat, x, cent)
        lists.set_prefilter("bf16")
        check(nat, lists, x, cent, members, q, 10, 6)
        lists.close()
    # norms whose squares leave float32: the flag sends the batch to the streaming scan
    x = (rng.randn(n, d) * 5e17).astype(np.float32)
    cent = x[rng.choice(n, 24, replace=False)].copy()
    q = (rng.randn(6, d) * 5e17).astype(np.float32)
    lists, members = build(nat, x, cent)
    lists.set_prefilter("bf16")
    check(nat, lists, x, cent, members, q, 10, 6, expect_overflow=True)
    lists.close()


def test_list_ranking_with_over_a_thousand_equal_centroids(nat):
    """The lists to probe are ranked behind bf16 keys too.  1300 copies of one centroid leave more candidates within a
    query's bound than the ranking block holds, so it computes every centroid's distance instead; the ranking among
    equal distances goes by centroid number, as in the scan without the matrix cores."""
    rng = np.random.RandomState(12)
    n, d = 16000, 32
    x = rng.randn(n, d).astype(np.float32)
    cent = np.concatenate([np.repeat(x[:1], 1300, axis=0), x[rng.choice(n, 200, replace=False)]]).astype(np.float32)
    q = np.concatenate([x[:1] + 0.01 * rng.randn(4, d), rng.randn(12, d)]).astype(np.float32)
    lists, members = build(nat, x, cent)
    lists.set_prefilter("bf16")
    dist, rows = lists.search(q, 10, 40)
    assert lists.search_stats() == (True, False)
    lists.set_prefilter(None)
    d2, r2 = lists.search(q, 10, 40)
    np.testing.assert_array_equal(rows, r2)
    np.testing.assert_array_equal(dist.view(np.uint32), d2.view(np.uint32))
    for i in range(len(q)):
        ref_rows, ref_dist = R.ivf_search(q[i], x, cent, members, 10, 40, "euclidean")
        np.testing.assert_array_equal(dist[i, :len(ref_rows)], ref_dist.astype(np.float32))
    lists.close()


def test_copy_follows_appends_moves_and_removals(nat):
    """The bf16 copy is refreshed from the position ranges that writes touched: rows appended behind their lists,
    lists that moved to the end of the buffer when they outgrew their place, rows swapped in by removals."""
    rng = np.random.RandomState(10)
    n0, d, nl = 8000, 64, 16
    x = rng.randn(n0 + 6000, d).astype(np.float32)
    cent = x[rng.choice(n0, nl, replace=False)].copy()
    assign = R.ivf_assign_batch(x, cent, "euclidean").astype(np.int32)
    q = rng.randn(25, d).astype(np.float32)
    lists = nat.IvfLists(d, nat.EUCLIDEAN)
    lists.set_centroids(cent)
    lists.build(x[:n0], assign[:n0])
    lists.set_prefilter("bf16")
    members = [list(np.flatnonzero(assign[:n0] == l)) for l in range(nl)]

    def verify():
        mem = [np.asarray(m, dtype=np.int64) for m in members]
        dist, rows = lists.search(q, 10, 6)
        assert lists.search_stats() == (True, False)
        for i in range(len(q)):
            ref_rows, ref_dist = R.ivf_search(q[i], x, cent, mem, 10, 6, "euclidean")
            np.testing.assert_array_equal(rows[i], ref_rows)
            np.testing.assert_array_equal(dist[i], ref_dist.astype(np.float32))

    verify()
    at = n0
    for step in range(12):
        count = [1, 3, 40, 700][step % 4]
        new = np.arange(at, at + count)
        lists.append(x[new], new, assign[new])
        for r in new:
            members[assign[r]].append(int(r))
        at += count
        if step % 3 == 2:
            for _ in range(5):                                     # swap-removals inside a list (index/ivf.py:708-745)
                l = int(rng.randint(nl))
                if len(members[l]) < 2:
                    continue
                pos = int(rng.randint(len(members[l])))
                lists.remove_at(l, pos)
                members[l][pos] = members[l][-1]
                members[l].pop()
        verify()
    live, positions, dead = lists.layout()
    assert dead > 0, "no list outgrew its place: the moved-list path was not exercised"
    lists.close()


def test_index_class_uses_it_and_keeps_the_reference_results(nat):
    """IVFIndex.search_batch against the reference's own outputs (ivf.npz: rows, order, distances at three n_probe
    values and behind a filter) with the copy on and off."""
    from conftest import load_golden
    from vectordb_amd.index import IVFIndex
    g = load_golden("ivf.npz")
    x, q, ref_assign = g["x"], g["q"], g["euclidean_assign_batch"]
    for mode in ("auto", "bf16", None):
        ix = IVFIndex(dimension=32, metric="euclidean", n_clusters=32, n_probe=8, seed=42, prefilter=mode)
        ix._set_centroids(g["euclidean_centroids"])
        for i in range(len(x)):
            ix._append(f"v{i}", x[i], {"cat": int(i % 3)}, int(ref_assign[i]))
        for npb in (1, 8, 32):
            got = ix.search_batch(q, k=10, n_probe=npb)
            assert ix._dev.search_stats() == (mode is not None, False)
            for i, res in enumerate(got):
                ref_rows, ref_dist = g[f"euclidean_np{npb}_rows"][i], g[f"euclidean_np{npb}_dist"][i]
                n = int((ref_rows >= 0).sum())
                assert [int(r.id[1:]) for r in res] == ref_rows[:n].tolist()
                assert [r.distance for r in res] == ref_dist[:n].tolist()
        got = ix.search_batch(q, k=10, n_probe=8, filter_fn=lambda _id, m: m["cat"] == 1)
        for i, res in enumerate(got):
            ref_rows, ref_dist = g["euclidean_filter_rows"][i], g["euclidean_filter_dist"][i]
            n = int((ref_rows >= 0).sum())
            assert [int(r.id[1:]) for r in res] == ref_rows[:n].tolist()
            assert [r.distance for r in res] == ref_dist[:n].tolist()
        assert IVFIndex.from_dict(ix.to_dict())._prefilter == mode
