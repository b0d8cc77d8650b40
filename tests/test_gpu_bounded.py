"""Batched searches with the direct-form metrics (Manhattan, Chebyshev, per-pair Euclidean) over a large flat row
set: the k best of the first rows bound every query's k-th distance, the scan of the whole set queues only the pairs
within that bound, and a queue that fills up makes the plain pipeline repeat the search.  Whatever path ran, rows,
order and bits must be the oracle's (ascending distance, equal distances by row number)."""

import numpy as np
import pytest

from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def oracle_dist(q, x, name):
    if name == "euclidean_direct":
        return np.sqrt(np.sum((x - q) ** 2, axis=1))                    # metrics.py:43, row by row
    if name == "euclidean_sq_direct":
        return np.sum((x - q) ** 2, axis=1)
    return R.query_distances(q, x, name)                                # metrics.py:486-490


def check(nat, x, q, k, name, mask=None):
    code = {"manhattan": nat.MANHATTAN, "chebyshev": nat.CHEBYSHEV, "euclidean_direct": nat.EUCLIDEAN_DIRECT,
            "euclidean_sq_direct": nat.EUCLIDEAN_SQ_DIRECT}[name]
    store = nat.RowStore(x.shape[1])
    store.append(x)
    dist, rows = store.search(q, k, code, nat.pack_mask(mask))
    store.close()
    for i in range(len(q)):
        d = oracle_dist(q[i], x, name).astype(np.float32)
        cand = np.arange(len(x)) if mask is None else np.flatnonzero(mask)
        order = cand[np.lexsort((cand, d[cand]))][:k]
        m = len(order)
        assert np.array_equal(rows[i, :m], order), (name, i)
        assert np.array_equal(dist[i, :m], d[order]), (name, i)
        assert (rows[i, m:] == -1).all()
    return dist, rows


@pytest.mark.parametrize("name", ["manhattan", "chebyshev", "euclidean_direct", "euclidean_sq_direct"])
def test_bounded_batch_equals_oracle(nat, name):
    rng = np.random.default_rng(17)
    n, d = 600_000, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[n - 5] = x[100]                       # equal rows far apart: equal distances, the lower row first
    x[n - 9] = x[77]
    q = rng.standard_normal((9, d), dtype=np.float32)
    q[0] = x[100] + np.float32(0.01)
    q[1] = x[n - 9]
    check(nat, x, q, 10, name)
    check(nat, x, q[:2], 1, name)
    check(nat, x, q, 32, name, mask=rng.random(n) < 0.3)
    big = rng.standard_normal((300, d), dtype=np.float32)          # more than one query group and more than one block of queries
    dist, rows = check(nat, x, big[:3], 10, name)
    store = nat.RowStore(d)
    store.append(x)
    code = {"manhattan": nat.MANHATTAN, "chebyshev": nat.CHEBYSHEV, "euclidean_direct": nat.EUCLIDEAN_DIRECT,
            "euclidean_sq_direct": nat.EUCLIDEAN_SQ_DIRECT}[name]
    dist_all, rows_all = store.search(big, 10, code)
    store.close()
    assert np.array_equal(rows_all[:3], rows) and np.array_equal(dist_all[:3], dist)
    for i in (150, 299):
        dd = oracle_dist(big[i], x, name).astype(np.float32)
        order = np.lexsort((np.arange(n), dd))[:10]
        assert np.array_equal(rows_all[i], order) and np.array_equal(dist_all[i], dd[order])


def test_bounded_batch_full_queue_falls_back(nat):
    """The first rows (the ones the bound comes from) are the farthest, and every other row is within the bound:
    the queues fill up and the plain pipeline repeats the search."""
    rng = np.random.default_rng(18)
    n, d = 600_000, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[:70_000] += np.float32(50.0)
    q = rng.standard_normal((5, d), dtype=np.float32)
    check(nat, x, q, 10, "manhattan")
    check(nat, x, q, 10, "euclidean_direct", mask=rng.random(n) < 0.5)


def test_bounded_batch_filter_leaves_no_bound(nat):
    """A filter that passes none of the first rows: no bound, everything that passes is queued (and overflows)."""
    rng = np.random.default_rng(19)
    n, d = 600_000, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((4, d), dtype=np.float32)
    mask = np.zeros(n, bool)
    mask[500_000:] = rng.random(100_000) < 0.2
    check(nat, x, q, 10, "chebyshev", mask=mask)
    few = np.zeros(n, bool)
    few[[5, 400_000, 599_999]] = True
    check(nat, x, q, 10, "manhattan", mask=few)
