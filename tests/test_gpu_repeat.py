"""Repeated small searches on one index reuse its grow-only workspaces and cached plans: every
repetition must return what the first search on a fresh index returns, with and without a
filter, and a change of the index (append, remove, growth past a reallocation) must show in the
next search."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init(native):
    return native


def ids_of(n, start=0):
    return [f"v{i}" for i in range(start, start + n)]


def make(kind, x):
    from vectordb_amd.index import FlatIndex, IVFIndex, PQIndex, HybridIndex
    d = x.shape[1]
    if kind == "flat":
        ix = FlatIndex(dimension=d)
    elif kind == "flat_l1":
        ix = FlatIndex(dimension=d, metric="manhattan")
    elif kind == "ivf":
        ix = IVFIndex(dimension=d, n_clusters=16, n_probe=6, seed=1)
    elif kind == "pq":
        ix = PQIndex(dimension=d, n_subvectors=8, n_bits=8, seed=1)
    else:
        ix = HybridIndex(dimension=d, n_clusters=16, n_probe=6, n_subvectors=8, n_bits=8)
    if kind not in ("flat", "flat_l1"):
        ix.train(x)
    ix.add_batch(ids_of(len(x)), x)
    return ix


@pytest.mark.parametrize("kind", ["flat", "flat_l1", "ivf", "pq", "ivf_pq"])
def test_repeated_searches_equal_first_and_follow_changes(kind):
    rng = np.random.RandomState(3)
    x = rng.randn(4000, 32).astype(np.float32)
    q = rng.randn(12, 32).astype(np.float32)
    ix = make(kind, x)
    # a second index in the same state: every query's first search is a first call there
    fresh = type(ix).from_dict(ix.to_dict())
    first = [[(r.id, r.distance) for r in fresh.search(q[i], k=5)] for i in range(12)]
    # same arguments, different query values
    for rep in range(3):
        for i in range(12):
            got = [(r.id, r.distance) for r in ix.search(q[i], k=5)]
            assert got == first[i], (kind, rep, i)
    # with a filter, repeated
    f = lambda id, m: int(id[1:]) % 2 == 0
    want = [r.id for r in fresh.search(q[0], k=5, filter_fn=f)]
    for rep in range(4):
        assert [r.id for r in ix.search(q[0], k=5, filter_fn=f)] == want
    # batches of a few queries
    b0 = [[r.id for r in res] for res in fresh.search_batch(q[:6], k=5)]
    for rep in range(3):
        assert [[r.id for r in res] for res in ix.search_batch(q[:6], k=5)] == b0
    # the index changes: a row right at q[0] must win the next search, and lose again once removed
    ix.add("new", q[0])
    for rep in range(3):
        assert ix.search(q[0], k=5)[0].id == "new"
    assert ix.remove("new")
    for rep in range(3):
        assert [(r.id, r.distance) for r in ix.search(q[0], k=5)] == first[0]
    # many more rows (reallocation of the row store and of workspaces)
    more = rng.randn(6000, 32).astype(np.float32)
    ix.add_batch(ids_of(len(more), start=100000), more)
    fresh.add_batch(ids_of(len(more), start=100000), more)
    for rep in range(3):
        assert [r.id for r in ix.search(q[3], k=7)] == [r.id for r in fresh.search(q[3], k=7)]


@pytest.mark.parametrize("d", [1, 5, 8, 13, 32, 100, 128, 130, 384, 768, 1000, 3000])
def test_norms_of_a_few_rows_bit_exact(native, d):
    """Up to 64 rows take a one-launch kernel for |x|^2 and |x| (a thread per row), up to 16384 another
    (eight lanes per row), more the streaming kernel: all the same
    pairwise order as np.sum (distance/metrics.py:455-456,469-470), bit for bit.  Checked on
    stored rows, and on query norms through the distance to a zero row:
    sqrt(|q|^2 + 0 - 2*0) in the identity form (distance/metrics.py:453-459)."""
    from vectordb_amd import _native as nat
    rng = np.random.RandomState(d)
    for n in (1, 3, 64, 65, 1000, 16384, 16385):
        x = (rng.randn(n, d) * rng.uniform(0.1, 10.0)).astype(np.float32)
        store = nat.RowStore(d)
        store.append(x)
        sq, nrm = store.norms()
        np.testing.assert_array_equal(sq, np.sum(x * x, axis=1))
        np.testing.assert_array_equal(nrm, np.linalg.norm(x, axis=1))
        store.close()
    zero = nat.RowStore(d)
    zero.append(np.zeros((1, d), np.float32))
    q = rng.randn(3, d).astype(np.float32)
    for i in range(3):
        dist, rows = zero.search(q[i], 1, nat.EUCLIDEAN)
        assert dist[0, 0] == np.sqrt(np.sum(q[i] * q[i])) and rows[0, 0] == 0
    zero.close()
