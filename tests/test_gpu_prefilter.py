"""Batched Flat searches behind a bf16 filter pass (vdb_rows_set_prefilter): the pass over all rows runs on
bf16 copies with a loosened bound, the fp32 dot products of the rows that can still be among the k best are
recomputed with the instruction sequence of the fp32 kernel.  The bar is not a tolerance: distances, rows and
their order must be bit for bit those of the same handle with the filter off, on random rows, ragged ends,
filters, near-ties by the thousand, exact ties, extreme magnitudes, and across writes; plus the usual parity
with the CPU oracle."""

import numpy as np
import pytest

from test_gpu_cabi import check_search

pytestmark = pytest.mark.gpu

N_MIN = 512 * 128           # smallest row count that gets a sampled bound
METRICS = ["euclidean", "euclidean_squared", "cosine", "dot"]


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def both_ways(nat, store, q, k, metric, mask=None, expect_filter=True, expect_overflow=False):
    """Search with the filter off and on; returns the (identical) result."""
    code, words = nat.METRIC_CODES[metric], nat.pack_mask(mask)
    store.set_prefilter(None)
    d0, r0 = store.search(q, k, code, words)
    assert not store.used_prefilter()
    store.set_prefilter("bf16")
    d1, r1 = store.search(q, k, code, words)
    assert store.used_prefilter() == expect_filter
    if expect_overflow is not None:
        assert store.search_stats()[1] == expect_overflow
    np.testing.assert_array_equal(r1, r0)
    np.testing.assert_array_equal(d1.view(np.uint32), d0.view(np.uint32))
    return d1, r1


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("d,n_extra", [(64, 0), (128, 77), (384, 5), (60, 131), (100, 3), (7, 64), (200, 0)])
def test_prefilter_equals_fp32_search(nat, metric, d, n_extra):
    """Dimensions that are not a multiple of 64: the bf16 copy is padded to whole 128-byte column steps, the fp32
    rows end in a partial step (d = 60, 100, 200) or are narrower than one (d = 7)."""
    rng = np.random.default_rng(d + n_extra)
    n = N_MIN + n_extra
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((300, d), dtype=np.float32)         # two 256-query tiles, the second ragged
    store = nat.RowStore(d)
    store.append(x)
    for k in (1, 10, 32):
        both_ways(nat, store, q, k, metric)
    both_ways(nat, store, q[:17], 10, metric)
    store.close()


def test_prefilter_matches_oracle(nat, monkeypatch):
    rng = np.random.default_rng(8)
    n, d = N_MIN + 300, 128
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((24, d), dtype=np.float32)
    made = []
    plain = nat.RowStore

    def with_filter(*a, **kw):
        s = plain(*a, **kw)
        s.set_prefilter("bf16")
        made.append(s)
        return s

    monkeypatch.setattr(nat, "RowStore", with_filter)
    mask = rng.random(n) < 0.3
    for metric in METRICS:
        check_search(nat, x, q, 10, metric)
        check_search(nat, x, q[:6], 10, metric, mask=mask)
    assert len(made) == 8


def test_prefilter_shapes_it_does_not_serve(nat):
    """k > 32, a single query and bf16 storage keep the kernels they had (or are refused)."""
    rng = np.random.default_rng(2)
    n = N_MIN
    x = rng.standard_normal((n, 64), dtype=np.float32)
    q = rng.standard_normal((20, 64), dtype=np.float32)
    store = nat.RowStore(64)
    store.append(x)
    both_ways(nat, store, q, 40, "cosine", expect_filter=False)
    both_ways(nat, store, q[:1], 10, "cosine", expect_filter=False)         # one query: the streaming scan
    store.close()
    store = nat.RowStore(64, storage="bf16")
    with pytest.raises(ValueError):
        store.set_prefilter("bf16")
    store.close()


@pytest.mark.parametrize("metric", METRICS)
def test_prefilter_with_mask(nat, metric):
    rng = np.random.default_rng(3)
    n, d = N_MIN + 1000, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((40, d), dtype=np.float32)
    store = nat.RowStore(d)
    store.append(x)
    for keep in (0.5, 0.01):
        mask = rng.random(n) < keep
        mask[-3:] = True
        both_ways(nat, store, q, 10, metric, mask=mask, expect_overflow=None)
    mask = np.zeros(n, dtype=bool)
    mask[[5, n - 1]] = True                                      # fewer passing rows than k
    d1, r1 = both_ways(nat, store, q[:4], 10, metric, mask=mask, expect_overflow=None)
    assert (np.sort(r1[:, :2], axis=1) == [5, n - 1]).all() and (r1[:, 2:] == -1).all()
    store.close()


@pytest.mark.parametrize("metric", METRICS)
def test_prefilter_near_ties_and_exact_ties(nat, metric):
    """A cloud of rows much closer to each other than bf16 can tell apart around every query: hundreds of rows
    lie within the rounding slack of the k-th best and all of them have to be looked at in fp32; copies of one row
    tie exactly and come out in row order."""
    rng = np.random.default_rng(17)
    n, d = N_MIN, 128
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((8, d), dtype=np.float32)
    for i in range(8):
        at = rng.choice(n, size=400, replace=False)
        x[at] = (3.0 * q[i] + 1e-3 * rng.standard_normal((400, d))).astype(np.float32)
        x[at[:3]] = x[at[3]]                                     # four identical rows
    store = nat.RowStore(d)
    store.append(x)
    d1, r1 = both_ways(nat, store, q, 10, metric, expect_overflow=None)
    if metric != "dot":
        for i in range(8):
            same = np.flatnonzero(d1[i, 1:] == d1[i, :-1])
            assert (r1[i, same + 1] > r1[i, same]).all()
    store.close()


@pytest.mark.parametrize("scale_x,scale_q", [(1e15, 1e-3), (1e-15, 1e10), (1e-18, 1e-18), (1e25, 1e20), (1.0, 0.0)])
def test_prefilter_extreme_magnitudes(nat, scale_x, scale_q):
    """Large and tiny values; products that overflow fp32 (the search is then left to the fp32 kernels); zero queries."""
    rng = np.random.default_rng(23)
    n, d = N_MIN, 64
    x = (scale_x * rng.standard_normal((n, d))).astype(np.float32)
    x[7] = 0.0
    q = (scale_q * rng.standard_normal((9, d))).astype(np.float32)
    store = nat.RowStore(d)
    store.append(x)
    for metric in METRICS:
        both_ways(nat, store, q, 10, metric, expect_overflow=None)
    store.close()


def test_prefilter_mixed_row_norms(nat):
    """Row norms spread over six orders of magnitude: the slack of dot and Euclidean searches follows the largest."""
    rng = np.random.default_rng(29)
    n, d = N_MIN, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    x *= (10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    q = rng.standard_normal((12, d), dtype=np.float32)
    store = nat.RowStore(d)
    store.append(x)
    for metric in METRICS:
        both_ways(nat, store, q, 10, metric, expect_overflow=None)
    store.close()


def test_prefilter_follows_writes(nat):
    """Appends, overwritten rows and a truncation followed by new rows all reach the bf16 copy."""
    rng = np.random.default_rng(31)
    n, d = N_MIN + 500, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((16, d), dtype=np.float32)
    store = nat.RowStore(d)
    store.set_prefilter("bf16")
    store.append(x[:N_MIN])
    code = nat.METRIC_CODES["euclidean"]
    store.search(q, 10, code)
    assert store.used_prefilter()
    store.append(x[N_MIN:])
    both_ways(nat, store, q, 10, "euclidean")
    # rows equal to the queries, written in place: they must now be found at distance ~0
    for i in range(4):
        store.set_row(1000 * (i + 1), q[i])
    d1, r1 = both_ways(nat, store, q, 10, "euclidean", expect_overflow=None)
    assert [int(r1[i, 0]) for i in range(4)] == [1000, 2000, 3000, 4000]
    # other rows behind a truncation
    store.truncate(N_MIN - 100)
    y = rng.standard_normal((700, d), dtype=np.float32)
    y[5] = q[9]
    store.append(y)
    d1, r1 = both_ways(nat, store, q, 10, "euclidean", expect_overflow=None)
    assert int(r1[9, 0]) == N_MIN - 100 + 5
    fresh = nat.RowStore(d)
    z = x[:N_MIN - 100].copy()
    for i in range(4):
        z[1000 * (i + 1)] = q[i]
    fresh.append(np.concatenate([z, y]))
    d2, r2 = fresh.search(q, 10, code)
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(d1, d2)
    fresh.close()
    store.close()


def test_prefilter_config1_full_size(nat):
    """Flat cosine 1M x 384, 1024 queries, k = 10: the bf16 filter pass and the fp32 kernels agree bit for bit."""
    rng = np.random.default_rng(42)
    n, d = 1_000_000, 384
    store = nat.RowStore(d)
    for o in range(0, n, 125_000):
        store.append(rng.standard_normal((125_000, d), dtype=np.float32))
    q = rng.standard_normal((1024, d), dtype=np.float32)
    both_ways(nat, store, q, 10, "cosine")
    both_ways(nat, store, q, 10, "dot")
    store.close()


def test_prefilter_through_the_index_and_collection_classes(nat):
    """FlatIndex / Collection with prefilter="bf16" and with the default ("auto": on while the copy fits): search_batch
    returns what the same classes return with prefilter=None (ids, distances, scores), and the device store reports
    the filter pass."""
    from vectordb_amd.index.flat import FlatIndex
    from vectordb_amd.core.collection import Collection
    rng = np.random.default_rng(41)
    n, d = N_MIN + 50, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((20, d), dtype=np.float32)
    ids = [f"v{i}" for i in range(n)]
    plain, filt, auto = FlatIndex(d, metric="cosine", prefilter=None), FlatIndex(d, metric="cosine", prefilter="bf16"), FlatIndex(d, metric="cosine")
    for ix in (plain, filt, auto):
        ix.add_batch(ids, x)
    a, b, c = plain.search_batch(q, 10), filt.search_batch(q, 10), auto.search_batch(q, 10)
    assert filt._store.used_prefilter() and auto._store.used_prefilter() and not plain._store.used_prefilter()
    assert [[(r.id, r.distance) for r in rs] for rs in a] == [[(r.id, r.distance) for r in rs] for rs in b]
    assert [[(r.id, r.distance) for r in rs] for rs in a] == [[(r.id, r.distance) for r in rs] for rs in c]
    odd = lambda id, metadata: int(id[1:]) % 2 == 1
    a, b = plain.search_batch(q, 5, filter_fn=odd), filt.search_batch(q, 5, filter_fn=odd)
    assert [[(r.id, r.distance) for r in rs] for rs in a] == [[(r.id, r.distance) for r in rs] for rs in b]

    cols = [Collection("c0", d, metric="euclidean", prefilter=None), Collection("c1", d, metric="euclidean", prefilter="bf16")]
    for c in cols:
        c.add_batch([{"id": ids[i], "vector": x[i], "metadata": {"g": i % 3}} for i in range(n)])
    a, b = (c.search_batch(q, 10) for c in cols)
    assert cols[1]._store.used_prefilter() and not cols[0]._store.used_prefilter()
    assert [[(r.id, r.distance, r.score) for r in rs] for rs in a] == [[(r.id, r.distance, r.score) for r in rs] for rs in b]
    a, b = (c.search_batch(q, 10, filter={"g": 1}) for c in cols)
    assert [[(r.id, r.distance) for r in rs] for rs in a] == [[(r.id, r.distance) for r in rs] for rs in b]
    with pytest.raises(ValueError):
        Collection("c2", d, prefilter="fp8")


def test_prefilter_more_than_one_block_of_queries(nat):
    """1100 queries: the queue pass runs in blocks of 1024 queries, each with its own bounds, slack and fp32 queries."""
    rng = np.random.default_rng(43)
    n, d = N_MIN + 17, 32
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((1100, d), dtype=np.float32)
    store = nat.RowStore(d)
    store.append(x)
    for metric in ("cosine", "euclidean"):
        both_ways(nat, store, q, 10, metric)
    store.close()


def test_prefilter_recovers_from_an_overwritten_outlier(nat):
    """One row with a norm a million times the others makes the slack of dot searches useless (every row passes the
    loosened bound, the queues fill up, the fp32 kernel repeats the batch); once it is overwritten the largest norm
    is taken afresh and the filter pass works again.  Results equal the fp32 search either way."""
    rng = np.random.default_rng(47)
    n, d = N_MIN, 64
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[123] *= 1e6
    q = rng.standard_normal((16, d), dtype=np.float32)
    store = nat.RowStore(d)
    store.append(x)
    both_ways(nat, store, q, 10, "dot", expect_overflow=True)
    store.set_row(123, rng.standard_normal(d).astype(np.float32))
    both_ways(nat, store, q, 10, "dot", expect_overflow=False)
    store.close()


def test_worst_case_rounding_is_inside_the_slack(native):
    """tests/worstcase_prefilter.py: every element half a bf16 step from its rounded value, the errors of a dot product
    all of one sign, the true k best separated from the rows that bf16 prefers by almost twice a row's largest error.
    In a process of its own: the Flat bound pass must read every row (VDB_PREFILTER_SAMPLE is read once)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, VDB_PREFILTER_SAMPLE="64")
    r = subprocess.run([sys.executable, os.path.join(here, "worstcase_prefilter.py")], capture_output=True, text=True,
                       timeout=300, env=env)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0 and "no failure" in r.stdout, tail
