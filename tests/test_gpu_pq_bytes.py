"""Batched PQ searches through the one-byte-table pass (vdb_pq_search_stats) against the oracle's scan.

The reference sums the fp32 table entries of a row in sub-space order and takes the root (index/pq.py:285-297); equal
distances keep insertion order.  The byte tables only choose the rows that get that sum, so every case demands the
oracle's rows and distance bits: every supported number of sub-spaces, batches that do and do not fill the groups of
eight queries, k from 1 to 32, filters from dense to nearly empty, duplicated codes (ties), a filter that leaves the
bound pass without k rows (the fp32 scan answers), and queries far from all codewords or on top of one."""

import os

import numpy as np
import pytest

from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    old = os.environ.get("VDB_PQ_BOUND_MIN_TILES")
    os.environ["VDB_PQ_BOUND_MIN_TILES"] = "64"          # code sets from 16 384 rows on take the bound pass
    yield _native
    if old is None:
        os.environ.pop("VDB_PQ_BOUND_MIN_TILES", None)
    else:
        os.environ["VDB_PQ_BOUND_MIN_TILES"] = old


def make(nat, n, d, m, seed, scale=1.0):
    rng = np.random.RandomState(seed)
    book = (rng.randn(m, 256, d // m) * scale).astype(np.float32)
    codes = rng.randint(0, 256, size=(n, m)).astype(np.uint8)
    pq = nat.PqCodes(d, m, 256)
    pq.set_codebook(book)
    pq.append_codes(codes)
    return rng, book, codes, pq


def check(nat, pq, book, codes, q, k, mask=None, expect=(True, False)):
    words = nat.pack_mask(mask) if mask is not None else None
    dist, rows = pq.search(q, k, words)
    assert pq.search_stats() == expect
    for i in range(len(q)):
        ref_rows, ref_dist = R.pq_search(q[i], book, codes, k, mask)
        n = len(ref_rows)
        np.testing.assert_array_equal(rows[i, :n], ref_rows, err_msg=f"query {i}")
        np.testing.assert_array_equal(dist[i, :n].view(np.uint32), ref_dist.astype(np.float32).view(np.uint32), err_msg=f"query {i}")
        assert (rows[i, n:] == -1).all() and np.isinf(dist[i, n:]).all()


@pytest.mark.parametrize("m", [4, 8, 16, 32])
def test_matches_the_scan_of_the_reference(nat, m):
    d = 64
    rng, book, codes, pq = make(nat, 150_000, d, m, 10 + m)
    codes[70_000] = codes[5]; codes[149_999] = codes[5]; codes[20] = codes[5]      # equal distances: the lower row first
    pq.truncate(0)
    pq.append_codes(codes)
    q = rng.randn(70, d).astype(np.float32)
    q[0] = np.concatenate([book[j, codes[5, j]] for j in range(m)])                  # a query on top of a stored code
    for nq in (2, 3, 8, 9, 70):
        check(nat, pq, book, codes, q[:nq], 10)
    check(nat, pq, book, codes, q[:5], 1)
    check(nat, pq, book, codes, q[:5], 32)
    pq.close()


def test_filters(nat):
    d, m = 128, 16
    rng, book, codes, pq = make(nat, 400_000, d, m, 3)
    q = rng.randn(11, d).astype(np.float32)
    for keep in (0.5, 0.01, 0.001):
        mask = rng.rand(len(codes)) < keep
        check(nat, pq, book, codes, q, 10, mask=mask)
    # fewer than k rows pass among the rows of the bound pass: no bound, the fp32 scan answers the batch
    mask = np.zeros(len(codes), dtype=bool)
    mask[rng.choice(len(codes), 5, replace=False)] = True
    check(nat, pq, book, codes, q, 10, mask=mask, expect=(True, True))
    check(nat, pq, book, codes, q, 10, mask=np.zeros(len(codes), dtype=bool), expect=(True, True))
    check(nat, pq, book, codes, q, 10)                                             # and the flag does not stick
    pq.close()


def test_scales_and_far_queries(nat):
    d, m = 64, 8
    for scale in (1e-6, 1.0, 1e6):
        rng, book, codes, pq = make(nat, 100_000, d, m, 5, scale)
        q = (rng.randn(9, d) * scale).astype(np.float32)
        q[1] *= 1000.0                                                              # far from every codeword
        q[2] = 0.0
        check(nat, pq, book, codes, q, 10)
        pq.close()


def test_one_query_and_small_sets_keep_the_fp32_scan(nat):
    d, m = 64, 8
    rng, book, codes, pq = make(nat, 100_000, d, m, 6)
    q = rng.randn(4, d).astype(np.float32)
    check(nat, pq, book, codes, q[:1], 10, expect=(False, False))
    pq.truncate(10_000)
    check(nat, pq, book, codes[:10_000], q, 10, expect=(False, False))
    pq.close()


def test_same_bits_as_the_fp32_scan_on_a_large_set(nat):
    d, m = 128, 16
    rng, book, codes, pq = make(nat, 3_000_000, d, m, 8)
    q = rng.randn(16, d).astype(np.float32)
    mask = rng.rand(len(codes)) < 0.3
    words = nat.pack_mask(mask)
    for w, k in ((None, 10), (words, 10), (None, 32)):      # k = 32: the largest a single pass selects
        dist, rows = pq.search(q, k, w)
        assert pq.search_stats() == (True, False)
        os.environ["VDB_PQ_NO_BYTE_TABLES_NOW"] = "1"
        try:
            d2, r2 = pq.search(q, k, w)
            assert pq.search_stats() == (False, False)
        finally:
            os.environ.pop("VDB_PQ_NO_BYTE_TABLES_NOW")
        np.testing.assert_array_equal(rows, r2)
        np.testing.assert_array_equal(dist.view(np.uint32), d2.view(np.uint32))
    pq.close()
