#!/usr/bin/env python3
"""Worst-case rounding for the bf16 filter passes (run by tests/test_gpu_prefilter.py in a process of its own, with
VDB_PREFILTER_SAMPLE=64 so that the Flat bound pass sees every row).

Every element of the query and of two families of rows sits just beside a bf16 rounding boundary, with the signs
arranged so that all rounding errors of a dot product add up:

  a = 1 + 2^-8 (1 + 1/16)   rounds up to 1 + 2^-7          b = 1 + 2^-8 (1 - 1/16)   rounds down to 1
  query        q = [ a ... | b ... | tuning ]
  "over"  rows c = [ a ... | -b ... ]      bf16 dot too large by about 2^-7 |q||c|  (looks better than it is)
  "under" rows c = [-a ... |  b ... ]      bf16 dot too small by about 2^-7 |q||c|  (looks worse than it is)

The tuning columns hold values that bf16 represents exactly; they shift the true scores so that the k best rows are
all "under" rows while the bf16 scores put every "over" row ahead of them, with a gap of almost twice the largest
error a row can have.  The bound comes from the "over" rows (ten of them or more are among the rows the bound pass
reads), so the "under" rows survive only if the slack covers the full worst-case error of the bound row plus that of
the candidate.  A slack from half the error (2^-9 per operand pair instead of 2^-8) loses them; the script checks
that property of its own data before it asks the library.  All other rows are far away.

Flat: dot product (sharp), cosine, both Euclidean forms (equality with the fp32 search only).  IVF: Euclidean, one
list, "over" rows among the first rows visited (the bound), "under" rows behind them."""

import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _native as N            # noqa: E402
from oracle import ref_numpy as R                # noqa: E402

H = 2.0 ** -8
A = np.float32(1.0 + H * (1.0 + 1.0 / 16.0))
B = np.float32(1.0 + H * (1.0 - 1.0 / 16.0))
HALF, TUNE = 188, 8
D = 2 * HALF + TUNE


def bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()


def family(sign_a, sign_b, tune):
    row = np.zeros(D, np.float32)
    row[:HALF] = sign_a * A
    row[HALF:2 * HALF] = sign_b * B
    row[2 * HALF:2 * HALF + len(tune)] = tune
    return row


def build(n, n_over, n_under, q_tune, over_tune, under_tune, seed):
    rng = np.random.default_rng(seed)
    q = family(+1, +1, q_tune)
    x = (-0.5 * q[None, :] + 0.01 * rng.standard_normal((n, D))).astype(np.float32)        # far from q in every metric
    place = rng.permutation(n)[:n_over + n_under]
    over_at, under_at = np.sort(place[:n_over]), np.sort(place[n_over:])
    x[over_at] = family(+1, -1, over_tune)
    x[under_at] = family(-1, +1, under_tune)
    return q, x, over_at, under_at


def sharpness(name, true_key, approx_key, over_at, under_at, k, smaller_is_better):
    """The data do what the header says: the k best are `under` rows, bf16 ranks every `over` row ahead of them, and
    the gap is more than 1.9 times the largest error of a single row."""
    sgn = 1.0 if smaller_is_better else -1.0
    t, a = sgn * true_key, sgn * approx_key
    best = np.argsort(t, kind="stable")[:k]
    assert set(best) <= set(under_at), f"{name}: the true k best are not all `under` rows"
    assert a[over_at].max() < a[under_at].min(), f"{name}: bf16 does not rank the `over` rows first"
    err = np.abs(a - t)[np.concatenate([over_at, under_at])].max()
    gap = a[under_at].min() - a[over_at].max()
    assert gap > 1.9 * err, f"{name}: gap {gap} is not close to twice the error {err}"
    print(f"{name}: largest bf16 error of a row {err:.4f}, bf16 gap between the families {gap:.4f} = {gap / err:.3f} errors; "
          f"true gap {t[over_at].min() - t[under_at].max():.4f}")


def flat_case():
    n, k = 512 * 128 + 100, 10
    # tuning column 0: q = 0.5, `under` rows 1.0 (adds exactly 0.5 to their dot product)
    q, x, over_at, under_at = build(n, 64, 16, [0.5], [0.0], [1.0], seed=1)
    true_dot = x.astype(np.float64) @ q.astype(np.float64)
    approx_dot = bf16(x).astype(np.float64) @ bf16(q).astype(np.float64)
    sharpness("flat dot", true_dot, approx_dot, over_at, under_at, k, smaller_is_better=False)
    rng = np.random.default_rng(2)
    sharp = np.stack([q] * 4)                                    # four queries: the batch path
    # permuted and random queries see the far rows nearly tied (they differ by noise below the slack): every row stays
    # a candidate and the fp32 search may have to repeat the batch, which is allowed; results must not change
    others = np.stack([rng.permutation(q) for _ in range(3)] + [rng.standard_normal(D).astype(np.float32) for _ in range(5)])
    store = N.RowStore(D)
    store.append(x)
    for metric in ("dot", "cosine", "euclidean", "euclidean_squared"):
        code = N.METRIC_CODES[metric]
        for queries, must_fit in ((sharp, True), (others, False)):
            store.set_prefilter(None)
            d0, r0 = store.search(queries, k, code)
            store.set_prefilter("bf16")
            d1, r1 = store.search(queries, k, code)
            assert store.used_prefilter(), f"{metric}: the filter pass did not run"
            assert not (must_fit and store.search_stats()[1]), f"{metric}: a candidate queue overflowed (the case is meant to fit)"
            assert np.array_equal(r0, r1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)), f"{metric}: results differ"
            if metric == "dot" and must_fit:
                assert set(r1[0]) <= set(under_at), "dot: the `under` rows were lost"
                ref_rows, ref_d = R.flat_search(q, x, k, "dot")
                assert set(ref_rows) == set(r1[0])
    store.close()
    print("flat: identical to the fp32 search for dot, cosine, euclidean, euclidean_squared")


def ivf_case():
    n, k = 6000, 10
    # tuning columns: q = 0; `over` rows carry 0.5 in three of them (0.75 more squared distance, exact in bf16)
    q, x, over_at, under_at = build(n, 32, 16, [0.0], [0.5, 0.5, 0.5], [0.0], seed=3)
    # one list in row order; the bound comes from the first 2048 rows visited: all `over` rows go there
    first = np.arange(32)
    x[[*first, *over_at]] = x[[*over_at, *first]]
    over_at = first
    under_at = np.setdiff1d(under_at, first)
    assert len(under_at) >= k
    diff = x.astype(np.float64) - q.astype(np.float64)
    true_sq = (diff ** 2).sum(1)
    xb, qb = bf16(x).astype(np.float64), bf16(q).astype(np.float64)
    approx_sq = (q.astype(np.float64) ** 2).sum() + (x.astype(np.float64) ** 2).sum(1) - 2.0 * (xb @ qb)
    sharpness("ivf euclidean", true_sq, approx_sq, over_at, under_at, k, smaller_is_better=True)
    cent = np.zeros((1, D), np.float32)
    lists = N.IvfLists(D, N.EUCLIDEAN)
    lists.set_centroids(cent)
    lists.build(x, np.zeros(n, np.int32))
    rng = np.random.default_rng(4)
    # (a permuted query sees the far rows nearly tied: all of them stay candidates and the streaming scan may repeat
    # the batch; only the results are compared for it)
    for queries, must_fit in ((np.stack([q, q]), True), (np.stack([q, rng.permutation(q), q]), False)):
        lists.set_prefilter(None)
        d0, r0 = lists.search(queries, k, 1)
        lists.set_prefilter("bf16")
        d1, r1 = lists.search(queries, k, 1)
        used, overflowed = lists.search_stats()
        assert used and not (must_fit and overflowed), (used, overflowed)
        assert np.array_equal(r0, r1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)), "ivf: results differ"
        assert set(r1[0]) <= set(under_at), "ivf: the `under` rows were lost"
    ref_rows, ref_d = R.ivf_search(q, x, cent, [np.arange(n)], k, 1, "euclidean")
    assert np.array_equal(ref_rows, r1[0]) and np.array_equal(ref_d.astype(np.float32), d1[0])
    lists.close()
    print("ivf: identical to the streaming scan and to the oracle")


if __name__ == "__main__":
    N.init(0)
    flat_case()
    ivf_case()
    print("no failure")
