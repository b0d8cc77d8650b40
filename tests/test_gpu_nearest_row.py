"""Nearest row of every query of a large batch (k = 1, Euclidean forms): the running-minimum kernel against the list
kernel it replaces for k-means assignment and index builds (KMeans._assign, index/ivf.py:228-231: identity-form
distances, clamp, sqrt, argmin = first minimum).  The two must agree bit for bit, rows and distances, also where
centroids are duplicated or sit on top of rows (keys tie or clamp to zero: those queries go through the list kernel
again), and both must agree with the oracle up to rounding-level ties of its BLAS products."""

import os

import numpy as np
import pytest

from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def both(nat, x, cent, metric):
    os.environ.pop("VDB_NO_ARGMIN", None)
    a1, d1 = nat.assign(x, cent, metric)
    os.environ["VDB_NO_ARGMIN"] = "1"
    try:
        a0, d0 = nat.assign(x, cent, metric)
    finally:
        os.environ.pop("VDB_NO_ARGMIN")
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_array_equal(d1.view(np.uint32), d0.view(np.uint32))
    return a1, d1


@pytest.mark.parametrize("d,k", [(128, 4096), (32, 256), (64, 1000), (96, 512)])
def test_equals_the_list_kernel(nat, d, k):
    rng = np.random.default_rng(d + k)
    n = 300_000
    x = rng.standard_normal((n, d), dtype=np.float32)
    cent = x[rng.choice(n, k, replace=False)].copy()            # every centroid sits on a row: distance 0, clamped keys
    cent[5] = cent[9]                                           # a duplicated centroid: equal keys, the lower one wins
    cent[k - 1] = cent[3]
    for metric in (nat.EUCLIDEAN, nat.EUCLIDEAN_SQ):
        a, dist = both(nat, x, cent, metric)
        assert not (a == 9).any() and not (a == k - 1).any()
    sample = rng.choice(n, 1500, replace=False)
    a, dist = both(nat, x, cent, nat.EUCLIDEAN)
    ref = R.kmeans_assign(x[sample], cent)
    moved = np.flatnonzero(a[sample] != ref)
    if len(moved):
        full = R.pairwise_euclidean(x[sample[moved]], cent)
        got, want = full[np.arange(len(moved)), a[sample[moved]]], full[np.arange(len(moved)), ref[moved]]
        assert len(moved) <= 3 and np.allclose(got, want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist[sample], np.min(R.pairwise_euclidean(x[sample], cent), axis=1), rtol=1e-4, atol=1e-2)      # a row on top of its centroid: 0 up to the identity form's cancellation


def test_all_rows_equal_and_many_duplicates(nat):
    """Every key ties with another: more unsure queries than the repeat list holds, the whole batch goes through the
    list kernel, same results."""
    rng = np.random.default_rng(3)
    n, d, k = 70_000, 64, 256
    cent = rng.standard_normal((k, d), dtype=np.float32)
    cent[k // 2:] = cent[:k // 2]                               # every centroid twice
    x = rng.standard_normal((n, d), dtype=np.float32)
    a, dist = both(nat, x, cent, nat.EUCLIDEAN)
    assert (a < k // 2).all()
    x[:] = x[0]
    both(nat, x, cent, nat.EUCLIDEAN_SQ)
