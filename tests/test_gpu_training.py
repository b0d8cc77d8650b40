"""k-means on resident rows against the reference's own training, one step at a time (tests/golden/round2.npz,
kmeans.npz, pq.npz, hybrid.npz).

The device repeats the reference's additions in the reference's order and types, so wherever its assignments
equal the reference's the new centroids are equal bit for bit.  Assignments come from BLAS in the reference and
from the matrix cores / fma here: a row may change cluster only on a rounding-level tie, which the fixtures do
not contain; the tests therefore ask for exact equality first and fall back to a stated tolerance with the
number of moved rows bounded."""

import numpy as np
import pytest

from oracle import ref_numpy as R
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def trainer_for(nat, x, col0, dsub, k):
    store = nat.RowStore(x.shape[1])
    store.append(x)
    return store, nat.KMeansTrainer(store, col0, dsub, k)


def test_kmeans_pp_and_two_lloyd_steps(nat):
    g = load_golden("round2.npz")
    x = g["km_x"]
    n, k = len(x), 12
    rng = np.random.RandomState(42)
    first = rng.randint(n)
    store, tr = trainer_for(nat, x, 0, x.shape[1], k)
    rows = tr.init_pp(first, rng.random_sample(k - 1))
    np.testing.assert_array_equal(x[rows], g["km_pp_init"])                 # the reference's k-means++ draws
    np.testing.assert_array_equal(tr.centroids(), g["km_pp_init"])
    a = tr.assignments()
    assert (a != g["km_assign_of_pp_init"]).sum() == 0
    for steps in (1, 2):
        counts, shift = tr.step(nat.KMEANS_IVF)
        assert counts.sum() == n and (counts > 0).all()
        tr.commit()
        np.testing.assert_array_equal(tr.centroids(), g[f"km_centroids_after_{steps}"])     # same sums, same order
    assert shift > 0
    tr.close(); store.close()


@pytest.mark.parametrize("d,scale", [(128, 1.0), (32, 1.0), (40, 1e-3), (64, 1e4), (8, 1.0)])
def test_kmeans_pp_rounds_over_the_bf16_copy_choose_the_same_rows(nat, d, scale, monkeypatch):
    """From the second round on k-means++ reads a bf16 copy of the rows and recomputes only the rows a new centroid can
    come nearer to than their minimum (in the distance pass's arithmetic).  The draws must be those of the rounds
    that read every row in float32: random rows, tight clusters (most minima lie within the slack of each other),
    repeated rows."""
    rng = np.random.RandomState(d)
    n, k = 50000, 150
    centres = rng.randn(40, d).astype(np.float32)
    x = np.concatenate([rng.randn(n // 2, d).astype(np.float32),
                        centres[rng.randint(0, 40, n // 2)] + 0.01 * rng.randn(n // 2, d).astype(np.float32)]) * np.float32(scale)
    x[1000:1200] = x[999]
    u = rng.random_sample(k - 1)
    store, tr = trainer_for(nat, x, 0, d, k)
    rows_bf16 = tr.init_pp(123, u)
    monkeypatch.setenv("VDB_KMPP_NO_BF16", "1")
    rows_f32 = tr.init_pp(123, u)
    monkeypatch.delenv("VDB_KMPP_NO_BF16")
    np.testing.assert_array_equal(rows_bf16, rows_f32)
    np.testing.assert_array_equal(tr.centroids(), x[rows_f32])
    tr.close(); store.close()


def test_kmeans_class_matches_reference_fit():
    """KMeans.fit end to end against the reference's fitted centroids (kmeans.npz: 600 x 16, 12 clusters, seed 42)."""
    from vectordb_amd.index.ivf import KMeans
    g = load_golden("kmeans.npz")
    km = KMeans(n_clusters=12, n_iter=20, seed=42)
    cent = km.fit(g["x"])
    exact = np.array_equal(cent, g["centroids"])
    print(f"KMeans.fit vs reference: bit-equal={exact}, max abs diff={np.abs(cent - g['centroids']).max():.3g}")
    np.testing.assert_allclose(cent, g["centroids"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(km.predict(g["x"]), g["labels"])
    few = KMeans(n_clusters=50, seed=1).fit(g["x"][:20])
    np.testing.assert_array_equal(few, g["few_centroids"])


def test_kmeans_empty_clusters_are_reseeded(nat):
    """Two identical starting centroids: the second one gets no members (argmin takes the first minimum), and
    the caller's replacement row lands in the pending centroids (ivf.py:185-190)."""
    rng = np.random.RandomState(2)
    x = rng.randn(500, 6).astype(np.float32)                # 6 columns: rows are padded to 8 on the device
    store, tr = trainer_for(nat, x, 0, 6, 5)
    tr.init_rows([3, 3, 10, 20, 30])
    counts, shift = tr.step(nat.KMEANS_IVF)
    assert counts[1] == 0 and counts.sum() == 500
    shift2 = tr.reseed([1], [77])
    tr.commit()
    cent = tr.centroids()
    np.testing.assert_array_equal(cent[1], x[77])
    assert shift2 != shift
    # the other centroids: the oracle's update from the same assignment
    a = R.kmeans_assign(x, x[[3, 3, 10, 20, 30]])
    ref = R.kmeans_update(x, a, 5, np.random.RandomState(0), x[[3, 3, 10, 20, 30]])
    for c in (0, 2, 3, 4):
        np.testing.assert_array_equal(cent[c], ref[c])
    tr.close(); store.close()


@pytest.mark.parametrize("which", ["pq", "hy"])
def test_subquantizer_steps_match_reference(nat, which):
    """The PQ / IVF-PQ sub-quantizer: fp32 first step, float64 afterwards (pq.py:144-180, hybrid.py:174-203)."""
    g = load_golden("round2.npz")
    sub = g["pq_sub"]
    # the sub-space sits at columns 8..16 of wider rows, as in PQIndex.train
    wide = np.concatenate([np.zeros_like(sub), sub, np.ones_like(sub)], axis=1)
    store, tr = trainer_for(nat, wide, 8, 8, 64)
    tr.init_rows(g[f"{which}_init_rows"])
    steps_wanted = (1, 2, 3, 20) if which == "pq" else (1, 2, 5)
    done = 0
    for steps in steps_wanted:
        while done < steps:
            counts, _ = tr.step(nat.KMEANS_PQ)
            assert counts.sum() == len(sub)
            tr.commit()
            done += 1
        got = tr.centroids(float64=True)
        want = g[f"{which}_centroids_after_{steps}"]
        exact = np.array_equal(got, want)
        print(f"{which} sub-quantizer after {steps} steps: bit-equal={exact}, max abs diff={np.abs(got - want).max():.3g}")
        if steps <= 3:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
        if which == "pq":
            # the assignment the next step would start from (a step without commit leaves the centroids as they are)
            tr.step(nat.KMEANS_PQ)
            assign = tr.assignments().astype(np.int64)
            ref_assign = g[f"pq_assign_from_{steps}"].astype(np.int64)
            if steps <= 3:
                np.testing.assert_array_equal(assign, ref_assign)
            else:
                assert (assign != ref_assign).mean() <= 0.005
    np.testing.assert_array_equal(tr.centroids(), tr.centroids(float64=True).astype(np.float32))
    tr.close(); store.close()


def test_pq_index_train_matches_reference_codebook():
    """PQIndex.train with the reference's seed against the reference's codebook (pq.npz: 1200 x 32, m = 4, 20 steps)."""
    from vectordb_amd.index.pq import PQIndex
    g = load_golden("pq.npz")
    idx = PQIndex(dimension=32, n_subvectors=4, n_bits=8, seed=42)
    idx.train(g["x"])
    diff = np.abs(idx._codebook - g["codebook"])
    print(f"PQIndex.train vs reference codebook: bit-equal={np.array_equal(idx._codebook, g['codebook'])}, "
          f"max abs diff={diff.max():.3g}, entries beyond 1e-6: {(diff > 1e-6).sum()} of {diff.size}")
    np.testing.assert_allclose(idx._codebook, g["codebook"], rtol=1e-4, atol=1e-5)
    # fewer rows than codewords: zero padding (pq.py:172-178)
    small = PQIndex(dimension=32, n_subvectors=8, n_bits=8, seed=1)
    small.train(g["x"][:100])
    np.testing.assert_allclose(small._codebook, g["small_codebook"], rtol=1e-4, atol=1e-5)


def test_hybrid_index_train_matches_reference():
    from vectordb_amd.index.hybrid import HybridIndex
    g = load_golden("hybrid.npz")
    idx = HybridIndex(dimension=32, n_clusters=16, n_probe=4, n_subvectors=4, n_bits=8, seed=11)
    idx.config.n_iter = int(g["n_iter"])
    idx.train(g["x"])
    dc = np.abs(idx._coarse_centroids - g["centroids"]).max()
    db = np.abs(idx._pq_codebook - g["codebook"]).max()
    print(f"HybridIndex.train vs reference: coarse max diff={dc:.3g}, codebook max diff={db:.3g}")
    np.testing.assert_allclose(idx._coarse_centroids, g["centroids"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(idx._pq_codebook, g["codebook"], rtol=1e-4, atol=1e-4)


def test_trainer_odd_shapes(nat):
    """Columns that are not a multiple of four, more clusters than one scan block, rows not a multiple of the
    sort chunk: the sums equal the oracle's sequential sums bit for bit given the device's own assignment."""
    rng = np.random.RandomState(4)
    x = rng.randn(30011, 11).astype(np.float32)
    k = 1500
    store, tr = trainer_for(nat, x, 3, 5, k)
    init = rng.choice(len(x), k, replace=False)
    tr.init_rows(init)
    counts, _ = tr.step(nat.KMEANS_IVF)
    a = tr.assignments()
    assert np.array_equal(np.bincount(a, minlength=k), counts)
    sub = np.ascontiguousarray(x[:, 3:8])
    ref_a = R.kmeans_assign(sub, sub[init])
    assert (a != ref_a).sum() <= 3                            # rounding-level ties only
    tr.commit()
    new = np.zeros((k, 5), np.float32)
    np.add.at(new, a, sub)                                    # sequential fp32, sample order
    fed = counts > 0
    want = (new[fed].astype(np.float64) / counts[fed, None].astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(tr.centroids()[fed], want)
    tr.close(); store.close()


def test_full_size_lloyd_steps_against_the_oracle(nat):
    """BASELINE configs[3] at its stated size: 1 048 576 x 128 rows, 4096 clusters, two Lloyd steps from 4096 rows of
    the set.  Per step: the assignments of 2000 sampled rows against KMeans._assign (index/ivf.py:228-231: identity-form
    distances, clamp, sqrt, argmin; a row may differ only where its two nearest centroids tie to rounding), and the new
    centroids of 16 sampled clusters against the reference's update (ivf.py:177-190: fp32 sums in row order, divided by
    the float64 count), bit for bit."""
    rng = np.random.default_rng(42)
    n, d, k = 1_048_576, 128, 4096
    x = rng.standard_normal((n, d), dtype=np.float32)
    store, tr = trainer_for(nat, x, 0, d, k)
    tr.init_rows(np.sort(rng.choice(n, k, replace=False)))
    sample = np.sort(rng.choice(n, 2000, replace=False))
    clusters = rng.choice(k, 16, replace=False)
    for step in range(2):
        cent = tr.centroids()
        a = tr.assignments()
        ref = R.kmeans_assign(x[sample], cent)
        moved = np.flatnonzero(a[sample] != ref)
        if len(moved):
            dist = R.pairwise_euclidean(x[sample[moved]], cent)
            got, want = dist[np.arange(len(moved)), a[sample[moved]]], dist[np.arange(len(moved)), ref[moved]]
            assert len(moved) <= 2 and np.allclose(got, want, rtol=1e-6), f"step {step}: {len(moved)} sampled rows in another cluster"
        counts, shift = tr.step(nat.KMEANS_IVF)
        assert counts.sum() == n and shift > 0
        np.testing.assert_array_equal(counts, np.bincount(a, minlength=k))
        tr.commit()
        new = tr.centroids()
        for c in clusters:
            members = np.flatnonzero(a == c)
            acc = np.zeros(d, np.float32)
            for r in members:                                  # sample order, one fp32 addition per row and column
                acc = acc + x[r]
            want = acc.copy()
            if len(members):
                want /= np.float64(len(members))               # fp32 row /= float64 count, cast back (ivf.py:186-187)
                np.testing.assert_array_equal(new[c], want, err_msg=f"step {step}, cluster {c}")
    tr.close(); store.close()


def test_kmeans_pp_over_identical_rows_raises_like_the_reference():
    """All rows equal: after the first centroid every weight is 0 and the reference's rng.choice(p=0/0) raises
    ValueError("probabilities contain NaN") (index/ivf.py:224); so does the device path, instead of picking a row."""
    from vectordb_amd.index.ivf import KMeans
    x = np.ones((500, 8), dtype=np.float32)
    with pytest.raises(ValueError, match="probabilities contain NaN"):
        KMeans(n_clusters=4, n_iter=3, seed=1).fit(x)
