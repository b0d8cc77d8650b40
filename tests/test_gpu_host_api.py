"""The Python host layer (vectordb_amd.distance / .index / .core) behaves like
the reference's packages of the same names: the checks restate the
reference's own unit tests (tests/unit/test_metrics.py, test_flat_index.py,
test_ivf_index.py, test_collection.py) and compare with the golden vectors the
reference produced.  Everything computes on the GPU through the C ABI."""

import numpy as np
import pytest

from oracle import ref_numpy as R
from conftest import load_golden, same_ids_or_near_ties

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _init(native):
    return native


def ids_of(n):
    return [f"v{i}" for i in range(n)]


def rows_of(results):
    return np.array([int(r.id[1:]) for r in results], dtype=np.int64)


# ================================================================= distance

def test_metric_known_answers():
    # tests/unit/test_metrics.py:46-50,80-92,116-132,153-157,174-178
    from vectordb_amd import distance as D
    z, p = np.array([0.0, 0.0]), np.array([3.0, 4.0])
    assert D.euclidean(z, p) == pytest.approx(5.0)
    assert D.euclidean_squared(z, p) == pytest.approx(25.0)
    assert D.manhattan(z, p) == pytest.approx(7.0)
    assert D.chebyshev(z, p) == pytest.approx(4.0)
    e0, e1 = np.array([1.0, 0.0]), np.array([0.0, 1.0])
    assert D.cosine_distance(e0, e0) == pytest.approx(0.0)
    assert D.cosine_distance(e0, e1) == pytest.approx(1.0)
    assert D.cosine_distance(e0, -e0) == pytest.approx(2.0)
    assert D.cosine_similarity(np.array([1.0, 2.0, 3.0]), np.array([2.0, 4.0, 6.0])) == pytest.approx(1.0, abs=1e-6)
    assert D.dot_product(np.array([1.0, 2.0, 3.0]), np.array([4.0, 5.0, 6.0])) == pytest.approx(32.0)
    assert D.negative_dot_product(np.array([1.0, 2.0]), np.array([3.0, 4.0])) == pytest.approx(-11.0)
    a = np.array([1.0, 2.0, 3.0])
    assert D.euclidean(a, a) == 0.0 and D.manhattan(a, a) == 0.0


def test_query_distances_and_pairwise():
    # tests/unit/test_metrics.py:199-288
    from vectordb_amd import distance as D
    q = np.zeros(3)
    np.testing.assert_allclose(D.query_distances(q, np.diag([1.0, 2.0, 3.0]), "euclidean"), [1.0, 2.0, 3.0], atol=1e-6)
    e0 = np.array([1.0, 0.0])
    np.testing.assert_allclose(D.query_distances(e0, np.array([[1.0, 0], [0, 1.0], [-1.0, 0]]), "cosine"), [0, 1, 2], atol=1e-6)
    rng = np.random.RandomState(0)
    x = rng.randn(10, 5).astype(np.float32)
    y = rng.randn(20, 5).astype(np.float32)
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        assert D.query_distances(x[0], y, metric).shape == (20,)
    d = D.pairwise_euclidean(x, y)
    assert d.shape == (10, 20)
    for i in range(3):
        for j in range(3):
            assert d[i, j] == pytest.approx(float(np.sqrt(np.sum((x[i] - y[j]) ** 2))), abs=1e-5)
    dx = D.pairwise_euclidean(x)
    np.testing.assert_allclose(dx, dx.T, atol=1e-5)
    np.testing.assert_allclose(np.diag(dx), 0, atol=5e-3)
    c = D.pairwise_cosine(x)
    np.testing.assert_allclose(np.diag(c), 0, atol=1e-6)
    assert (c >= -1e-6).all() and (c <= 2 + 1e-6).all()
    np.testing.assert_allclose(D.pairwise_dot(x), -(x @ x.T), rtol=RTOL, atol=1e-5)
    with pytest.raises(ValueError):
        D.query_distances(x[0], y, "nope")
    g = load_golden("metrics.npz")
    xs = g["s42_x"]
    np.testing.assert_array_equal(D.pairwise_manhattan(xs[:100], xs[100:200]), g["s42_pw_manhattan_large"])
    np.testing.assert_allclose(D.pairwise_euclidean(xs[:20], xs[40:75]), g["s42_pw_euclidean"], rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(D.pairwise_cosine(xs[:20], xs[40:75]), g["s42_pw_cosine"], rtol=RTOL, atol=2e-6)


def test_registry():
    # tests/unit/test_metrics.py:294-347
    from vectordb_amd import distance as D
    assert D.get_metric("l2").name == "euclidean" and D.get_metric("ip").name == "dot" and D.get_metric("l1").name == "manhattan"
    assert D.metric_exists("cosine_distance") and not D.metric_exists("nope")
    with pytest.raises(KeyError):
        D.get_metric("nope")
    assert set(["euclidean", "cosine", "dot", "manhattan"]) <= set(D.list_metrics())
    fn = D.get_metric_fn("manhattan")
    assert fn(np.array([0.0, 0.0]), np.array([3.0, 4.0])) == pytest.approx(7.0)
    D.register_metric("half_l1", lambda a, b: 0.5 * float(np.sum(np.abs(a - b))), aliases=["hl1"])
    assert D.get_metric_fn("hl1")(np.array([0.0]), np.array([4.0])) == 2.0
    calc = D.DistanceCalculator("cosine")
    assert calc.distance(np.array([1.0, 0.0]), np.array([0.0, 1.0])) == pytest.approx(1.0)
    assert calc.pairwise(np.eye(3, dtype=np.float32)).shape == (3, 3)


def test_batch_calculator_golden():
    # tests/unit/test_metrics.py:353-439 + golden outputs of BatchDistanceCalculator.top_k
    from vectordb_amd.distance import BatchDistanceCalculator, compute_top_k, find_nearest_neighbors
    g = load_golden("topk.npz")
    x, q = g["x"], g["q"]
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        calc = BatchDistanceCalculator(metric=metric)
        for n in (50, 9000, 12000):
            for k in ((1, 10, n, n + 5) if n == 50 else (1, 10)):
                idx, dist = calc.top_k(q, x[:n], k)
                ref_idx, ref_dist = g[f"{metric}_n{n}_k{k}_idx"], g[f"{metric}_n{n}_k{k}_dist"]
                assert idx.dtype == np.int64 and dist.dtype == np.float32 and len(idx) == min(k, n)
                assert (np.diff(dist) >= 0).all()
                full = R.query_distances(q, x[:n], metric)
                assert same_ids_or_near_ties(idx, ref_idx, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=1e-4)
                if metric == "manhattan":
                    np.testing.assert_array_equal(idx, ref_idx)
                    np.testing.assert_array_equal(dist, ref_dist)
                else:
                    np.testing.assert_allclose(dist, ref_dist, rtol=RTOL, atol=2e-4)
    idx, dist = compute_top_k(q, x[:500], 7, "euclidean")
    assert len(idx) == 7
    nn_i, nn_d = find_nearest_neighbors(x[:200], 3, "manhattan")
    for i in range(0, 200, 37):
        d = R.query_distances(x[i], x[:200], "manhattan")
        d[i] = np.inf
        np.testing.assert_array_equal(nn_i[i], np.argsort(d, kind="stable")[:3])
    with pytest.raises(KeyError):
        BatchDistanceCalculator(metric="nope")


def test_gpu_distance_calculator():
    from vectordb_amd.distance import gpu as G
    assert G.HAS_GPU and G.get_gpu_info()["available"]
    rng = np.random.RandomState(1)
    v = rng.randn(300, 20).astype(np.float32)
    q = rng.randn(20).astype(np.float32)
    calc = G.GPUDistanceCalculator("euclidean")
    assert calc.is_gpu_available
    np.testing.assert_array_equal(calc.compute(q, v, force_gpu=True), np.sqrt(np.sum((v - q) ** 2, axis=1)))
    np.testing.assert_array_equal(G.GPUDistanceCalculator("manhattan").compute(q, v), np.sum(np.abs(v - q), axis=1))
    np.testing.assert_allclose(G.gpu_dot_batch(q, v), v @ q, rtol=RTOL, atol=1e-5)          # positive dot, gpu.py:198
    np.testing.assert_allclose(G.gpu_cosine_batch(q, v), R.query_distances(q, v, "cosine"), rtol=RTOL, atol=2e-6)
    with pytest.raises(ValueError):
        G.GPUDistanceCalculator("chebyshev").compute(q, v)


# ================================================================= FlatIndex

@pytest.fixture()
def flat128():
    from vectordb_amd.index import FlatIndex
    rng = np.random.RandomState(5)
    x = rng.randn(100, 128).astype(np.float32)
    idx = FlatIndex(dimension=128, metric="euclidean")
    idx.add_batch(ids_of(100), x, [{"i": i, "even": i % 2 == 0} for i in range(100)])
    return idx, x


def test_flat_basic_contract(flat128):
    # tests/unit/test_flat_index.py:55-67,138-217
    from vectordb_amd.index import FlatIndex, IndexType, create_index
    idx, x = flat128
    assert len(idx) == 100 and idx.index_type == IndexType.FLAT and idx.is_trained
    with pytest.raises(ValueError, match="already exists"):
        idx.add("v3", x[3])
    with pytest.raises(ValueError):
        idx.add("new", np.zeros(64, np.float32))
    with pytest.raises(ValueError):
        idx.add_batch(["a", "b"], np.zeros((3, 128), np.float32))
    res = idx.search(x[0], k=10)
    assert len(res) == 10 and res[0].id == "v0"
    d = [r.distance for r in res]
    assert d == sorted(d) and all(r.score == pytest.approx(1.0 / (1.0 + r.distance)) for r in res)
    assert res[0].metadata == {"i": 0, "even": True} and res[0].vector is None
    assert len(idx.search(x[0], k=500)) == 100                           # k > n capped
    assert FlatIndex(dimension=128).search(x[0]) == []                   # empty index
    res = idx.search(x[0], k=10, filter_fn=lambda id, m: m["even"])
    assert len(res) == 10 and all(r.metadata["even"] for r in res)
    res = idx.search(x[1], k=3, include_vectors=True, include_metadata=False)
    np.testing.assert_array_equal(res[0].vector, x[1])
    assert res[0].metadata is None
    batch = idx.search_batch(x[:7], k=5)
    for i in range(7):
        assert [r.id for r in batch[i]] == [r.id for r in idx.search(x[i], k=5)]
    res = idx.search_by_id("v5", k=5)
    assert "v5" not in [r.id for r in res]
    with pytest.raises(KeyError):
        idx.search_by_id("nope")
    cos = create_index("flat", 128, metric="cosine")
    cos.add_batch(ids_of(100), x)
    assert all(-1e-6 <= r.distance <= 2 + 1e-6 for r in cos.search(x[0], k=100))
    bad = FlatIndex(dimension=4, metric="nope")                         # checked at search time, as flat.py:331
    bad.add("a", np.zeros(4, np.float32))
    with pytest.raises(ValueError):
        bad.search(np.zeros(4, np.float32))


def test_flat_row_at_a_time_and_mutation(flat128):
    # tests/unit/test_flat_index.py: remove/clear/get/update/to_dict classes
    from vectordb_amd.index import FlatIndex
    idx, x = flat128
    one = FlatIndex(dimension=128, metric="manhattan", initial_capacity=4)
    for i in range(60):
        one.add(f"v{i}", x[i], {"i": i})
    assert [r.id for r in one.search(x[7], k=3)] == [f"v{j}" for j in R.flat_search(x[7], x[:60], 3, "manhattan")[0]]
    for i in range(60, 100):                                            # more rows after the first upload
        one.add(f"v{i}", x[i])
    assert rows_of(one.search(x[80], k=5)).tolist() == R.flat_search(x[80], x, 5, "manhattan")[0].tolist()
    assert one.remove("v10") and not one.remove("v10") and len(one) == 99
    assert "v10" not in one and one.get("v10") is None
    keep = np.array([i for i in range(100) if i != 10])
    got = [r.id for r in one.search(x[10], k=4)]
    want = [f"v{keep[j]}" for j in np.argsort(R.query_distances(x[10], x[keep], "manhattan"), kind="stable")[:4]]
    assert sorted(got) == sorted(want)
    vec, meta = one.get("v3")
    np.testing.assert_array_equal(vec, x[3])
    assert one.update("v3", vector=x[50], metadata={"new": 1}) and not one.update("zz", vector=x[0])
    assert {r.id for r in one.search(x[50], k=2)} == {"v3", "v50"}
    assert one.remove_batch(["v1", "v2", "nope"]) == 2
    data = one.to_dict()
    back = FlatIndex.from_dict(data)
    assert len(back) == len(one)
    assert [r.id for r in back.search(x[33], k=6)] == [r.id for r in one.search(x[33], k=6)]
    st = one.stats().to_dict()
    assert st["vector_count"] == 97 and st["index_type"] == "flat"
    assert one.clear() == 97 and len(one) == 0 and one.search(x[0]) == []
    one.add("again", x[0])
    assert one.search(x[0], k=1)[0].id == "again"


def test_flat_range_search_grid():
    # tests/unit/test_flat_index.py:223-257
    from vectordb_amd.index import FlatIndex
    idx = FlatIndex(dimension=2, metric="euclidean")
    pts = [(i, j) for i in range(5) for j in range(5)]
    idx.add_batch([f"p{i}_{j}" for i, j in pts], np.array(pts, dtype=np.float32))
    res = idx.range_search(np.array([2.0, 2.0]), radius=1.0)
    assert {r.id for r in res} == {"p2_2", "p1_2", "p3_2", "p2_1", "p2_3"} and res[0].id == "p2_2"
    assert len(idx.range_search(np.array([2.0, 2.0]), radius=0.5)) == 1
    assert len(idx.range_search(np.array([2.0, 2.0]), radius=10.0)) == 25
    assert len(idx.range_search(np.array([2.0, 2.0]), radius=10.0, max_results=7)) == 7


def test_flat_index_golden():
    from vectordb_amd.index import FlatIndex
    g = load_golden("flat_small.npz")
    x, q = g["x"], g["q"]
    meta = [{"cat": int(i % 5)} for i in range(len(x))]
    for metric in ("euclidean", "cosine", "dot", "manhattan", "chebyshev", "euclidean_squared"):
        idx = FlatIndex(dimension=48, metric=metric)
        idx.add_batch(ids_of(len(x)), x, meta)
        batch = idx.search_batch(q, k=10)
        for i in range(len(q)):
            res = idx.search(q[i], k=10)
            full = R.query_distances(q[i], x, metric)
            ref_rows, ref_dist = g[f"{metric}_k10_rows"][i], g[f"{metric}_k10_dist"][i]
            for got in (res, batch[i]):
                assert same_ids_or_near_ties(rows_of(got), ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=2e-3)
            if metric in ("manhattan", "chebyshev"):
                assert [r.distance for r in res] == ref_dist.tolist()
            res = idx.search(q[i], k=10, filter_fn=lambda _id, m: m["cat"] in (1, 3))
            assert same_ids_or_near_ties(rows_of(res), g[f"{metric}_filter_rows"][i], lambda r: full[r],
                                         float(g[f"{metric}_filter_dist"][i][-1]), rtol=RTOL, atol=2e-3)
    idx = FlatIndex(dimension=48, metric="euclidean", normalize=True)
    idx.add_batch(ids_of(len(x)), x)
    for i in range(len(q)):
        res = idx.search(q[i], k=10)
        assert set(rows_of(res)) == set(g["normalize_rows"][i])
        np.testing.assert_allclose([r.distance for r in res], g["normalize_dist"][i], rtol=RTOL, atol=2e-3)


# ================================================================= Collection

def test_collection_config1_golden():
    """BASELINE config 1 through Collection.search: ids, distances and the
    reference's k-th-distance-relative scores."""
    from vectordb_amd.core import Collection, VectorExistsError
    g = load_golden("flat_config1.npz")
    np.random.seed(42)
    x = np.random.randn(10000, 128).astype(np.float32)
    np.random.seed(123)
    q = np.random.randn(100, 128).astype(np.float32)
    coll = Collection("c", 128, metric="euclidean")
    out = coll.add_batch([{"id": f"v{i}", "vector": x[i], "metadata": {"g": int(i % 4)}} for i in range(len(x))])
    assert out["success_count"] == 10000 and len(coll) == 10000
    batch = coll.search_batch(q, k=10)
    for i in range(100):
        res = coll.search(q[i], k=10) if i < 25 else batch[i]
        full = R.query_distances(q[i], x, "euclidean")
        assert same_ids_or_near_ties(rows_of(res), g["coll_rows"][i], lambda r: full[r], float(g["coll_dist"][i][-1]), rtol=RTOL)
        np.testing.assert_allclose([r.distance for r in res], g["coll_dist"][i], rtol=RTOL)
        np.testing.assert_allclose([r.score for r in res], g["coll_score"][i], atol=2e-4)
        assert all(0.0 <= r.score <= 1.0 for r in res) and res[0].metadata == {"g": int(rows_of(res)[0] % 4)}
    for i in range(20):
        res = coll.search(q[i], k=10, filter={"g": 1})
        assert set(rows_of(res)) == set(g["coll_filter_rows"][i])
        np.testing.assert_allclose([r.distance for r in res], g["coll_filter_dist"][i], rtol=RTOL)
    assert coll.search(q[0], k=5, filter={"g": 99}) == [] and coll.search(q[0], k=5, filter={"zz": 1}) == []
    assert len(coll.search(q[0], k=3, filter={"g": {"$in": [0, 2]}})) == 3
    assert all(r.metadata["g"] >= 2 for r in coll.search(q[0], k=8, filter={"g": {"$gte": 2}}))
    with pytest.raises(VectorExistsError):
        coll.add("v1", x[1])
    with pytest.raises(ValueError):
        coll.search(np.zeros(64, np.float32))
    assert coll.delete("v0") and len(coll.search(x[0], k=1)) == 1 and coll.search(x[0], k=1)[0].id != "v0"


# ================================================================= KMeans / IVF

def test_kmeans_contract_and_golden():
    # tests/unit/test_ivf_index.py:17-49
    from vectordb_amd.index import KMeans
    g = load_golden("kmeans.npz")
    x = g["x"]
    km = KMeans(n_clusters=12, n_iter=20, seed=42)
    cent = km.fit(x)
    assert cent.shape == (12, 16) and cent.dtype == np.float32
    labels = km.predict(x)
    assert labels.shape == (600,) and labels.min() >= 0 and labels.max() < 12
    np.testing.assert_array_equal(KMeans(n_clusters=12, n_iter=0, seed=42)._init_centroids_pp(x), g["pp_init"])
    np.testing.assert_allclose(cent, g["centroids"], rtol=1e-4, atol=1e-4)     # same seed -> same draws -> same fixed point
    np.testing.assert_array_equal(km._assign(x, g["centroids"]), g["labels"])
    few = KMeans(n_clusters=50, seed=1).fit(x[:20])
    np.testing.assert_array_equal(few, g["few_centroids"])
    with pytest.raises(RuntimeError):
        KMeans(4).predict(x)


def test_ivf_contract_and_recall():
    # tests/unit/test_ivf_index.py:79-95,161-190,253-302,397-438
    from vectordb_amd.index import FlatIndex, IVFIndex
    rng = np.random.RandomState(42)
    x = rng.randn(5000, 32).astype(np.float32)
    q = np.random.RandomState(123).randn(20, 32).astype(np.float32)
    ivf = IVFIndex(dimension=32, n_clusters=50, n_probe=20, seed=42)
    with pytest.raises(RuntimeError):
        ivf.add("a", x[0])
    with pytest.raises(RuntimeError):
        ivf.search(x[0])
    assert not ivf.is_trained()
    ivf.train(x)
    assert ivf.is_trained() and ivf.centroids.shape == (50, 32)
    assert ivf.search(x[0]) == []
    ivf.add_batch(ids_of(5000), x, [{"i": i} for i in range(5000)])
    with pytest.raises(ValueError, match="already exists"):
        ivf.add("v0", x[0])
    flat = FlatIndex(dimension=32)
    flat.add_batch(ids_of(5000), x)
    recalls = []
    for npb in (1, 5, 20, 50):
        hit = 0
        for i in range(20):
            truth = {r.id for r in flat.search(q[i], k=10)}
            hit += len(truth & {r.id for r in ivf.search(q[i], k=10, n_probe=npb)})
        recalls.append(hit / 200)
    assert recalls[2] >= 0.7 and recalls[3] >= 0.999
    assert all(b >= a - 1e-9 for a, b in zip(recalls, recalls[1:]))        # non-decreasing in n_probe
    res = ivf.search(q[0], k=10, filter_fn=lambda id, m: m["i"] % 2 == 0)
    assert res and all(int(r.id[1:]) % 2 == 0 for r in res)
    d = [r.distance for r in res]
    assert d == sorted(d)
    batch = ivf.search_batch(q, k=10)
    assert [r.id for r in batch[3]] == [r.id for r in ivf.search(q[3], k=10)]
    ids, dist = ivf.search_with_scores(q[0], k=5)
    assert len(ids) == 5 and dist == sorted(dist)
    assert ivf.remove("v7") and not ivf.contains("v7") and len(ivf) == 4999
    assert "v7" not in [r.id for r in ivf.search(x[7], k=5, n_probe=50)]
    back = IVFIndex.from_dict(ivf.to_dict())
    assert [r.id for r in back.search(q[1], k=10)] == [r.id for r in ivf.search(q[1], k=10)]
    st = ivf.stats().to_dict()
    assert st["n_clusters"] == 50 and st["distance_computations"] > 0
    info = ivf.get_cluster_info()
    assert info["total_vectors"] == 4999 and sum(c["size"] for c in info["cluster_sizes"]) == 4999
    c0 = ivf.assign_cluster(x[0])
    assert "v0" in ivf.get_vectors_in_cluster(c0) and ivf.get_centroid(c0).shape == (32,)
    ivf.set_n_probe(1000)
    assert ivf.n_probe == 50
    small = IVFIndex(dimension=32, n_clusters=100, n_probe=10)
    small.train(x[:30])
    assert small.n_clusters == 30 and small.n_probe == 10


@pytest.mark.parametrize("metric", ["euclidean", "cosine", "manhattan", "dot"])
def test_ivf_index_golden(metric):
    """Same centroids -> same list assignments and the same results at equal n_probe."""
    from vectordb_amd.index import IVFIndex
    g = load_golden("ivf.npz")
    x, q = g["x"], g["q"]
    idx = IVFIndex(dimension=32, metric=metric, n_clusters=32, n_probe=8, seed=42)
    idx._set_centroids(g[f"{metric}_centroids"])
    idx.add_batch(ids_of(len(x)), x, [{"cat": int(i % 3)} for i in range(len(x))])
    got = np.array([idx._id_to_cluster[i] for i in ids_of(len(x))])
    ref = g[f"{metric}_assign_batch"]
    moved = np.flatnonzero(got != ref)
    assert len(moved) <= (0 if metric == "manhattan" else 2)
    if len(moved):
        # a row may sit in another list than in the reference only when the reference's own distances to the two
        # centroids tie to rounding (argmin over BLAS-form values); verify that, then go on with the reference's
        # lists so that every search below is still compared with the fixture
        cent = g[f"{metric}_centroids"]
        for i in moved:
            d = (R.pairwise_euclidean(x[i:i + 1], cent)[0] if metric == "euclidean" else
                 R.pairwise_cosine(x[i:i + 1], cent)[0] if metric == "cosine" else
                 R.rows_scalar_metric(x[i], cent, metric))
            assert abs(float(d[got[i]]) - float(d[ref[i]])) <= 1e-5 * max(1.0, abs(float(d[ref[i]]))), (i, d[got[i]], d[ref[i]])
        idx = IVFIndex(dimension=32, metric=metric, n_clusters=32, n_probe=8, seed=42)
        idx._set_centroids(cent)
        for i in range(len(x)):
            idx._append(f"v{i}", x[i], {"cat": int(i % 3)}, int(ref[i]))
    exact = metric in ("euclidean", "manhattan")
    for npb in (1, 8, 32):
        for i in range(len(q)):
            res = idx.search(q[i], k=10, n_probe=npb)
            ref_rows, ref_dist = g[f"{metric}_np{npb}_rows"][i], g[f"{metric}_np{npb}_dist"][i]
            n = int((ref_rows >= 0).sum())
            if exact:
                assert rows_of(res).tolist() == ref_rows[:n].tolist()
                assert [r.distance for r in res] == ref_dist[:n].tolist()
            else:
                assert sorted(rows_of(res)) == sorted(ref_rows[:n])
                np.testing.assert_allclose([r.distance for r in res], ref_dist[:n], rtol=RTOL, atol=2e-6)
    for i in range(len(q)):
        res = idx.search(q[i], k=10, n_probe=8, filter_fn=lambda _id, m: m["cat"] == 1)
        ref_rows = g[f"{metric}_filter_rows"][i]
        n = int((ref_rows >= 0).sum())
        assert sorted(rows_of(res)) == sorted(ref_rows[:n])
    single = np.array([idx.assign_cluster(v) for v in x[:100]])
    assert (single != g[f"{metric}_assign_single"][:100]).sum() <= (0 if exact else 1)


def test_ivf_train_golden():
    from vectordb_amd.index import IVFIndex
    g = load_golden("ivf.npz")
    idx = IVFIndex(dimension=32, n_clusters=32, n_probe=8, seed=42)
    idx.train(g["x"])
    np.testing.assert_allclose(idx.centroids, g["euclidean_centroids"], rtol=1e-3, atol=1e-3)


# ================================================================= PQ

def test_pq_index_golden():
    from vectordb_amd.index import PQIndex
    g = load_golden("pq.npz")
    x, q = g["x"], g["q"]
    with pytest.raises(ValueError):
        PQIndex(dimension=30, n_subvectors=4)
    idx = PQIndex(dimension=32, n_subvectors=4, n_bits=8, seed=42)
    with pytest.raises(RuntimeError):
        idx.add("a", x[0])
    with pytest.raises(RuntimeError):
        idx.search(x[0])
    idx._set_codebook(g["codebook"])
    assert idx.search(x[0]) == []
    meta = [{"cat": int(i % 4)} for i in range(len(x))]
    idx.add_batch(ids_of(600), x[:600], meta[:600])
    for i in range(600, 640):
        idx.add(f"v{i}", x[i], meta[i])
    idx.add_batch(ids_of(len(x))[640:], x[640:], meta[640:])
    np.testing.assert_array_equal(np.stack([idx._codes[i] for i in ids_of(len(x))]), g["codes"])
    with pytest.raises(ValueError, match="already exists"):
        idx.add("v5", x[5])
    batch = idx.search_batch(q, k=10)
    for i in range(len(q)):
        res = idx.search(q[i], k=10)
        assert rows_of(res).tolist() == g["rows"][i].tolist() == rows_of(batch[i]).tolist()
        assert [r.distance for r in res] == g["dist"][i].tolist()
        assert isinstance(res[0].distance, np.float32)
        res = idx.search(q[i], k=10, filter_fn=lambda _id, m: m["cat"] == 2)
        assert rows_of(res).tolist() == g["filter_rows"][i].tolist()
        assert [r.distance for r in res] == g["filter_dist"][i].tolist()
    vec, _ = idx.get("v3")
    np.testing.assert_array_equal(vec, np.concatenate([g["codebook"][j, g["codes"][3, j]] for j in range(4)]))
    back = PQIndex.from_dict(idx.to_dict())
    assert rows_of(back.search(q[0], k=10)).tolist() == g["rows"][0].tolist()
    assert idx.remove("v0") and len(idx) == len(x) - 1
    ref_rows, ref_dist = R.pq_search(q[0], g["codebook"], g["codes"][1:], 10)
    res = idx.search(q[0], k=10)
    assert rows_of(res).tolist() == (ref_rows + 1).tolist() and [r.distance for r in res] == ref_dist.tolist()
    assert idx.stats().to_dict()["n_subvectors"] == 4


def test_pq_train_quality():
    """Training follows the reference's procedure (same RNG draws, 20 Lloyd
    steps); centroids are float32 between steps here and float64 there, so the
    check is on the codebook's quantisation error, not its bits."""
    from vectordb_amd.index import PQIndex
    g = load_golden("pq.npz")
    x = g["x"]
    idx = PQIndex(dimension=32, n_subvectors=4, n_bits=8, seed=42)
    idx.train(x)
    assert idx._codebook.shape == (4, 256, 8) and idx._codebook.dtype == np.float32

    def qerr(book):
        codes = R.pq_encode(x, book)
        rec = np.concatenate([book[j, codes[:, j]] for j in range(4)], axis=1)
        return float(np.mean(np.sum((x - rec) ** 2, axis=1)))
    assert qerr(idx._codebook) <= 1.02 * qerr(g["codebook"])
    small = PQIndex(dimension=32, n_subvectors=8, n_bits=8, seed=1)
    small.train(x[:100])
    assert not small._codebook[:, 100:].any()                            # zero padding beyond the trained codewords


def test_flat_index_bf16_storage_kwarg():
    """FlatIndex(storage="bf16"): same interface, device rows in bfloat16."""
    from vectordb_amd.index import FlatIndex
    rng = np.random.RandomState(3)
    x = rng.randn(3000, 48).astype(np.float32)
    ids = [f"v{i}" for i in range(len(x))]
    exact = FlatIndex(48, metric="cosine")
    half = FlatIndex(48, metric="cosine", storage="bf16")
    exact.add_batch(ids, x)
    half.add_batch(ids, x)
    hits = 0
    for qi in range(20):
        a = [r.id for r in exact.search(x[qi] + 0.05 * rng.randn(48).astype(np.float32), k=10)]
        b = [r.id for r in half.search(x[qi] + 0.05 * rng.randn(48).astype(np.float32), k=10)]
        assert b[0] == f"v{qi}" and a[0] == f"v{qi}"
        hits += 1
    assert hits == 20
    with pytest.raises(ValueError):
        FlatIndex(48, storage="fp8")


# ================================================================= HybridIndex (IVF-PQ)
def test_hybrid_index_matches_reference_outputs():
    """With the reference's trained quantizers: list, code and search results of the reference's own
    HybridIndex (tests/golden/hybrid.npz), through the index class; then removal, to_dict/from_dict."""
    from vectordb_amd.index import HybridIndex, IndexType, create_index
    g = load_golden("hybrid.npz")
    x, q = g["x"], g["q"]
    ids = [f"v{i}" for i in range(len(x))]
    meta = [{"cat": int(i % 3)} for i in range(len(x))]
    idx = create_index("ivf_pq", 32, n_clusters=16, n_probe=4, n_subvectors=4, n_bits=8, seed=11)
    assert isinstance(idx, HybridIndex) and idx.index_type == IndexType.IVF_PQ
    with pytest.raises(RuntimeError):
        idx.add("a", x[0])
    idx._set_quantizers(g["centroids"], g["codebook"])
    assert idx.add_batch(ids, x, meta) == len(x)
    with pytest.raises(ValueError):
        idx.add("v3", x[3])
    got_lists = np.array([idx._id_to_cluster[i] for i in ids])
    assert (got_lists != g["lists"]).sum() <= 2
    for tag, n_probe in (("p4", 4), ("p16", 16)):
        batch = idx.search_batch(q, k=10, n_probe=n_probe)
        for i in range(len(q)):
            res = idx.search(q[i], k=10, n_probe=n_probe)
            assert [int(r.id[1:]) for r in res] == g[f"rows_{tag}"][i].tolist()
            assert [r.distance for r in res] == g[f"dist_{tag}"][i].tolist()
            assert [r.id for r in batch[i]] == [r.id for r in res]
            res = idx.search(q[i], k=10, n_probe=n_probe, filter_fn=lambda _id, m: m["cat"] == 1)
            n = int((g[f"filter_rows_{tag}"][i] >= 0).sum())
            assert [int(r.id[1:]) for r in res] == g[f"filter_rows_{tag}"][i][:n].tolist()
    assert idx.search(q[0], k=3)[0].score == pytest.approx(1.0 / (1.0 + g["dist_p4"][0][0]))
    for r in g["removed"]:
        assert idx.remove(ids[r])
    assert not idx.remove("v3") and len(idx) == len(x) - 4
    res = idx.search(q[0], k=10, n_probe=16)
    assert [int(r.id[1:]) for r in res] == g["rows_after_remove"].tolist()
    back = HybridIndex.from_dict(idx.to_dict())
    res2 = back.search(q[0], k=10, n_probe=16)
    assert [r.id for r in res2] == [r.id for r in res] and [r.distance for r in res2] == [r.distance for r in res]
    vec, md = back.get("v7")
    assert vec.shape == (32,) and md == {"cat": 1}
    assert back.stats().index_type == "ivf_pq" and back.stats().extra["compression_ratio"] > 1


def test_hybrid_index_training_on_device():
    """train() on the device: the coarse centroids follow the reference's (same k-means++ draws; fp64
    sums instead of sequential fp32), the residual codebook quantises as well as the reference's."""
    from vectordb_amd.index import HybridIndex
    g = load_golden("hybrid.npz")
    x = g["x"]
    idx = HybridIndex(32, n_clusters=16, n_probe=4, n_subvectors=4, n_bits=8, seed=11)
    idx.config.n_iter = int(g["n_iter"])
    idx.train(x)
    np.testing.assert_allclose(idx._coarse_centroids, g["centroids"], rtol=1e-3, atol=1e-3)

    def quant_error(cent, book):
        lists, codes = R.hybrid_encode(x, cent, book)
        rec = cent[lists] + np.concatenate([book[j][codes[:, j]] for j in range(4)], axis=1)
        return float(np.mean(np.sum((x - rec) ** 2, axis=1)))
    assert quant_error(idx._coarse_centroids, idx._pq_codebook) <= 1.02 * quant_error(g["centroids"], g["codebook"])
    idx.add_batch([f"v{i}" for i in range(len(x))], x)
    hits = sum(idx.search(x[i], k=5, n_probe=16)[0].id == f"v{i}" for i in range(0, 200, 10))
    assert hits >= 14          # PQ distances are approximate: the row itself is usually, not always, first


# ================================================================= Collection(index_type=...) and index files

def test_collection_index_type_routes_through_the_index():
    """docs/indexing_guide.md:32-36,66-73 / examples/semantic_search.py:75-81: the
    documented ``index_type`` argument.  Results must equal the same index class
    used directly on the same rows; "flat" must equal the default exact scan."""
    from vectordb_amd.core import Collection
    from vectordb_amd.index import IVFIndex, PQIndex
    rng = np.random.RandomState(42)
    x = rng.randn(3000, 32).astype(np.float32)
    q = np.random.RandomState(123).randn(8, 32).astype(np.float32)
    items = [{"id": f"v{i}", "vector": x[i], "metadata": {"g": int(i % 3)}} for i in range(len(x))]

    exact = Collection("e", 32)
    exact.add_batch(items)
    flat = Collection("f", 32, index_type="flat")
    flat.add_batch(items)
    for i in range(8):
        a, b = exact.search(q[i], k=10), flat.search(q[i], k=10)
        assert [r.id for r in a] == [r.id for r in b]
        np.testing.assert_allclose([r.distance for r in a], [r.distance for r in b], rtol=RTOL)
        np.testing.assert_allclose([r.score for r in a], [r.score for r in b], atol=2e-4)

    coll = Collection("i", 32, index_type="ivf", n_clusters=40, n_probe=8, seed=42)
    coll.add_batch(items)
    ivf = IVFIndex(dimension=32, n_clusters=40, n_probe=8, seed=42)
    ivf.train(x)
    ivf.add_batch(ids_of(3000), x, [{"g": int(i % 3)} for i in range(3000)])
    batch = coll.search_batch(q, k=10)
    for i in range(8):
        want = ivf.search(q[i], k=10)
        got = coll.search(q[i], k=10) if i < 4 else batch[i]
        assert [r.id for r in got] == [r.id for r in want]
        assert [r.distance for r in got] == [float(np.float32(r.distance)) for r in want]
        assert abs(got[-1].score) < 1e-6 and all(0.0 <= r.score <= 1.0 for r in got)      # collection.py:762-771
    got = coll.search(q[0], k=10, filter={"g": 1})
    want = ivf.search(q[0], k=10, filter_fn=lambda id, m: m["g"] == 1)
    assert [r.id for r in got] == [r.id for r in want] and all(r.metadata["g"] == 1 for r in got)
    # writes after the index exists are applied to it
    coll.add("extra", q[0], {"g": 1})
    assert coll.search(q[0], k=1)[0].id == "extra"
    assert coll.delete("extra") and coll.search(q[0], k=1)[0].id != "extra"
    coll.add("v5", q[1], {"g": 2}, overwrite=True)
    assert coll.search(q[1], k=1)[0].id == "v5"

    pq = Collection("p", 32, index_params={"n_subvectors": 8, "n_bits": 8, "seed": 3}, index_type="pq")
    pq.add_batch(items)
    ref = PQIndex(dimension=32, n_subvectors=8, n_bits=8, seed=3)
    ref.train(x)
    ref.add_batch(ids_of(3000), x)
    assert [r.id for r in pq.search(q[2], k=10)] == [r.id for r in ref.search(q[2], k=10)]
    with pytest.raises(TypeError):
        Collection("bad", 32, n_clusters=4)
    assert Collection("empty", 32, index_type="ivf").search(q[0]) == []


def test_index_files_round_trip(tmp_path):
    """save_index / load_index: the same search results before and after, for every index
    class, with nothing pickled in the file (index/flat.py:791-840, ivf.py:897-972,
    pq.py:392-427, hybrid.py:456-520 are the dict forms the file stands for)."""
    from vectordb_amd.index import FlatIndex, IVFIndex, PQIndex, HybridIndex, save_index, load_index
    rng = np.random.RandomState(7)
    x = rng.randn(1500, 32).astype(np.float32)
    q = rng.randn(5, 32).astype(np.float32)
    meta = [{"i": i, "tag": f"t{i % 5}"} for i in range(len(x))]

    def build(kind):
        if kind == "flat":
            ix = FlatIndex(dimension=32, metric="cosine")
        elif kind == "ivf":
            ix = IVFIndex(dimension=32, n_clusters=20, n_probe=5, seed=1)
        elif kind == "pq":
            ix = PQIndex(dimension=32, n_subvectors=8, n_bits=8, seed=1)
        else:
            ix = HybridIndex(dimension=32, n_clusters=20, n_probe=5, n_subvectors=8, n_bits=8)
        if kind != "flat":
            ix.train(x)
        ix.add_batch(ids_of(len(x)), x, meta)
        ix.remove("v3")
        return ix

    for kind in ("flat", "ivf", "pq", "ivf_pq"):
        ix = build(kind)
        path = save_index(ix, str(tmp_path / kind))
        with np.load(path, allow_pickle=False) as z:              # readable without unpickling anything
            assert "small_json" in z.files
        back = load_index(path)
        assert type(back) is type(ix) and len(back) == len(ix) == len(x) - 1 and not back.contains("v3")
        for i in range(5):
            a, b = ix.search(q[i], k=10), back.search(q[i], k=10)
            assert [r.id for r in a] == [r.id for r in b]
            assert [r.distance for r in a] == [r.distance for r in b]
        f = lambda id, m: m["tag"] == "t2"
        assert [r.id for r in ix.search(q[0], k=5, filter_fn=f)] == [r.id for r in back.search(q[0], k=5, filter_fn=f)]
        assert back.search(q[0], k=1)[0].metadata == ix.search(q[0], k=1)[0].metadata
    # an untrained index saves and loads too
    empty = load_index(save_index(IVFIndex(dimension=32, n_clusters=4), str(tmp_path / "untrained")))
    assert not empty.is_trained() and len(empty) == 0
    with pytest.raises(TypeError):
        save_index(object(), str(tmp_path / "x"))


def test_flat_find_duplicates_and_sample():
    """index/flat.py:903-956, the cases of the reference's unit tests: an exact copy is the only pair
    under 1e-5; pairs come out with i < j in insertion order; sample(n, seed) draws through NumPy's
    global generator, so the same seed gives the reference's choice of rows."""
    from vectordb_amd.index import FlatIndex
    rng = np.random.RandomState(4)
    for metric in ("euclidean", "cosine", "manhattan"):
        ix = FlatIndex(dimension=10, metric=metric)
        v1 = rng.randn(10).astype(np.float32)
        ix.add("vec1", v1)
        ix.add("vec2", v1.copy())
        ix.add("vec3", rng.randn(10).astype(np.float32))
        dup = ix.find_duplicates(threshold=1e-5)
        assert [(a, b) for a, b, _ in dup] == [("vec1", "vec2")], metric
        assert dup[0][2] <= 1e-5
    # near-duplicates under a larger threshold, against the oracle's pair distances
    ix = FlatIndex(dimension=24, metric="euclidean")
    x = rng.randn(300, 24).astype(np.float32)
    x[100] = x[7] + 1e-3
    x[200] = x[7]
    x[250] = x[100]
    ix.add_batch([f"id{i}" for i in range(300)], x)
    dup = ix.find_duplicates(threshold=0.05)
    assert [(a, b) for a, b, _ in dup] == [("id7", "id100"), ("id7", "id200"), ("id7", "id250"), ("id100", "id200"),
                                           ("id100", "id250"), ("id200", "id250")]
    for a, b, d in dup:
        i, j = int(a[2:]), int(b[2:])
        assert abs(d - float(np.sqrt(np.sum((x[i] - x[j]) ** 2)))) <= 1e-6
    assert FlatIndex(dimension=4).find_duplicates() == []

    s1 = ix.sample(5, seed=42)
    np.random.seed(42)
    want = np.random.choice(300, 5, replace=False)
    assert [id for id, _, _ in s1] == [f"id{i}" for i in want]
    for (id, vec, meta), i in zip(s1, want):
        np.testing.assert_array_equal(vec, x[i])
        assert meta == {}
    assert len(ix.sample(1000)) == 300
