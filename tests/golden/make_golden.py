#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the real reference.

Run in the build container only (it needs the reference checkout):

    python tests/golden/make_golden.py [/root/reference] [--only-round2]

The reference's ``vectordb/index/__init__.py`` uses ``Dict``/``Any`` without
importing them, so the two names are put into ``builtins`` before the import;
nothing in the reference is modified.  The outputs are plain ``.npz`` files of
inputs (or the seeds they come from) and the reference's results; no reference
source is stored.
"""

import builtins
import os
import sys
import typing

import numpy as np

_paths = [a for a in sys.argv[1:] if not a.startswith("--")]
REF = _paths[0] if _paths else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

builtins.Dict = typing.Dict
builtins.Any = typing.Any
sys.path.insert(0, REF)

from vectordb.distance import metrics as M                      # noqa: E402
from vectordb.distance.batch import BatchDistanceCalculator     # noqa: E402
from vectordb.distance.registry import get_metric_fn            # noqa: E402
from vectordb.index.flat import FlatIndex                       # noqa: E402
from vectordb.index.ivf import IVFIndex, KMeans                 # noqa: E402
from vectordb.index.pq import PQIndex                           # noqa: E402
from vectordb.index.hybrid import HybridIndex                   # noqa: E402
from vectordb.core.collection import Collection                 # noqa: E402


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def ids_of(n):
    return [f"v{i}" for i in range(n)]


def rows_of(results):
    return np.array([int(r.id[1:]) for r in results], dtype=np.int64)


# --------------------------------------------------------------------------
def gen_metrics():
    out = {}
    for seed in (0, 42, 123):
        rng = np.random.RandomState(seed)
        x = rng.randn(300, 32).astype(np.float32)
        q = rng.randn(32).astype(np.float32)
        x[7] = 0.0                                   # zero row (cosine guard)
        x[11] = q                                    # exact self match
        out[f"s{seed}_x"] = x
        out[f"s{seed}_q"] = q
        for metric in ("euclidean", "euclidean_squared", "cosine", "dot", "manhattan", "chebyshev"):
            out[f"s{seed}_qd_{metric}"] = M.query_distances(q, x, metric)
        out[f"s{seed}_qd_cosine_zeroq"] = M.query_distances(np.zeros(32, np.float32), x, "cosine")
        a, b = x[:20], x[40:75]
        out[f"s{seed}_pw_euclidean"] = M.pairwise_euclidean(a, b)
        out[f"s{seed}_pw_euclidean_squared"] = M.pairwise_euclidean_squared(a, b)
        out[f"s{seed}_pw_cosine"] = M.pairwise_cosine(a, b)
        out[f"s{seed}_pw_dot"] = M.pairwise_dot(a, b)
        out[f"s{seed}_pw_manhattan_small"] = M.pairwise_manhattan(a, b)          # n*m < 10000
        out[f"s{seed}_pw_manhattan_large"] = M.pairwise_manhattan(x[:100], x[100:200])  # n*m >= 10000
        out[f"s{seed}_pw_euclidean_self"] = M.pairwise_euclidean(a)
        # scalar functions, one vector against rows
        for metric in ("euclidean", "euclidean_squared", "cosine", "dot", "manhattan", "chebyshev"):
            fn = get_metric_fn(metric)
            out[f"s{seed}_scalar_{metric}"] = np.array([fn(q, r) for r in x[:64]], dtype=np.float64)
    # odd dimensions exercise the pairwise-sum tails (n<8, n%8!=0, n>128)
    rng = np.random.RandomState(5)
    for d in (3, 8, 13, 100, 130, 384, 768):
        x = rng.randn(40, d).astype(np.float32)
        q = rng.randn(d).astype(np.float32)
        out[f"d{d}_x"] = x
        out[f"d{d}_q"] = q
        out[f"d{d}_sqnorm"] = np.sum(x ** 2, axis=1)
        for metric in ("euclidean", "cosine", "dot", "manhattan", "chebyshev"):
            out[f"d{d}_qd_{metric}"] = M.query_distances(q, x, metric)
        fn = get_metric_fn("euclidean")
        out[f"d{d}_scalar_euclidean"] = np.array([fn(q, r) for r in x], dtype=np.float64)
    save("metrics.npz", **out)


# --------------------------------------------------------------------------
def gen_flat_config1():
    """BASELINE config 1: 10k x 128 fp32 euclidean, 100 queries, k=10."""
    np.random.seed(42)
    x = np.random.randn(10000, 128).astype(np.float32)
    np.random.seed(123)
    q = np.random.randn(100, 128).astype(np.float32)
    ids = ids_of(len(x))

    idx = FlatIndex(dimension=128, metric="euclidean")
    idx.add_batch(ids, x)
    rows = np.zeros((100, 10), dtype=np.int64)
    dist = np.zeros((100, 10), dtype=np.float64)
    score = np.zeros((100, 10), dtype=np.float64)
    for i in range(100):
        res = idx.search(q[i], k=10)
        rows[i], dist[i], score[i] = rows_of(res), [r.distance for r in res], [r.score for r in res]

    coll = Collection(name="c", dimension=128, metric="euclidean")
    coll.add_batch([{"id": ids[i], "vector": x[i], "metadata": {"g": int(i % 4)}} for i in range(len(x))])
    crow = np.zeros((100, 10), dtype=np.int64)
    cdist = np.zeros((100, 10), dtype=np.float64)
    cscore = np.zeros((100, 10), dtype=np.float64)
    for i in range(100):
        res = coll.search(q[i], k=10)
        crow[i], cdist[i], cscore[i] = rows_of(res), [r.distance for r in res], [r.score for r in res]
    frow = np.zeros((20, 10), dtype=np.int64)
    fdist = np.zeros((20, 10), dtype=np.float64)
    for i in range(20):
        res = coll.search(q[i], k=10, filter={"g": 1})
        frow[i], fdist[i] = rows_of(res), [r.distance for r in res]
    save("flat_config1.npz", flat_rows=rows, flat_dist=dist, flat_score=score,
         coll_rows=crow, coll_dist=cdist, coll_score=cscore,
         coll_filter_rows=frow, coll_filter_dist=fdist,
         x_checksum=np.array([x.sum(dtype=np.float64)]), q_checksum=np.array([q.sum(dtype=np.float64)]))


# --------------------------------------------------------------------------
def gen_flat_small():
    out = {}
    rng = np.random.RandomState(7)
    x = rng.randn(1500, 48).astype(np.float32)
    x[100] = x[3]            # duplicate rows -> exact ties
    x[900] = x[3]
    q = rng.randn(12, 48).astype(np.float32)
    q[0] = x[3]
    out["x"], out["q"] = x, q
    ids = ids_of(len(x))
    meta = [{"cat": int(i % 5)} for i in range(len(x))]
    keep = np.array([m["cat"] in (1, 3) for m in meta])
    out["mask"] = keep
    for metric in ("euclidean", "euclidean_squared", "cosine", "dot", "manhattan", "chebyshev"):
        idx = FlatIndex(dimension=48, metric=metric)
        idx.add_batch(ids, x, meta)
        for k in (1, 10, 37):
            rows = np.zeros((len(q), k), dtype=np.int64)
            dist = np.zeros((len(q), k), dtype=np.float64)
            for i in range(len(q)):
                res = idx.search(q[i], k=k)
                rows[i], dist[i] = rows_of(res), [r.distance for r in res]
            out[f"{metric}_k{k}_rows"], out[f"{metric}_k{k}_dist"] = rows, dist
        rows = np.zeros((len(q), 10), dtype=np.int64)
        dist = np.zeros((len(q), 10), dtype=np.float64)
        for i in range(len(q)):
            res = idx.search(q[i], k=10, filter_fn=lambda _id, m: m["cat"] in (1, 3))
            rows[i], dist[i] = rows_of(res), [r.distance for r in res]
        out[f"{metric}_filter_rows"], out[f"{metric}_filter_dist"] = rows, dist
    # normalize=True
    idx = FlatIndex(dimension=48, metric="euclidean", normalize=True)
    idx.add_batch(ids, x)
    rows = np.zeros((len(q), 10), dtype=np.int64)
    dist = np.zeros((len(q), 10), dtype=np.float64)
    for i in range(len(q)):
        res = idx.search(q[i], k=10)
        rows[i], dist[i] = rows_of(res), [r.distance for r in res]
    out["normalize_rows"], out["normalize_dist"] = rows, dist
    # k >= n//2 (full argsort branch) and k > n on a small index
    small = FlatIndex(dimension=48, metric="euclidean")
    small.add_batch(ids[:60], x[:60])
    for k in (30, 60, 65):
        res = small.search(q[1], k=k)
        out[f"small_k{k}_rows"] = rows_of(res)
        out[f"small_k{k}_dist"] = np.array([r.distance for r in res])
    # range_search
    idx_range = FlatIndex(dimension=48, metric="euclidean")
    idx_range.add_batch(ids, x)
    res = idx_range.range_search(q[2], radius=8.5)
    out["range_rows"], out["range_dist"] = rows_of(res), np.array([r.distance for r in res])
    save("flat_small.npz", **out)


# --------------------------------------------------------------------------
def gen_topk():
    out = {}
    rng = np.random.RandomState(11)
    x = rng.randn(12000, 24).astype(np.float32)
    q = rng.randn(24).astype(np.float32)
    out["x"], out["q"] = x, q
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        calc = BatchDistanceCalculator(metric=metric)
        for n in (50, 9000, 12000):                       # <= chunk, <= chunk, > chunk (heap path)
            for k in (1, 10, n, n + 5) if n == 50 else (1, 10):
                i, d = calc.top_k(q, x[:n], k)
                out[f"{metric}_n{n}_k{k}_idx"] = np.asarray(i, dtype=np.int64)
                out[f"{metric}_n{n}_k{k}_dist"] = np.asarray(d)
    save("topk.npz", **out)


# --------------------------------------------------------------------------
def gen_kmeans():
    out = {}
    rng = np.random.RandomState(3)
    x = rng.randn(600, 16).astype(np.float32)
    out["x"] = x
    km = KMeans(n_clusters=12, n_iter=20, seed=42)
    cent = km.fit(x)
    out["centroids"] = cent
    out["labels"] = km.predict(x)
    # one Lloyd step from given centroids (assignment + update), no empty clusters
    out["step_assign"] = km._assign(x, cent)
    # k-means++ draw sequence
    km2 = KMeans(n_clusters=12, n_iter=0, seed=42)
    out["pp_init"] = km2._init_centroids_pp(x)
    # n < k
    km3 = KMeans(n_clusters=50, seed=1)
    out["few_centroids"] = km3.fit(x[:20])
    save("kmeans.npz", **out)


# --------------------------------------------------------------------------
def gen_ivf():
    out = {}
    rng = np.random.RandomState(21)
    x = rng.randn(2000, 32).astype(np.float32)
    q = rng.randn(10, 32).astype(np.float32)
    out["x"], out["q"] = x, q
    ids = ids_of(len(x))
    meta = [{"cat": int(i % 3)} for i in range(len(x))]
    out["mask"] = np.array([m["cat"] == 1 for m in meta])
    for metric in ("euclidean", "cosine", "manhattan", "dot"):
        idx = IVFIndex(dimension=32, metric=metric, n_clusters=32, n_probe=8, seed=42)
        idx.train(x)
        out[f"{metric}_centroids"] = idx.centroids
        idx.add_batch(ids, x, meta)
        out[f"{metric}_assign_batch"] = np.array([idx._id_to_cluster[i] for i in ids], dtype=np.int64)
        out[f"{metric}_assign_single"] = np.array([idx.assign_cluster(v) for v in x[:300]], dtype=np.int64)
        for npb in (1, 8, 32):
            rows = np.full((len(q), 10), -1, dtype=np.int64)
            dist = np.full((len(q), 10), np.nan, dtype=np.float64)
            for i in range(len(q)):
                res = idx.search(q[i], k=10, n_probe=npb)
                rows[i, :len(res)], dist[i, :len(res)] = rows_of(res), [r.distance for r in res]
            out[f"{metric}_np{npb}_rows"], out[f"{metric}_np{npb}_dist"] = rows, dist
        rows = np.full((len(q), 10), -1, dtype=np.int64)
        dist = np.full((len(q), 10), np.nan, dtype=np.float64)
        for i in range(len(q)):
            res = idx.search(q[i], k=10, n_probe=8, filter_fn=lambda _id, m: m["cat"] == 1)
            rows[i, :len(res)], dist[i, :len(res)] = rows_of(res), [r.distance for r in res]
        out[f"{metric}_filter_rows"], out[f"{metric}_filter_dist"] = rows, dist
    save("ivf.npz", **out)


# --------------------------------------------------------------------------
def gen_pq():
    out = {}
    rng = np.random.RandomState(31)
    x = rng.randn(1200, 32).astype(np.float32)
    q = rng.randn(6, 32).astype(np.float32)
    out["x"], out["q"] = x, q
    ids = ids_of(len(x))
    meta = [{"cat": int(i % 4)} for i in range(len(x))]
    out["mask"] = np.array([m["cat"] == 2 for m in meta])

    idx = PQIndex(dimension=32, n_subvectors=4, n_bits=8, seed=42)
    idx.train(x)
    out["codebook"] = idx._codebook
    idx.add_batch(ids, x, meta)
    out["codes"] = np.stack([idx._codes[i] for i in ids])
    # full ADC distance list + top-k for each query
    rows = np.zeros((len(q), 10), dtype=np.int64)
    dist = np.zeros((len(q), 10), dtype=np.float32)
    frows = np.zeros((len(q), 10), dtype=np.int64)
    fdist = np.zeros((len(q), 10), dtype=np.float32)
    full = np.zeros((len(q), len(x)), dtype=np.float32)
    for i in range(len(q)):
        res = idx.search(q[i], k=10)
        rows[i], dist[i] = rows_of(res), [r.distance for r in res]
        res = idx.search(q[i], k=10, filter_fn=lambda _id, m: m["cat"] == 2)
        frows[i], fdist[i] = rows_of(res), [r.distance for r in res]
        res = idx.search(q[i], k=len(x))
        order = rows_of(res)
        full[i, order] = [r.distance for r in res]
    out["rows"], out["dist"], out["filter_rows"], out["filter_dist"], out["full_dist"] = rows, dist, frows, fdist, full

    # fewer training rows than codewords -> zero-padded codebook (pq.py:172-178)
    small = PQIndex(dimension=32, n_subvectors=8, n_bits=8, seed=1)
    small.train(x[:100])
    out["small_codebook"] = small._codebook
    small.add_batch(ids[:300], x[:300])
    out["small_codes"] = np.stack([small._codes[i] for i in ids[:300]])
    res = small.search(q[0], k=10)
    out["small_rows"], out["small_dist"] = rows_of(res), np.array([r.distance for r in res], dtype=np.float32)
    save("pq.npz", **out)


def gen_hybrid():
    """HybridIndex (IVF-PQ): trained quantizers, the (list, code) of every row, searches at two
    probe counts with and without a filter, and a removal that keeps list order."""
    out = {}
    rng = np.random.RandomState(77)
    x = rng.randn(1500, 32).astype(np.float32)
    q = rng.randn(5, 32).astype(np.float32)
    out["x"], out["q"] = x, q
    ids = ids_of(len(x))
    meta = [{"cat": int(i % 3)} for i in range(len(x))]
    out["mask"] = np.array([m["cat"] == 1 for m in meta])
    idx = HybridIndex(dimension=32, n_clusters=16, n_probe=4, n_subvectors=4, n_bits=8, seed=11)
    idx.config.n_iter = 5                         # the default 20 only makes the fixture slower to produce
    idx.train(x)
    out["n_iter"] = np.array(5)
    out["centroids"], out["codebook"] = idx._coarse_centroids, idx._pq_codebook
    idx.add_batch(ids, x, meta)
    lists = np.array([idx._id_to_cluster[i] for i in ids], dtype=np.int64)
    codes = np.zeros((len(x), 4), dtype=np.uint8)
    for l, entries in idx._inverted_lists.items():
        for id_, c, _ in entries:
            codes[int(id_[1:])] = c
    out["lists"], out["codes"] = lists, codes
    for tag, n_probe in (("p4", 4), ("p16", 16)):
        rows = np.full((len(q), 10), -1, dtype=np.int64)
        dist = np.full((len(q), 10), np.inf, dtype=np.float32)
        frows = np.full((len(q), 10), -1, dtype=np.int64)
        fdist = np.full((len(q), 10), np.inf, dtype=np.float32)
        for i in range(len(q)):
            res = idx.search(q[i], k=10, n_probe=n_probe)
            rows[i, :len(res)], dist[i, :len(res)] = rows_of(res), [r.distance for r in res]
            res = idx.search(q[i], k=10, n_probe=n_probe, filter_fn=lambda _id, m: m["cat"] == 1)
            frows[i, :len(res)], fdist[i, :len(res)] = rows_of(res), [r.distance for r in res]
        out[f"rows_{tag}"], out[f"dist_{tag}"], out[f"filter_rows_{tag}"], out[f"filter_dist_{tag}"] = rows, dist, frows, fdist
    for i in (3, 500, 501, 1499):
        idx.remove(ids[i])
    out["removed"] = np.array([3, 500, 501, 1499])
    res = idx.search(q[0], k=10, n_probe=16)
    out["rows_after_remove"], out["dist_after_remove"] = rows_of(res), np.array([r.distance for r in res], dtype=np.float32)
    save("hybrid.npz", **out)


def gen_round2():
    """Cases added later: range_search with max_results, pairwise_cosine with a caller's eps, the
    hamming / angular functions and an angular IVF index, the batch functions of distance/simd.py,
    the dtype of a zero cosine query, and k-means one step at a time (KMeans, the PQ and the
    IVF-PQ sub-quantizers)."""
    from vectordb.distance.simd import get_optimized_distance_fn
    out = {}
    rng = np.random.RandomState(5)
    x = rng.randn(1500, 24).astype(np.float32)
    x[40] = x[7]
    x[1200] = x[7]
    q = rng.randn(4, 24).astype(np.float32)
    q[0] = x[7] + np.float32(0.01)
    out["x"], out["q"] = x, q
    ids = ids_of(len(x))
    meta = [{"cat": int(i % 4)} for i in range(len(x))]
    out["mask"] = np.array([m["cat"] != 2 for m in meta])
    radii = {"euclidean": 6.2, "manhattan": 24.5, "cosine": 0.8, "dot": 1.0}
    for metric, radius in radii.items():
        idx = FlatIndex(dimension=24, metric=metric)
        idx.add_batch(ids, x, meta)
        out[f"range_{metric}_radius"] = np.array(radius)
        for tag, kw in (("all", {}), ("max5", {"max_results": 5}), ("max40", {"max_results": 40}),
                        ("filter_max10", {"max_results": 10, "filter_fn": lambda _id, m: m["cat"] != 2})):
            res = idx.range_search(q[0] if metric != "dot" else q[1], radius=radius, **kw)
            out[f"range_{metric}_{tag}_rows"] = rows_of(res)
            out[f"range_{metric}_{tag}_dist"] = np.array([r.distance for r in res], dtype=np.float64)
    # pairwise_cosine with eps: rows whose norm lies between the default 1e-8 and the caller's eps
    y = x[:200].copy()
    y[3] *= np.float32(1e-4)
    y[11] *= np.float32(1e-6)
    y[50] = 0.0
    xs = x[300:340].copy()
    xs[5] *= np.float32(1e-5)
    out["eps_x"], out["eps_y"] = xs, y
    for tag, eps in (("default", 1e-8), ("1e-3", 1e-3), ("1e-2", 1e-2)):
        out[f"pairwise_cosine_eps_{tag}"] = M.pairwise_cosine(xs, y, eps=eps)
    # hamming / angular through the registry, the way IVFIndex reaches them
    xi = rng.randint(0, 3, size=(300, 24)).astype(np.float32)
    qi = rng.randint(0, 3, size=24).astype(np.float32)
    out["ham_x"], out["ham_q"] = xi, qi
    out["hamming"] = np.array([get_metric_fn("hamming")(qi, v) for v in xi], dtype=np.float64)
    out["angular"] = np.array([get_metric_fn("angular")(q[2], v) for v in x[:400]], dtype=np.float64)
    idx = IVFIndex(dimension=24, metric="angular", n_clusters=16, n_probe=4, seed=3)
    idx.train(x)
    idx.add_batch(ids, x, meta)
    out["angular_ivf_centroids"] = idx.centroids
    out["angular_ivf_assign"] = np.array([idx._id_to_cluster[i] for i in ids], dtype=np.int64)
    rows = np.full((len(q), 10), -1, dtype=np.int64)
    dist = np.full((len(q), 10), np.nan, dtype=np.float64)
    for i in range(len(q)):
        res = idx.search(q[i], k=10, n_probe=4)
        rows[i, :len(res)], dist[i, :len(res)] = rows_of(res), [r.distance for r in res]
    out["angular_ivf_rows"], out["angular_ivf_dist"] = rows, dist
    # distance/simd.py
    for metric in ("euclidean", "cosine", "dot", "manhattan"):
        out[f"simd_{metric}"] = np.asarray(get_optimized_distance_fn(metric)(q[3], x[:500]))
    # a zero cosine query: dtype of the result (metrics.py:472-473)
    z = M.query_distances(np.zeros(24, dtype=np.float32), x[:10], "cosine")
    out["zero_query_cosine"] = z
    out["zero_query_cosine_dtype"] = np.array(str(z.dtype))
    # ---- k-means, one step at a time
    xs16 = rng.randn(900, 16).astype(np.float32)
    out["km_x"] = xs16
    for steps in (1, 2):
        km = KMeans(n_clusters=12, n_iter=steps, seed=42)
        out[f"km_centroids_after_{steps}"] = km._fit_once(xs16)
    out["km_pp_init"] = KMeans(n_clusters=12, n_iter=0, seed=42)._init_centroids_pp(xs16)
    out["km_assign_of_pp_init"] = KMeans(n_clusters=12, seed=42)._assign(xs16, out["km_pp_init"])
    # PQ sub-quantizer (pq.py:144-180): centroids after 1, 2, 3 and 20 steps from the same seeded start, and the
    # assignments those centroids give (fp32 at the first step, float64 afterwards)
    sub = rng.randn(1100, 8).astype(np.float32)
    out["pq_sub"] = sub
    out["pq_init_rows"] = PQIndex(dimension=8, n_subvectors=1, n_bits=6, seed=9)._rng.choice(len(sub), 64, replace=False)
    for steps in (1, 2, 3, 20):
        p = PQIndex(dimension=8, n_subvectors=1, n_bits=6, seed=9)
        p.config.n_iter = steps
        cent = p._train_subquantizer(sub)
        out[f"pq_centroids_after_{steps}"] = np.asarray(cent)
        out[f"pq_assign_from_{steps}"] = np.argmin(p._pairwise_l2(sub, cent), axis=1)
    out["pq_assign_from_0"] = np.argmin(PQIndex(dimension=8, n_subvectors=1, n_bits=6)._pairwise_l2(
        sub, sub[out["pq_init_rows"]].copy()), axis=1)
    # IVF-PQ sub-quantizer (hybrid.py:174-203)
    out["hy_init_rows"] = HybridIndex(dimension=8, n_clusters=4, n_subvectors=1, n_bits=6, seed=9)._rng.choice(
        len(sub), 64, replace=False)
    for steps in (1, 2, 5):
        h = HybridIndex(dimension=8, n_clusters=4, n_subvectors=1, n_bits=6, seed=9)
        h.config.n_iter = steps
        out[f"hy_centroids_after_{steps}"] = np.asarray(h._train_subquantizer(sub))
    save("round2.npz", **out)


if __name__ == "__main__":
    if "--only-round2" in sys.argv:
        gen_round2()
        sys.exit(0)
    gen_metrics()
    gen_flat_config1()
    gen_flat_small()
    gen_topk()
    gen_kmeans()
    gen_ivf()
    gen_pq()
    gen_hybrid()
    gen_round2()
