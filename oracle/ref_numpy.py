"""NumPy restatement of the reference's search hot path (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The product package
never imports this module; it is the checker the HIP kernels are compared
with and the CPU baseline that ``bench.py`` times beside them.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  The restatement is written so that, wherever the
reference's own result is reproducible, this module produces the same bits:

* element-wise fp32 ops followed by ``np.sum(..., axis=1)`` over a
  C-contiguous matrix (direct-form Euclidean, Manhattan, row norms, PQ tables
  and PQ encoding) use NumPy's pairwise summation and are bit-identical to
  the reference's per-vector scalar functions;
* anything that goes through BLAS (``X @ Y.T``, ``C @ q``, ``np.dot``) is
  only reproducible to rounding: the reference itself does not pin those bits
  (its GEMM and GEMV paths already disagree), so those results carry the
  1e-4 relative tolerance.

Pinned against golden vectors produced by the real reference: see
``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------
# metric names
# ---------------------------------------------------------------------------

# distance/registry.py:105,119,133,148,162,176 -- alias -> canonical name
METRIC_ALIASES: Dict[str, str] = {
    "l2": "euclidean", "euclidean_distance": "euclidean",
    "l2_squared": "euclidean_squared", "sqeuclidean": "euclidean_squared",
    "cosine_distance": "cosine",
    "inner_product": "dot", "ip": "dot",
    "l1": "manhattan", "cityblock": "manhattan", "taxicab": "manhattan",
    "linf": "chebyshev", "chessboard": "chebyshev",
}


def canonical_metric(name: str) -> str:
    return METRIC_ALIASES.get(name, name)


# ---------------------------------------------------------------------------
# a1: one query against every row         distance/metrics.py:431-493
# ---------------------------------------------------------------------------

def _sq_row_norms(m: np.ndarray) -> np.ndarray:
    # metrics.py:456 -- np.sum(collection ** 2, axis=1): pairwise fp32 sum
    return np.sum(m ** 2, axis=1)


def query_distances(query: np.ndarray, collection: np.ndarray,
                    metric: str = "euclidean") -> np.ndarray:
    """distance/metrics.py:431-493 (identity form for L2, gemv for dot terms)."""
    q = query
    c = collection
    if metric in ("euclidean", "euclidean_squared"):
        # metrics.py:455-459 / 462-466
        sq = (np.dot(q, q) + _sq_row_norms(c)) - 2 * (c @ q)
        sq = np.maximum(sq, 0)
        return np.sqrt(sq) if metric == "euclidean" else sq
    if metric == "cosine":
        # metrics.py:469-481
        qn = np.linalg.norm(q)
        if qn < 1e-8:
            return np.ones(len(c))          # float64 ones, as the reference
        cn = np.linalg.norm(c, axis=1)
        cn = np.where(cn < 1e-8, 1.0, cn)
        sim = np.clip((c @ q) / (qn * cn), -1.0, 1.0)
        return 1.0 - sim
    if metric == "dot":
        return -(c @ q)                      # metrics.py:484
    if metric == "manhattan":
        return np.sum(np.abs(c - q), axis=1)  # metrics.py:487
    if metric == "chebyshev":
        return np.max(np.abs(c - q), axis=1)  # metrics.py:490
    raise ValueError(f"Unknown metric: {metric}")  # metrics.py:493


# ---------------------------------------------------------------------------
# a2-a5: pairwise matrices                distance/metrics.py:272-424
# ---------------------------------------------------------------------------

def pairwise_euclidean_squared(x: np.ndarray, y: Optional[np.ndarray] = None) -> np.ndarray:
    """metrics.py:309-333."""
    y = x if y is None else y
    xs = np.sum(x ** 2, axis=1, keepdims=True)
    ys = np.sum(y ** 2, axis=1, keepdims=True)
    return np.maximum(xs + ys.T - 2 * (x @ y.T), 0)


def pairwise_euclidean(x: np.ndarray, y: Optional[np.ndarray] = None) -> np.ndarray:
    """metrics.py:272-306."""
    return np.sqrt(pairwise_euclidean_squared(x, y))


def pairwise_cosine(x: np.ndarray, y: Optional[np.ndarray] = None, eps: float = 1e-8) -> np.ndarray:
    """metrics.py:336-373 (rows are normalised first, then one GEMM)."""
    y = x if y is None else y
    xn = np.linalg.norm(x, axis=1, keepdims=True)
    yn = np.linalg.norm(y, axis=1, keepdims=True)
    xn = np.where(xn < eps, 1.0, xn)
    yn = np.where(yn < eps, 1.0, yn)
    sim = np.clip((x / xn) @ (y / yn).T, -1.0, 1.0)
    return 1.0 - sim


def pairwise_dot(x: np.ndarray, y: Optional[np.ndarray] = None) -> np.ndarray:
    """metrics.py:376-391."""
    y = x if y is None else y
    return -(x @ y.T)


def pairwise_manhattan(x: np.ndarray, y: Optional[np.ndarray] = None) -> np.ndarray:
    """metrics.py:394-424 -- both branches reduce each (i, j) over d in the
    same pairwise order, so one row-wise formulation covers them."""
    y = x if y is None else y
    out = np.empty((len(x), len(y)), dtype=np.result_type(x.dtype, y.dtype))
    for i in range(len(x)):
        out[i] = np.sum(np.abs(x[i] - y), axis=1)
    return out


# ---------------------------------------------------------------------------
# a6: the per-pair scalar functions, evaluated for one vector against rows
#                                         distance/metrics.py:24-214
# ---------------------------------------------------------------------------

def rows_scalar_metric(a: np.ndarray, rows: np.ndarray, metric: str) -> np.ndarray:
    """``[fn(a, r) for r in rows]`` for fn = get_metric_fn(metric)
    (registry.py:305-319).  Returned as float64 because the scalar functions
    return Python floats."""
    metric = canonical_metric(metric)
    if len(rows) == 0:
        return np.zeros(0, dtype=np.float64)
    if metric == "euclidean":            # metrics.py:43
        return np.sqrt(np.sum((a - rows) ** 2, axis=1)).astype(np.float64)
    if metric == "manhattan":            # metrics.py:190
        return np.sum(np.abs(a - rows), axis=1).astype(np.float64)
    if metric == "chebyshev":            # metrics.py:214
        return np.max(np.abs(a - rows), axis=1).astype(np.float64)
    out = np.empty(len(rows), dtype=np.float64)
    if metric == "euclidean_squared":    # metrics.py:68-69
        for i, r in enumerate(rows):
            diff = a - r
            out[i] = float(np.dot(diff, diff))
        return out
    if metric == "dot":                  # metrics.py:166
        for i, r in enumerate(rows):
            out[i] = -float(np.dot(a, r))
        return out
    if metric in ("cosine", "angular"):  # metrics.py:92-100,123 / :535-538
        na = np.linalg.norm(a)
        for i, r in enumerate(rows):
            den = na * np.linalg.norm(r)
            sim = 0.0 if den < 1e-8 else float(np.dot(a, r) / den)
            out[i] = 1.0 - sim if metric == "cosine" else float(np.arccos(np.clip(sim, -1.0, 1.0)))
        return out
    if metric == "hamming":              # metrics.py:237
        return np.sum(a != rows, axis=1).astype(np.float64)
    raise KeyError(f"Unknown metric: '{metric}'")   # registry.py:241-245


# ---------------------------------------------------------------------------
# top-k selection with the deterministic tie rule
# ---------------------------------------------------------------------------

def topk_stable(dist: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k smallest values, ascending, ties to the lowest index.

    This is what ``list.sort`` gives the reference in Collection.search
    (core/collection.py:757), IVFIndex.search (index/ivf.py:631) and
    PQIndex.search (index/pq.py:300); FlatIndex's argpartition
    (index/flat.py:354-362) returns the same *set* but does not pin the order
    of exact ties.
    """
    k = min(int(k), len(dist))
    return np.argsort(dist, kind="stable")[:k].astype(np.int64)


def flat_search(query: np.ndarray, rows: np.ndarray, k: int, metric: str,
                mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """FlatIndex.search without the result objects (index/flat.py:319-362,
    396-413): distances by a1, optional boolean row mask standing for
    ``filter_fn``, top-k ascending.  Returns (row indices int64, distances)."""
    d = query_distances(query, rows, metric)
    if mask is not None:
        keep = np.flatnonzero(mask)
        order = topk_stable(d[keep], k)
        sel = keep[order]
    else:
        sel = topk_stable(d, k)
    return sel.astype(np.int64), d[sel]


def flat_range_search(query: np.ndarray, rows: np.ndarray, radius: float, metric: str,
                      mask: Optional[np.ndarray] = None, max_results: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """FlatIndex.range_search (index/flat.py:501-572): rows inside the radius (and the filter) are
    collected in insertion order until ``max_results`` of them are found (:545-567; a max_results of
    0 or None means no limit), then sorted by distance with a stable sort (:570).  Returns
    (row indices int64, distances)."""
    d = query_distances(query, rows, metric)
    inside = ~(d > radius)                                   # :546 skips rows with distance > radius
    if mask is not None:
        inside &= np.asarray(mask, dtype=bool)
    sel = np.flatnonzero(inside)
    if max_results:
        sel = sel[:max_results]
    order = np.argsort(d[sel].astype(np.float64), kind="stable")
    sel = sel[order]
    return sel.astype(np.int64), d[sel]


def simd_batch(metric: str, query: np.ndarray, vectors: np.ndarray) -> np.ndarray:
    """The functions get_optimized_distance_fn returns when Numba is absent (distance/simd.py:122-146,
    170-176): direct-form L2, the +1e-10 cosine, the POSITIVE dot product, Manhattan."""
    if metric == "euclidean":
        diff = vectors - query
        return np.sqrt(np.sum(diff * diff, axis=1))
    if metric == "cosine":
        qn = np.linalg.norm(query)
        vn = np.linalg.norm(vectors, axis=1)
        return 1.0 - np.dot(vectors, query) / (qn * vn + 1e-10)
    if metric == "dot":
        return np.dot(vectors, query)
    if metric == "manhattan":
        return np.sum(np.abs(vectors - query), axis=1)
    raise ValueError(f"Unknown metric: {metric}")            # simd.py:178-179


def flat_scores(dist: np.ndarray) -> np.ndarray:
    """index/flat.py:370-373 -- score = 1/(1+d) on Python floats."""
    return np.array([1.0 / (1.0 + float(x)) for x in dist], dtype=np.float64)


def collection_scores(dist: np.ndarray) -> np.ndarray:
    """core/collection.py:762-771 -- score relative to the k-th distance.
    ``distance`` and ``max_distance`` are NumPy scalars there, so the
    arithmetic stays in the array's dtype before the final ``float()``."""
    if len(dist) == 0:
        return np.zeros(0, dtype=np.float64)
    dmax = dist[-1]
    out = np.empty(len(dist), dtype=np.float64)
    for i, d in enumerate(dist):
        out[i] = float(1.0 - (d / (dmax + 1e-8))) if dmax > 0 else 1.0
    return out


def collection_search(query: np.ndarray, rows: np.ndarray, ids: Sequence[str], k: int,
                      metric: str = "euclidean") -> List[Tuple[str, float, float]]:
    """Collection.search without a filter, written the way the reference spends
    its time (core/collection.py:750-780): a1 over the whole matrix, a Python
    list of (id, distance) pairs, a stable ``list.sort`` on the distance, the
    first k, and the score of each against the k-th distance.  Returns
    (id, distance, score) triples."""
    distances = query_distances(query, rows, metric)
    id_distances = list(zip([ids[i] for i in range(len(distances))], distances))   # collection.py:751-754
    id_distances.sort(key=lambda x: x[1])                                           # collection.py:757
    top_k = id_distances[:k]
    max_distance = top_k[-1][1] if top_k else 1.0                                   # collection.py:762
    out = []
    for id_, distance in top_k:
        score = 1.0 - (distance / (max_distance + 1e-8)) if max_distance > 0 else 1.0   # collection.py:768-771
        out.append((id_, float(distance), float(score)))
    return out


def batch_top_k(query: np.ndarray, rows: np.ndarray, k: int, metric: str = "euclidean",
                chunk_size: int = 10000) -> Tuple[np.ndarray, np.ndarray]:
    """BatchDistanceCalculator.top_k (distance/batch.py:136-220).  Both of the
    reference's branches return the k smallest sorted ascending; the chunked
    branch sorts (distance, index) tuples (batch.py:212), the other leaves tie
    order to argpartition.  The restatement uses (distance, index) order for
    both."""
    n = len(rows)
    k = min(k, n)
    if n <= chunk_size:
        d = query_distances(query, rows, metric)
    else:
        d = np.zeros(n, dtype=np.float32)
        for s in range(0, n, chunk_size):
            d[s:s + chunk_size] = query_distances(query, rows[s:s + chunk_size], metric)
    idx = topk_stable(d, k)
    return idx, d[idx].astype(np.float32)


# ---------------------------------------------------------------------------
# a10: k-means                             index/ivf.py:102-237
# ---------------------------------------------------------------------------

def kmeans_assign(x: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """ivf.py:228-231 -- argmin is taken after clamp and sqrt."""
    return np.argmin(pairwise_euclidean(x, centroids), axis=1)


def kmeans_pp_init(x: np.ndarray, k: int, rng: np.random.RandomState) -> np.ndarray:
    """ivf.py:204-226.  Draw order: randint(n), then k-1 x choice(n, p)."""
    n, d = x.shape
    cent = np.zeros((k, d), dtype=x.dtype)
    cent[0] = x[rng.randint(n)]
    for j in range(1, k):
        d2 = np.min(pairwise_euclidean(x, cent[:j]) ** 2, axis=1)
        cent[j] = x[rng.choice(n, p=d2 / d2.sum())]
    return cent


def kmeans_update(x: np.ndarray, assign: np.ndarray, k: int,
                  rng: np.random.RandomState, prev: np.ndarray) -> np.ndarray:
    """ivf.py:177-190.  The reference adds rows into fp32 sums one sample at a
    time in sample order; ``np.add.at`` applies the same additions in the same
    order.  The division runs in float64 (fp32 row /= float64 count) and is
    cast back; empty clusters take ``x[rng.randint(n)]`` in ascending cluster
    order."""
    new = np.zeros_like(prev)
    np.add.at(new, assign, x)
    counts = np.bincount(assign, minlength=k).astype(np.float64)
    for c in range(k):
        if counts[c] > 0:
            new[c] /= counts[c]
        else:
            new[c] = x[rng.randint(len(x))]
    return new


def kmeans_fit(x: np.ndarray, k: int, n_iter: int = 20, seed: Optional[int] = None) -> np.ndarray:
    """KMeans.fit with n_redo=1 (ivf.py:126-202)."""
    n = len(x)
    if n < k:
        return x.copy()                                    # ivf.py:138-141
    rng = np.random.RandomState(seed)
    cent = kmeans_pp_init(x, k, rng)
    for _ in range(n_iter):
        new = kmeans_update(x, kmeans_assign(x, cent), k, rng, cent)
        shift = np.sum((new - cent) ** 2)                  # ivf.py:193
        cent = new
        if shift < 1e-6:                                   # ivf.py:199
            break
    return cent


# ---------------------------------------------------------------------------
# a11-a12: IVF                              index/ivf.py:385-648
# ---------------------------------------------------------------------------

def ivf_assign_batch(x: np.ndarray, centroids: np.ndarray, metric: str) -> np.ndarray:
    """IVFIndex._find_nearest_centroids_batch (ivf.py:553-565)."""
    if metric == "euclidean":
        d = pairwise_euclidean(x, centroids)
    elif metric == "cosine":
        d = pairwise_cosine(x, centroids)
    else:
        d = np.array([rows_scalar_metric(v, centroids, metric) for v in x])
    return np.argmin(d, axis=1)


def ivf_assign_single(v: np.ndarray, centroids: np.ndarray, metric: str) -> int:
    """IVFIndex._find_nearest_centroid (ivf.py:546-551)."""
    return int(np.argmin(rows_scalar_metric(v, centroids, metric)))


def ivf_build_lists(assign: Sequence[int], n_lists: int) -> List[np.ndarray]:
    """Inverted lists as arrays of row numbers in insertion order (ivf.py:529-542)."""
    assign = np.asarray(assign)
    return [np.flatnonzero(assign == c).astype(np.int64) for c in range(n_lists)]


def ivf_probe_order(query: np.ndarray, centroids: np.ndarray, metric: str, n_probe: int) -> np.ndarray:
    """ivf.py:611-614 -- ascending centroid distance; equal distances resolved
    to the lower centroid number."""
    d = rows_scalar_metric(query, centroids, metric)
    return np.argsort(d, kind="stable")[:n_probe]


def ivf_search(query: np.ndarray, rows: np.ndarray, centroids: np.ndarray,
               lists: List[np.ndarray], k: int, n_probe: int, metric: str,
               mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """IVFIndex.search (ivf.py:607-632): probe the n_probe nearest lists in
    order, direct-form distance to every (unmasked) member, stable sort.  Ties
    keep scan order (probe rank, then position in the list).  Returns (row
    numbers, float64 distances)."""
    n_probe = min(n_probe, len(centroids))
    probes = ivf_probe_order(query, centroids, metric, n_probe)
    cand = [lists[c] for c in probes]
    cand = np.concatenate(cand) if cand else np.zeros(0, dtype=np.int64)
    if mask is not None:
        cand = cand[mask[cand]]
    d = rows_scalar_metric(query, rows[cand], metric)
    order = np.argsort(d, kind="stable")[:k]
    return cand[order], d[order]


# ---------------------------------------------------------------------------
# a13-a15: PQ                               index/pq.py:115-318
# ---------------------------------------------------------------------------

def pq_train_subquantizer(sub: np.ndarray, ksub: int, n_iter: int,
                          rng: np.random.RandomState) -> np.ndarray:
    """PQIndex._train_subquantizer (pq.py:144-180).  After the first update the
    centroids are float64 (fp32 sums divided by float64 counts), so distances
    and sums are float64 from the second iteration on, exactly as there."""
    n = len(sub)
    k = min(ksub, n)
    cent = sub[rng.choice(n, k, replace=False)].copy()
    for _ in range(n_iter):
        d = (np.sum(sub ** 2, axis=1, keepdims=True)
             + np.sum(cent ** 2, axis=1, keepdims=True).T - 2 * sub @ cent.T)   # pq.py:182-186
        a = np.argmin(d, axis=1)
        sums = np.zeros_like(cent)
        np.add.at(sums, a, sub)                       # sequential, sample order (pq.py:163-165)
        counts = np.maximum(np.bincount(a, minlength=k).astype(np.float64), 1)
        cent = sums / counts[:, None]
    if k < ksub:                                      # pq.py:172-178
        full = np.zeros((ksub, sub.shape[1]), dtype=np.float32)
        full[:k] = cent
        cent = full
    return cent


def pq_train(x: np.ndarray, m: int, n_bits: int = 8, n_iter: int = 20,
             seed: Optional[int] = None) -> np.ndarray:
    """PQIndex.train (pq.py:115-142): codebook (m, 2**n_bits, d/m) float32."""
    n, d = x.shape
    ds = d // m
    ksub = 2 ** n_bits
    rng = np.random.RandomState(seed)
    book = np.zeros((m, ksub, ds), dtype=np.float32)
    for j in range(m):
        book[j] = pq_train_subquantizer(x[:, j * ds:(j + 1) * ds], ksub, n_iter, rng)
    return book


def pq_encode(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """PQIndex._encode for a batch (pq.py:188-200): per sub-space, direct-form
    squared distance to each codeword, first minimum."""
    m, ksub, ds = codebook.shape
    codes = np.empty((len(x), m), dtype=np.uint8)
    for j in range(m):
        sub = x[:, j * ds:(j + 1) * ds]
        for s in range(0, len(x), 4096):
            blk = sub[s:s + 4096]
            d = np.sum((codebook[j][None, :, :] - blk[:, None, :]) ** 2, axis=2)
            codes[s:s + 4096, j] = np.argmin(d, axis=1)
    return codes


def pq_lut(query: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """pq.py:270-283 -- table[m, c] = sum((codebook[m, c] - q_m) ** 2)."""
    m, ksub, ds = codebook.shape
    lut = np.zeros((m, ksub), dtype=np.float32)
    for j in range(m):
        lut[j] = np.sum((codebook[j] - query[j * ds:(j + 1) * ds]) ** 2, axis=1)
    return lut


def pq_adc(codes: np.ndarray, lut: np.ndarray) -> np.ndarray:
    """pq.py:296-297 -- ``sum(table[m, code[m]] for m)`` starts from int 0 and
    adds fp32 scalars left to right, so the accumulation is sequential fp32 in
    m; then sqrt."""
    acc = lut[0, codes[:, 0]].copy()
    for j in range(1, lut.shape[0]):
        acc = acc + lut[j, codes[:, j]]
    return np.sqrt(acc)


def pq_search(query: np.ndarray, codebook: np.ndarray, codes: np.ndarray, k: int,
              mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """PQIndex.search (pq.py:268-301).  Returns (row numbers, fp32 distances);
    ties keep insertion order."""
    d = pq_adc(codes, pq_lut(query, codebook))
    if mask is not None:
        keep = np.flatnonzero(mask)
        sel = keep[topk_stable(d[keep], k)]
    else:
        sel = topk_stable(d, k)
    return sel, d[sel]


# ---------------------------------------------------------------------------
# f1: HybridIndex (IVF-PQ)                   index/hybrid.py:41-536
# ---------------------------------------------------------------------------

def hybrid_assign_coarse(x: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """hybrid.py:205-210 -- argmin of |x|^2 + |c|^2 - 2 x.c (no clamp, no sqrt; BLAS)."""
    xs = np.sum(x ** 2, axis=1, keepdims=True)
    cs = np.sum(centroids ** 2, axis=1, keepdims=True)
    return np.argmin(xs + cs.T - 2 * x @ centroids.T, axis=1)


def hybrid_train_subquantizer(sub: np.ndarray, ksub: int, n_iter: int, rng: np.random.RandomState) -> np.ndarray:
    """hybrid.py:177-203 -- random initial rows, Lloyd with identity-form assignment,
    sequential accumulation, empty clusters become zero rows (counts clamped to 1)."""
    n = len(sub)
    k = min(ksub, n)
    cent = sub[rng.choice(n, k, replace=False)].copy()
    for _ in range(n_iter):
        xs = np.sum(sub ** 2, axis=1, keepdims=True)
        cs = np.sum(cent ** 2, axis=1, keepdims=True)
        assign = np.argmin(xs + cs.T - 2 * sub @ cent.T, axis=1)
        new = np.zeros_like(cent)
        counts = np.zeros(k)
        for i, c in enumerate(assign):
            new[c] += sub[i]
            counts[c] += 1
        cent = new / np.maximum(counts, 1)[:, None]
    if k < ksub:
        full = np.zeros((ksub, sub.shape[1]), dtype=np.float32)
        full[:k] = cent
        cent = full
    return cent


def hybrid_train(x: np.ndarray, n_clusters: int, m: int, n_bits: int = 8, n_iter: int = 20,
                 seed: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """HybridIndex.train (hybrid.py:126-175).  The coarse k-means has its own
    RandomState(seed) (ivf.py:121); the sub-quantizers draw from a second
    RandomState(seed) created in __init__ (hybrid.py:111)."""
    cent = kmeans_fit(x, min(n_clusters, len(x)), n_iter=n_iter, seed=seed)
    assign = hybrid_assign_coarse(x, cent)
    res = x - cent[assign]
    rng = np.random.RandomState(seed)
    ds = x.shape[1] // m
    book = np.zeros((m, 2 ** n_bits, ds), dtype=np.float32)
    for j in range(m):
        book[j] = hybrid_train_subquantizer(res[:, j * ds:(j + 1) * ds], 2 ** n_bits, n_iter, rng)
    return cent, book


def hybrid_encode(x: np.ndarray, centroids: np.ndarray, codebook: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """hybrid.py:266-270 -- coarse list, residual, PQ code (direct form, first minimum)."""
    lists = hybrid_assign_coarse(x, centroids)
    return lists, pq_encode(x - centroids[lists], codebook)


def hybrid_search(query: np.ndarray, centroids: np.ndarray, codebook: np.ndarray, codes: np.ndarray,
                  members: List[np.ndarray], k: int, n_probe: int,
                  mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """HybridIndex.search (hybrid.py:276-352).  `members[l]` lists the row numbers of
    list l in scan order; `codes[row]` is the row's PQ code.  Returns (row numbers,
    fp32 distances), ties in visiting order."""
    cs = np.sum(centroids ** 2, axis=1)
    coarse = np.sum(query ** 2) + cs - 2 * (query @ centroids.T)
    probes = np.argsort(coarse)[:n_probe]
    rows, dist = [], []
    for l in probes:
        lut = pq_lut(query - centroids[l], codebook)
        mem = members[l]
        if mask is not None:
            mem = mem[mask[mem]]
        if len(mem):
            rows.append(mem)
            dist.append(pq_adc(codes[mem], lut))
    if not rows:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float32)
    rows, dist = np.concatenate(rows), np.concatenate(dist)
    order = np.argsort(dist, kind="stable")[:k]
    return rows[order], dist[order]


# ---------------------------------------------------------------------------
# NumPy's fp32 pairwise summation, spelled out (documentation of the order the
# HIP kernels reproduce for row norms, direct-form distances and PQ tables)
# ---------------------------------------------------------------------------

def pairwise_sum_f32(a: np.ndarray) -> np.float32:
    """Bit-exact emulation of ``np.sum`` over a contiguous fp32 run."""
    a = np.asarray(a, dtype=np.float32)
    n = len(a)
    if n < 8:
        r = np.float32(0.0)
        for v in a:
            r = np.float32(r + v)
        return r
    if n <= 128:
        r = [np.float32(v) for v in a[:8]]
        i = 8
        while i + 8 <= n:
            for j in range(8):
                r[j] = np.float32(r[j] + a[i + j])
            i += 8
        res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3]))
                         + np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
        while i < n:
            res = np.float32(res + a[i])
            i += 1
        return res
    h = (n // 2) - ((n // 2) % 8)
    return np.float32(pairwise_sum_f32(a[:h]) + pairwise_sum_f32(a[h:]))
