"""Distance functions of vectordb's ``distance.metrics`` module, computed by
the HIP kernels of libvdbcore.so.

Same names, arguments, return shapes and error behaviour as the reference
(distance/metrics.py).  Nothing here does distance arithmetic on the host: the
functions upload their arguments, run the streaming or matrix-core kernel and
return the result.  For data that is searched repeatedly use the index classes
(the rows stay resident in HBM there); these functions copy the collection on
every call, like the reference's own accelerator seam does
(distance/gpu.py:111-134).
"""

from __future__ import annotations

from typing import Optional

import threading

import numpy as np

from .. import _native as N

_QUERY_METRICS = {
    "euclidean": N.EUCLIDEAN,
    "euclidean_squared": N.EUCLIDEAN_SQ,
    "cosine": N.COSINE,
    "dot": N.DOT,
    "manhattan": N.MANHATTAN,
    "chebyshev": N.CHEBYSHEV,
}


def _matrix(a, name: str) -> np.ndarray:
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError(f"{name} must be 2D, got {a.ndim}D")
    return N.f32(a)


def _distances(queries: np.ndarray, rows: np.ndarray, code: int, cosine_eps: Optional[float] = None) -> np.ndarray:
    if rows.shape[0] == 0 or queries.shape[0] == 0:
        return np.zeros((queries.shape[0], rows.shape[0]), dtype=np.float32)
    store = N.RowStore(rows.shape[1])
    try:
        store.append(rows)
        if cosine_eps is not None:
            store.set_cosine_eps(cosine_eps)
        return store.distances(queries, code)
    finally:
        store.close()


# ---------------------------------------------------------------- query vs collection

def query_distances(query, collection, metric: str = "euclidean") -> np.ndarray:
    """distance/metrics.py:431-493.  ``(d,)`` query against ``(n, d)`` rows ->
    ``(n,)`` distances.  Unknown metric -> ValueError."""
    if metric not in _QUERY_METRICS:
        raise ValueError(f"Unknown metric: {metric}")
    q = N.f32(np.asarray(query)).reshape(1, -1)
    c = _matrix(collection, "collection")
    if c.shape[1] != q.shape[1]:
        raise ValueError(f"query dimension {q.shape[1]} != collection dimension {c.shape[1]}")
    out = _distances(q, c, _QUERY_METRICS[metric])[0]
    if metric == "cosine" and len(out) and float(np.linalg.norm(q[0])) < 1e-8:
        # distance/metrics.py:469-473 returns np.ones(n), i.e. float64, for a zero query; the device wrote
        # the same ones in fp32 (the test on |q| is the reference's own, it only picks the dtype)
        return out.astype(np.float64)
    return out


# ---------------------------------------------------------------- pairwise matrices

def _pairwise(X, Y, code: int, cosine_eps: Optional[float] = None) -> np.ndarray:
    x = _matrix(X, "X")
    y = x if Y is None else _matrix(Y, "Y")
    if x.shape[1] != y.shape[1]:
        raise ValueError(f"dimension mismatch: {x.shape[1]} != {y.shape[1]}")
    return _distances(x, y, code, cosine_eps)


def pairwise_euclidean(X, Y=None) -> np.ndarray:
    """distance/metrics.py:272-306."""
    return _pairwise(X, Y, N.EUCLIDEAN)


def pairwise_euclidean_squared(X, Y=None) -> np.ndarray:
    """distance/metrics.py:309-333."""
    return _pairwise(X, Y, N.EUCLIDEAN_SQ)


def pairwise_cosine(X, Y=None, eps: float = 1e-8) -> np.ndarray:
    """distance/metrics.py:336-373.  Rows of X and Y whose norm is below
    ``eps`` count as unit-norm."""
    return _pairwise(X, Y, N.COSINE, cosine_eps=float(eps))


def pairwise_dot(X, Y=None) -> np.ndarray:
    """distance/metrics.py:376-391 (negative dot products)."""
    return _pairwise(X, Y, N.DOT)


def pairwise_manhattan(X, Y=None) -> np.ndarray:
    """distance/metrics.py:394-424."""
    return _pairwise(X, Y, N.MANHATTAN)


def query_distances_matrix(vectors, metric: str = "euclidean") -> np.ndarray:
    """distance/metrics.py:597-622."""
    table = {"euclidean": pairwise_euclidean, "euclidean_squared": pairwise_euclidean_squared,
             "cosine": pairwise_cosine, "dot": pairwise_dot, "manhattan": pairwise_manhattan}
    if metric not in table:
        raise ValueError(f"Unknown metric: {metric}")
    return table[metric](vectors)


# ---------------------------------------------------------------- per-pair functions
# distance/metrics.py:24-265.  One pair is one query against a one-row store.

_pair_stores: dict = {}                    # dimension -> a one-row store kept between calls
_pair_lock = threading.Lock()


def _pair(a, b, code: int) -> float:
    a = N.f32(np.asarray(a)).reshape(1, -1)
    b = N.f32(np.asarray(b)).reshape(1, -1)
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch: {a.shape} vs {b.shape}")
    # one pair = one query against a one-row store; the store (its stream and buffers) is reused, so a scalar call
    # costs a row upload and a launch, not an allocation
    with _pair_lock:
        store = _pair_stores.get(a.shape[1])
        if store is None:
            if len(_pair_stores) >= 16:
                _pair_stores.popitem()[1].close()
            store = _pair_stores[a.shape[1]] = N.RowStore(a.shape[1])
        store.truncate(0)
        store.append(b)
        return float(store.distances(a, code)[0, 0])


def euclidean(a, b) -> float:
    return _pair(a, b, N.EUCLIDEAN_DIRECT)


def euclidean_squared(a, b) -> float:
    return _pair(a, b, N.EUCLIDEAN_SQ_DIRECT)


def cosine_distance(a, b, eps: float = 1e-8) -> float:
    return _pair(a, b, N.COSINE_PAIR)


def cosine_similarity(a, b, eps: float = 1e-8) -> float:
    return 1.0 - _pair(a, b, N.COSINE_PAIR)


def negative_dot_product(a, b) -> float:
    return _pair(a, b, N.DOT)


def dot_product(a, b) -> float:
    return -_pair(a, b, N.DOT)


def manhattan(a, b) -> float:
    return _pair(a, b, N.MANHATTAN)


def chebyshev(a, b) -> float:
    return _pair(a, b, N.CHEBYSHEV)


def hamming(a, b) -> float:
    """distance/metrics.py:217-237: number of positions where the (fp32) vectors differ."""
    return _pair(a, b, N.HAMMING)


def angular_distance(a, b, eps: float = 1e-8) -> float:
    """distance/metrics.py:521-538: arccos of the clipped cosine similarity, in radians."""
    return _pair(a, b, N.ANGULAR)


def normalize_vectors(vectors, eps: float = 1e-8) -> np.ndarray:
    """distance/metrics.py:545-558.  The row norms come from the device (the
    same pairwise-order sums the searches use)."""
    v = _matrix(vectors, "vectors")
    if v.shape[0] == 0:
        return v.copy()
    store = N.RowStore(v.shape[1])
    try:
        store.append(v)
        _, norms = store.norms()
    finally:
        store.close()
    norms = np.where(norms < eps, np.float32(1.0), norms)[:, None]
    return v / norms
