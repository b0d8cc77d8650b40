"""``BatchDistanceCalculator`` and friends (distance/batch.py) on the HIP core:
``top_k`` is one fused distance + selection launch instead of
argpartition/heap loops over chunks."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .. import _native as N
from .metrics import _QUERY_METRICS, normalize_vectors, query_distances
from .registry import get_metric


class BatchDistanceCalculator:
    """distance/batch.py:19-220.  ``chunk_size`` is accepted for signature
    compatibility; the device path needs no host-side chunking."""

    def __init__(self, metric: str = "euclidean", chunk_size: int = 10000, normalize: bool = False):
        self.metric = metric
        self.chunk_size = chunk_size
        self.normalize = normalize
        self.metric_info = get_metric(metric)          # KeyError for unknown names, as the reference

    def _normalize(self, v: np.ndarray) -> np.ndarray:
        v = np.asarray(v, dtype=np.float32)
        if v.ndim == 1:
            return normalize_vectors(v[None, :])[0]
        return normalize_vectors(v)

    def compute(self, query, collection, return_indices: bool = False):
        """batch.py:51-125"""
        query = np.asarray(query)
        collection = np.asarray(collection)
        if self.normalize:
            query, collection = self._normalize(query), self._normalize(collection)
        if query.ndim == 1:
            d = query_distances(query, collection, self.metric)
            return (d, np.argsort(d, kind="stable")) if return_indices else d
        if self.metric not in _QUERY_METRICS:
            raise ValueError(f"Unknown metric: {self.metric}")
        store = N.RowStore(collection.shape[1])
        try:
            store.append(collection)
            d = store.distances(query, _QUERY_METRICS[self.metric])
        finally:
            store.close()
        return (d, np.argsort(d, axis=1, kind="stable")) if return_indices else d

    def top_k(self, query, collection, k: int, return_distances: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """batch.py:136-220 -> (int64 indices, fp32 distances), ascending,
        k capped at the number of rows."""
        if self.metric not in _QUERY_METRICS:
            raise ValueError(f"Unknown metric: {self.metric}")
        query = np.asarray(query)
        collection = np.asarray(collection)
        if self.normalize:
            query, collection = self._normalize(query), self._normalize(collection)
        n = len(collection)
        k = min(int(k), n)
        if k <= 0:
            return np.zeros(0, np.int64), (np.zeros(0, np.float32) if return_distances else None)
        store = N.RowStore(collection.shape[1])
        try:
            store.append(collection)
            dist, rows = store.search(query, k, _QUERY_METRICS[self.metric])
        finally:
            store.close()
        return rows[0], (dist[0] if return_distances else None)


def compute_all_distances(query, collection, metric: str = "euclidean") -> np.ndarray:
    return query_distances(query, collection, metric)


def compute_top_k(query, collection, k: int, metric: str = "euclidean"):
    """batch.py:242-261"""
    return BatchDistanceCalculator(metric=metric).top_k(query, collection, k)


def compute_pairwise_matrix(vectors, metric: str = "euclidean", chunk_size: int = 1000) -> np.ndarray:
    from .metrics import query_distances_matrix
    return query_distances_matrix(vectors, metric)


def find_nearest_neighbors(vectors, k: int, metric: str = "euclidean", include_self: bool = False):
    """batch.py:284-322: all-pairs k-NN as ONE batched search instead of n
    single-query calls."""
    if metric not in _QUERY_METRICS:
        raise ValueError(f"Unknown metric: {metric}")
    v = N.f32(np.asarray(vectors))
    n = len(v)
    kk = min(k + (0 if include_self else 1), n)
    store = N.RowStore(v.shape[1])
    try:
        store.append(v)
        dist, rows = store.search(v, kk, _QUERY_METRICS[metric])
    finally:
        store.close()
    out_i = np.zeros((n, k), dtype=np.int64)
    out_d = np.zeros((n, k), dtype=np.float32)
    for i in range(n):
        r, d = rows[i], dist[i]
        if not include_self:
            keep = r != i
            r, d = r[keep][:k], d[keep][:k]
        out_i[i, :len(r)], out_d[i, :len(d)] = r[:k], d[:k]
    return out_i, out_d
