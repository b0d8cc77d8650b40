"""The reference's second accelerator seam (distance/simd.py) on the HIP core: the same names,
``fn(query (d,), vectors (n, d)) -> (n,)`` fp32 arrays, the same unknown-metric error.

The reference fills these names with Numba ``prange`` loops, or NumPy expressions when Numba is
absent; both compute the DIRECT forms (sqrt(sum((q - v)^2)), sum|q - v|, the +1e-10 cosine), and the
dot function returns the POSITIVE dot product.  Here the same forms run in the streaming kernels.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _native as N

HAS_NUMBA = False          # nothing here is compiled by Numba; the flag keeps importers of the name working


def _scan(query, vectors, code: int) -> np.ndarray:
    v = np.asarray(vectors)
    if v.ndim != 2:
        raise ValueError(f"vectors must be 2D, got {v.ndim}D")
    v = N.f32(v)
    q = N.f32(np.asarray(query)).reshape(1, -1)
    if q.shape[1] != v.shape[1]:
        raise ValueError(f"query dimension {q.shape[1]} != vectors dimension {v.shape[1]}")
    if len(v) == 0:
        return np.zeros(0, dtype=np.float32)
    store = N.RowStore(v.shape[1])
    try:
        store.append(v)
        return store.distances(q, code)[0]
    finally:
        store.close()


def simd_euclidean_batch(query, vectors) -> np.ndarray:
    """distance/simd.py:29-50: sqrt(sum((query - v)^2)) per row."""
    return _scan(query, vectors, N.EUCLIDEAN_DIRECT)


def simd_dot_product_batch(query, vectors) -> np.ndarray:
    """distance/simd.py:54-71: the dot products themselves (not negated)."""
    return -_scan(query, vectors, N.DOT)


def simd_cosine_batch(query, vectors) -> np.ndarray:
    """distance/simd.py:75-98: 1 - q.v / (|q||v| + 1e-10); served by the per-pair cosine kernel, equal to
    rounding for rows whose norm product is not within a few 1e-10 of zero."""
    return _scan(query, vectors, N.COSINE_PAIR)


def simd_manhattan_batch(query, vectors) -> np.ndarray:
    """distance/simd.py:102-119."""
    return _scan(query, vectors, N.MANHATTAN)


# the NumPy fallbacks of the reference (distance/simd.py:122-146) compute the same three forms
numpy_euclidean_batch = simd_euclidean_batch
numpy_cosine_batch = simd_cosine_batch
numpy_dot_batch = simd_dot_product_batch


def get_optimized_distance_fn(metric: str, use_numba: Optional[bool] = None):
    """distance/simd.py:149-181: the batch function of a metric; ``use_numba`` is accepted and has
    nothing to choose between here.  Unknown metric -> ValueError."""
    functions = {"euclidean": simd_euclidean_batch, "cosine": simd_cosine_batch, "dot": simd_dot_product_batch,
                 "manhattan": simd_manhattan_batch}
    if metric not in functions:
        raise ValueError(f"Unknown metric: {metric}")
    return functions[metric]


__all__ = ["HAS_NUMBA", "simd_euclidean_batch", "simd_dot_product_batch", "simd_cosine_batch", "simd_manhattan_batch",
           "numpy_euclidean_batch", "numpy_cosine_batch", "numpy_dot_batch", "get_optimized_distance_fn"]
