"""The reference's accelerator operator API (distance/gpu.py) with the HIP core
behind it instead of CuPy: same class, function names, arguments and the same
GPU/CPU selection rule.  ``force_cpu`` and small inputs are still served by the
device (there is no second implementation); they only exist so that callers
written against the reference keep working."""

from __future__ import annotations

import numpy as np

from .. import _native as N

try:
    N.init()                                        # loads libvdbcore.so and selects the device
    HAS_GPU = True
except Exception:                                   # library not built, or no GPU visible
    HAS_GPU = False

_CODES = {
    # distance/gpu.py:118-131: direct-form L2; cosine with +1e-10 in the denominator
    # (served by the per-pair cosine kernel: equal to rounding for non-degenerate rows)
    "euclidean": N.EUCLIDEAN_DIRECT,
    "cosine": N.COSINE_PAIR,
    "dot": N.DOT,
    "manhattan": N.MANHATTAN,
}


class GPUDistanceCalculator:
    """distance/gpu.py:24-151"""

    MIN_VECTORS_FOR_GPU = 10000

    def __init__(self, metric: str = "euclidean", min_vectors_for_gpu: int = 10000, device_id: int = 0):
        self.metric = metric
        self.min_vectors_for_gpu = min_vectors_for_gpu
        self.device_id = device_id
        self._gpu_available = HAS_GPU
        if self._gpu_available:
            # One process drives one GPU (LOCAL_RANK picks it): the calculator works on the device the
            # library already uses.  `device_id` is the reference's argument (gpu.py:63-67 switches the
            # CuPy device); here it is recorded, and a value that names another device than the
            # selected one does not move the library.
            try:
                N.init()
            except Exception:
                self._gpu_available = False

    @property
    def is_gpu_available(self) -> bool:
        return self._gpu_available

    def compute(self, query, vectors, force_gpu: bool = False, force_cpu: bool = False) -> np.ndarray:
        if force_gpu and not self._gpu_available:
            raise RuntimeError("GPU requested but the HIP search core is not available")     # gpu.py:103-104
        if self.metric not in _CODES:
            raise ValueError(f"Unknown metric: {self.metric}")                               # gpu.py:131,151
        if not self._gpu_available:
            raise RuntimeError("the HIP search core is not available (no GPU or libvdbcore.so missing)")
        vectors = N.f32(np.asarray(vectors))
        query = N.f32(np.asarray(query))
        store = N.RowStore(vectors.shape[1])
        try:
            store.append(vectors)
            return store.distances(query, _CODES[self.metric])[0]
        finally:
            store.close()


def gpu_euclidean_batch(query, vectors) -> np.ndarray:
    return GPUDistanceCalculator("euclidean").compute(query, vectors)


def gpu_cosine_batch(query, vectors) -> np.ndarray:
    return GPUDistanceCalculator("cosine").compute(query, vectors)


def gpu_dot_batch(query, vectors) -> np.ndarray:
    """distance/gpu.py:191-202 returns the POSITIVE dot product."""
    return -GPUDistanceCalculator("dot").compute(query, vectors)


def get_gpu_info() -> dict:
    """distance/gpu.py:205-225"""
    if not HAS_GPU:
        return {"available": False}
    try:
        info = N.device_info()
        return {"available": True, "device_id": N.current_device(),
                "name": info["name"], "memory_total": info["memory_total"], "memory_free": info["memory_free"]}
    except Exception as e:
        return {"available": False, "error": str(e)}


__all__ = ["HAS_GPU", "GPUDistanceCalculator", "gpu_euclidean_batch", "gpu_cosine_batch", "gpu_dot_batch", "get_gpu_info"]
