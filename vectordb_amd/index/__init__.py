"""vectordb's ``index`` package on the MI355X search core: FlatIndex, IVFIndex,
PQIndex and HybridIndex (IVF-PQ) with the reference's constructor kwargs and methods.  The HNSW
graph walk is pointer chasing, not a data-parallel scan, and stays on the
reference implementation."""

from .base import BaseIndex, IndexConfig, IndexStats, IndexType, SearchResult, FilterFunction   # noqa: F401
from .flat import FlatIndex, FlatIndexConfig                                                     # noqa: F401
from .ivf import IVFIndex, IVFConfig, KMeans                                                     # noqa: F401
from .pq import PQIndex, PQConfig                                                                # noqa: F401
from .hybrid import HybridIndex, HybridConfig                                                    # noqa: F401
from .io import save_index, load_index                                                           # noqa: F401

__all__ = ["BaseIndex", "IndexConfig", "IndexStats", "SearchResult", "IndexType", "FlatIndex", "FlatIndexConfig",
           "IVFIndex", "IVFConfig", "KMeans", "PQIndex", "PQConfig", "HybridIndex", "HybridConfig", "create_index",
           "recommend_index", "save_index", "load_index"]


def create_index(index_type: str, dimension: int, metric: str = "euclidean", **kwargs) -> BaseIndex:
    """index/__init__.py:56-96 ("pq" and "ivf_pq" are added: the reference has the
    classes but does not reach them from the factory)."""
    index_type = index_type.lower()
    if index_type == "flat":
        return FlatIndex(dimension=dimension, metric=metric, **kwargs)
    if index_type == "ivf":
        return IVFIndex(dimension=dimension, metric=metric, **kwargs)
    if index_type == "pq":
        return PQIndex(dimension=dimension, metric=metric, **kwargs)
    if index_type in ("ivf_pq", "hybrid"):
        return HybridIndex(dimension=dimension, metric=metric, **kwargs)
    if index_type == "hnsw":
        raise ValueError("the HNSW graph index is not part of the device search core; use the reference's HNSWIndex")
    raise ValueError(f"Unknown index type: {index_type}. Available: flat, ivf, pq, ivf_pq")


def recommend_index(n_vectors: int, dimension: int, priority: str = "balanced") -> dict:
    """index/__init__.py:99-170: the reference's size and priority table, value for value (``dimension`` does not
    enter it there either).  Where it names "hnsw", the graph index meant is the reference's own; ``create_index``
    here does not build one."""
    if n_vectors < 10000:
        return {"index_type": "flat", "params": {}, "reason": "Dataset small enough for exact search"}
    hnsw = {"speed": ({"M": 12, "ef_construction": 100, "ef_search": 32}, "HNSW optimized for speed"),
            "recall": ({"M": 32, "ef_construction": 400, "ef_search": 200}, "HNSW optimized for recall")}
    if n_vectors < 100000:
        params, reason = hnsw.get(priority, ({"M": 16, "ef_construction": 200, "ef_search": 64}, "HNSW with balanced parameters"))
        return {"index_type": "hnsw", "params": params, "reason": reason}
    n_clusters = int(4 * (n_vectors ** 0.5))
    if priority == "speed":
        return {"index_type": "ivf", "params": {"n_clusters": n_clusters, "n_probe": max(1, n_clusters // 100)},
                "reason": "IVF optimized for speed on large datasets"}
    if priority == "recall":
        return {"index_type": "hnsw", "params": {"M": 32, "ef_construction": 400, "ef_search": 200},
                "reason": "HNSW provides better recall for large datasets"}
    return {"index_type": "ivf", "params": {"n_clusters": n_clusters, "n_probe": max(1, n_clusters // 20)},
            "reason": "IVF with balanced parameters for large datasets"}
