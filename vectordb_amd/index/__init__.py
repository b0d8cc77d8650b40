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
           "save_index", "load_index"]


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
