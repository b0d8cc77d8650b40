"""Index contract shared by the device-backed indices: the result and stats
containers and the abstract interface of the reference's ``index.base``
(index/base.py:30-485), so code written against it runs unchanged."""

from __future__ import annotations

import threading
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple

import numpy as np

FilterFunction = Callable[[str, Dict[str, Any]], bool]          # index/base.py:125


class IndexType(str, Enum):                                      # index/base.py:30-36
    FLAT = "flat"
    IVF = "ivf"
    HNSW = "hnsw"
    PQ = "pq"
    IVF_PQ = "ivf_pq"


@dataclass
class IndexConfig:                                               # index/base.py:39-49
    dimension: int
    metric: str = "euclidean"

    def validate(self) -> None:
        if self.dimension < 1:
            raise ValueError(f"Dimension must be >= 1, got {self.dimension}")


@dataclass
class IndexStats:                                                # index/base.py:52-79
    index_type: str
    dimension: int
    metric: str
    vector_count: int
    memory_bytes: int
    is_trained: bool
    build_time_seconds: float = 0.0
    extra: Dict[str, Any] = field(default_factory=dict)

    def to_dict(self) -> Dict[str, Any]:
        out = {"index_type": self.index_type, "dimension": self.dimension, "metric": self.metric,
               "vector_count": self.vector_count, "memory_bytes": self.memory_bytes,
               "memory_mb": round(self.memory_bytes / (1024 * 1024), 2), "is_trained": self.is_trained,
               "build_time_seconds": self.build_time_seconds}
        out.update(self.extra)
        return out


@dataclass
class SearchResult:                                              # index/base.py:82-121
    id: str
    distance: float
    score: float = 0.0
    vector: Optional[np.ndarray] = None
    metadata: Optional[Dict[str, Any]] = None

    def __post_init__(self):
        if self.score == 0.0 and self.distance >= 0:
            self.score = 1.0 / (1.0 + self.distance)

    def to_dict(self, include_vector: bool = False) -> Dict[str, Any]:
        out = {"id": self.id, "distance": self.distance, "score": self.score}
        if self.metadata is not None:
            out["metadata"] = self.metadata
        if include_vector and self.vector is not None:
            out["vector"] = self.vector.tolist()
        return out

    def __repr__(self) -> str:
        return f"SearchResult(id='{self.id}', distance={self.distance:.4f})"


class BaseIndex(ABC):
    """index/base.py:128-485"""

    filter_cache_slots = 4          # evaluated filters kept per index (FilterCache); 0 evaluates on every search

    def __init__(self, dimension: int, metric: str = "euclidean", **kwargs):
        self._dimension = dimension
        self._metric = metric
        self._is_trained = False
        self._build_time = 0.0
        self._lock = threading.RLock()
        self._filters = FilterCache(self.filter_cache_slots)

    @property
    def dimension(self) -> int:
        return self._dimension

    @property
    def metric(self) -> str:
        return self._metric

    @property
    def is_trained(self) -> bool:
        return self._is_trained

    @property
    @abstractmethod
    def index_type(self) -> IndexType: ...

    @property
    @abstractmethod
    def size(self) -> int: ...

    @abstractmethod
    def add(self, id: str, vector, metadata: Optional[Dict[str, Any]] = None) -> None: ...

    @abstractmethod
    def add_batch(self, ids: List[str], vectors, metadata: Optional[List[Dict[str, Any]]] = None) -> int: ...

    @abstractmethod
    def search(self, query, k: int = 10, filter_fn: Optional[FilterFunction] = None,
               include_vectors: bool = False, include_metadata: bool = True) -> List[SearchResult]: ...

    @abstractmethod
    def search_batch(self, queries, k: int = 10, filter_fn: Optional[FilterFunction] = None) -> List[List[SearchResult]]: ...

    @abstractmethod
    def remove(self, id: str) -> bool: ...

    @abstractmethod
    def remove_batch(self, ids: List[str]) -> int: ...

    @abstractmethod
    def get(self, id: str) -> Optional[Tuple[np.ndarray, Dict[str, Any]]]: ...

    @abstractmethod
    def contains(self, id: str) -> bool: ...

    @abstractmethod
    def clear(self) -> int: ...

    @abstractmethod
    def stats(self) -> IndexStats: ...

    def train(self, vectors) -> None:
        self._is_trained = True

    def rebuild(self) -> None:
        pass

    def optimize(self) -> None:
        pass

    @abstractmethod
    def to_dict(self) -> Dict[str, Any]: ...

    @classmethod
    @abstractmethod
    def from_dict(cls, data: Dict[str, Any]) -> "BaseIndex": ...

    @abstractmethod
    def iter_ids(self) -> Iterator[str]: ...

    @abstractmethod
    def iter_vectors(self) -> Iterator[Tuple[str, np.ndarray, Dict[str, Any]]]: ...

    # input coercion: contiguous fp32, dimension checked with ValueError (index/base.py:435-469)
    def _validate_vector(self, vector) -> np.ndarray:
        if not isinstance(vector, np.ndarray):
            vector = np.array(vector, dtype=np.float32)
        if vector.dtype != np.float32:
            vector = vector.astype(np.float32)
        if vector.ndim != 1:
            raise ValueError(f"Vector must be 1D, got {vector.ndim}D")
        if len(vector) != self._dimension:
            raise ValueError(f"Vector dimension {len(vector)} != index dimension {self._dimension}")
        return vector

    def _validate_vectors(self, vectors) -> np.ndarray:
        if not isinstance(vectors, np.ndarray):
            vectors = np.array(vectors, dtype=np.float32)
        if vectors.dtype != np.float32:
            vectors = vectors.astype(np.float32)
        if vectors.ndim != 2:
            raise ValueError(f"Vectors must be 2D, got {vectors.ndim}D")
        if vectors.shape[1] != self._dimension:
            raise ValueError(f"Vector dimension {vectors.shape[1]} != index dimension {self._dimension}")
        return vectors

    def _filter_for_device(self, filter_fn: Optional[FilterFunction], ids, metadata_of, host_words: bool = False):
        """What a device search takes as its filter: None, or the resident mask of ``filter_fn`` over ``ids`` (a row
        whose id is None never passes).  ``host_words=True`` returns the packed words instead."""
        if filter_fn is None:
            return None
        if self._filters.slots != self.filter_cache_slots:
            self._filters.slots = self.filter_cache_slots
        words, dev, _ = self._filters.lookup(
            filter_fn, lambda: np.fromiter((i is not None and bool(filter_fn(i, metadata_of(i))) for i in ids),
                                           dtype=bool, count=len(ids)))
        return words if (host_words or dev is None) else dev

    def __len__(self) -> int:
        return self.size

    def __contains__(self, id: str) -> bool:
        return self.contains(id)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(dimension={self._dimension}, metric='{self._metric}', size={self.size})"


# ---------------------------------------------------------------- shared helpers

def unit_rows(vectors: np.ndarray) -> np.ndarray:
    """Row normalisation used by ``normalize=True`` (index/flat.py:243-246)."""
    norms = np.linalg.norm(vectors, axis=1, keepdims=True)
    norms = np.where(norms < 1e-8, 1.0, norms)
    return (vectors / norms).astype(np.float32)


def unit_vector(vector: np.ndarray) -> np.ndarray:
    """index/flat.py:186-189,314-317"""
    norm = np.linalg.norm(vector)
    return vector / norm if norm > 1e-8 else vector


class FilterCache:
    """Evaluated filters of one index, kept on the device.

    The reference calls ``filter_fn(id, metadata)`` for every row inside every filtered search
    (index/flat.py:396-413, index/ivf.py:622, index/pq.py:287-297).  Here a filter is evaluated once per write
    version of the index: the packed row mask stays in HBM (``_native.DeviceMask``) and later searches with the
    same filter name it, so they neither call ``filter_fn`` nor upload anything.  "The same filter" is the same
    callable object (or an equal string key, which is what ``Collection`` derives from a filter dict); a callable
    whose answers change without a write to the index (it reads outside state) must not be cached: set
    ``filter_cache_slots = 0`` on the index.  Any add, remove, update or clear drops every entry."""

    def __init__(self, slots: int = 4):
        self.slots = slots
        self.version = 0                 # bumped by every write of the owning index
        self._entries: list = []         # (key, version, packed host words, DeviceMask, any row passes); newest last
        self.evaluations = 0             # filters evaluated over all rows (a cache hit does not count)

    def touch(self) -> None:
        self.version += 1
        if self._entries:
            self.drop()

    def drop(self) -> None:
        for e in self._entries:
            e[3].close()
        self._entries = []

    def lookup(self, key, build):
        """(host words, resident mask, any) for ``key``; ``build()`` -> boolean row mask runs on a miss."""
        if self.slots <= 0 and self._entries:
            self.drop()
        for i, e in enumerate(self._entries):
            if e[1] == self.version and (e[0] is key or (isinstance(key, str) and isinstance(e[0], str) and e[0] == key)):
                self._entries.append(self._entries.pop(i))
                return e[2], e[3], e[4]
        from .. import _native as N
        mask = np.asarray(build(), dtype=bool)
        self.evaluations += 1
        words = N.pack_mask(mask)
        if self.slots <= 0:
            return words, None, bool(mask.any())
        dev = N.DeviceMask(words, n_bits=len(mask))
        while len(self._entries) >= self.slots:
            self._entries.pop(0)[3].close()
        self._entries.append((key, self.version, words, dev, bool(mask.any())))
        return words, dev, bool(mask.any())


def filter_mask(filter_fn: Optional[FilterFunction], ids, metadata_of) -> Optional[np.ndarray]:
    """Evaluate ``filter_fn(id, metadata)`` once per row on the host; the
    kernels test the resulting bit inside the scan (index/flat.py:398-403,
    index/ivf.py:622, index/pq.py:292 call it row by row inside the loop)."""
    if filter_fn is None:
        return None
    return np.fromiter((bool(filter_fn(i, metadata_of(i))) for i in ids), dtype=bool, count=len(ids))
