"""Index contract shared by the device-backed indices: the result and stats
containers and the abstract interface of the reference's ``index.base``
(index/base.py:30-485), so code written against it runs unchanged."""

from __future__ import annotations

import threading
from contextlib import contextmanager
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


class IndexLock:
    """The lock of an index: writes exclusive and re-entrant (``with lock:``, as the ``threading.RLock`` of
    index/base.py:125), searches shared (``acquire_shared`` / ``release_shared``), so searches of one index do not wait
    for each other.  A writer can turn its hold into a shared one without letting another writer in between
    (``downgrade``): a search that first has to bring the device copy up to date does that and then searches on.
    Waiting writers go before new searches.  A thread that holds the lock for writing may also take it shared."""

    def __init__(self):
        self._cond = threading.Condition(threading.Lock())
        self._readers = 0
        self._writer = None          # thread id of the exclusive holder
        self._depth = 0              # its nesting depth
        self._waiting = 0            # writers waiting
        self._local = threading.local()

    # -- exclusive
    def acquire(self) -> bool:
        me = threading.get_ident()
        with self._cond:
            if self._writer == me:
                self._depth += 1
                return True
            if getattr(self._local, "shared", 0):
                raise RuntimeError("a thread that is searching an index cannot write to it")
            self._waiting += 1
            try:
                while self._writer is not None or self._readers:
                    self._cond.wait()
            finally:
                self._waiting -= 1
            self._writer, self._depth = me, 1
        return True

    def release(self) -> None:
        with self._cond:
            if self._writer != threading.get_ident():
                raise RuntimeError("release of an index lock that this thread does not hold")
            self._depth -= 1
            if self._depth == 0:
                self._writer = None
                self._cond.notify_all()

    __enter__ = acquire

    def __exit__(self, *exc) -> None:
        self.release()

    # -- shared
    def acquire_shared(self) -> str:
        """-> the token ``release_shared`` takes back ("w": this thread holds the lock for writing anyway)."""
        me = threading.get_ident()
        with self._cond:
            if self._writer == me:
                self._depth += 1
                return "w"
            nested = getattr(self._local, "shared", 0)
            while self._writer is not None or (self._waiting and not nested):
                self._cond.wait()
            self._readers += 1
            self._local.shared = nested + 1
        return "r"

    def release_shared(self, token: str) -> None:
        if token == "w":
            self.release()
            return
        with self._cond:
            self._readers -= 1
            self._local.shared -= 1
            if self._readers == 0:
                self._cond.notify_all()

    def downgrade(self) -> str:
        """The exclusive hold of this thread (depth 1) becomes a shared one; -> its token."""
        with self._cond:
            if self._writer != threading.get_ident() or self._depth != 1:
                raise RuntimeError("downgrade needs exactly one exclusive hold by this thread")
            self._writer, self._depth = None, 0
            self._readers += 1
            self._local.shared = getattr(self._local, "shared", 0) + 1
            self._cond.notify_all()
        return "r"


class BaseIndex(ABC):
    """index/base.py:128-485"""

    filter_cache_slots = 4          # evaluated filters kept per index (FilterCache); 0 evaluates on every search

    def __init__(self, dimension: int, metric: str = "euclidean", **kwargs):
        self._dimension = dimension
        self._metric = metric
        self._is_trained = False
        self._build_time = 0.0
        self._lock = IndexLock()
        self._filters = FilterCache(self.filter_cache_slots)

    # ---- searches hold the lock shared; the device copy is brought up to date under an exclusive hold first
    def _mirror_stale(self) -> bool:
        """True while the device copy lacks writes of the host side (subclasses with a device copy override)."""
        return False

    def _sync_mirror(self) -> None:
        """Bring the device copy up to date; runs under the exclusive lock."""

    @contextmanager
    def _searching(self):
        """Shared hold of the index lock around a device search and the mapping of its rows to ids.  The reference's
        searches take no lock at all (index/flat.py:288-343); here rows move on the device when a write swaps them,
        so a search excludes writes, but not other searches."""
        lock = self._lock
        token = lock.acquire_shared()
        try:
            if self._mirror_stale():
                if token == "w":
                    self._sync_mirror()
                else:
                    lock.release_shared(token)
                    token = None
                    lock.acquire()
                    try:
                        if self._mirror_stale():
                            self._sync_mirror()
                    except BaseException:
                        lock.release()
                        raise
                    token = lock.downgrade()
            yield
        finally:
            if token is not None:
                lock.release_shared(token)

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
        self._mu = threading.Lock()      # searches of one index run side by side and share this cache

    def touch(self) -> None:
        """A write to the index (under its exclusive lock: no search is using an entry)."""
        with self._mu:
            self.version += 1
            self._drop()

    def drop(self) -> None:
        with self._mu:
            self._drop()

    def _drop(self) -> None:
        for e in self._entries:
            e[3].close()
        self._entries = []

    def lookup(self, key, build):
        """(host words, resident mask, any) for ``key``; ``build()`` -> boolean row mask runs on a miss.  An entry
        that makes room for a new one is only forgotten, not closed: a search in another thread may still be using
        its mask, which frees its device memory when the last reference goes."""
        def find():
            for i, e in enumerate(self._entries):
                if e[1] == self.version and (e[0] is key or (isinstance(key, str) and isinstance(e[0], str) and e[0] == key)):
                    self._entries.append(self._entries.pop(i))
                    return e
            return None

        with self._mu:
            if self.slots <= 0 and self._entries:
                self._entries = []
            e = find()
            version = self.version
        if e is not None:
            return e[2], e[3], e[4]
        from .. import _native as N
        mask = np.asarray(build(), dtype=bool)          # the caller's callable runs outside the cache's own lock
        words = N.pack_mask(mask)
        with self._mu:
            self.evaluations += 1
            if self.slots <= 0 or version != self.version:
                return words, None, bool(mask.any())
            e = find()                                  # another search evaluated the same filter meanwhile
            if e is not None:
                return e[2], e[3], e[4]
            dev = N.DeviceMask(words, n_bits=len(mask))
            while len(self._entries) >= self.slots:
                self._entries.pop(0)
            self._entries.append((key, self.version, words, dev, bool(mask.any())))
            return words, dev, bool(mask.any())


def filter_mask(filter_fn: Optional[FilterFunction], ids, metadata_of) -> Optional[np.ndarray]:
    """Evaluate ``filter_fn(id, metadata)`` once per row on the host; the
    kernels test the resulting bit inside the scan (index/flat.py:398-403,
    index/ivf.py:622, index/pq.py:292 call it row by row inside the loop)."""
    if filter_fn is None:
        return None
    return np.fromiter((bool(filter_fn(i, metadata_of(i))) for i in ids), dtype=bool, count=len(ids))
