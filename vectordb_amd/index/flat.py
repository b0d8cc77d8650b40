"""FlatIndex (index/flat.py) on the MI355X search core.

The fp32 matrix lives twice: a growable host matrix (what ``get``,
``iter_vectors`` and ``to_dict`` read, exactly the reference's storage,
index/flat.py:136-141,276-282) and a mirror resident in HBM together with its
row norms.  Rows added one at a time only touch the host matrix; the pending
tail is uploaded at the next search, so ``add`` loops stay cheap.  All
distance arithmetic and the top-k selection run on the device.
"""

from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

from .. import _native as N
from ..distance.metrics import _QUERY_METRICS
from .base import (BaseIndex, FilterFunction, IndexConfig, IndexStats, IndexType, SearchResult, filter_mask,
                   unit_rows, unit_vector)


@dataclass
class FlatIndexConfig(IndexConfig):                       # index/flat.py:49-66
    initial_capacity: int = 1000
    use_matrix: bool = True
    growth_factor: float = 2.0
    normalize: bool = False
    batch_size: int = 10000


class FlatIndex(BaseIndex):
    """Exact k-NN over every stored row (index/flat.py:69-956)."""

    def __init__(self, dimension: int, metric: str = "euclidean", normalize: bool = False, use_matrix: bool = True,
                 initial_capacity: int = 1000, storage: str = "f32", prefilter: Optional[str] = "auto", **kwargs):
        """`storage` is not a reference argument: "bf16" keeps the device copy of the rows in
        bfloat16 (euclidean, euclidean_squared, cosine, dot only; distances are those of inputs
        rounded to bf16).  The host copy returned by get()/include_vectors stays fp32.
        `prefilter` (fp32 storage only, not a reference argument either): "bf16" runs the pass of a batched
        search over all rows on a bf16 copy of them and recomputes the few remaining rows in fp32: results
        are bit for bit those without it, the device holds half as many bytes again.  "auto" (the default)
        does so while the copy leaves a tenth of the device memory free; None never does."""
        super().__init__(dimension, metric, **kwargs)
        if storage not in N.STORAGE:
            raise ValueError(f"storage must be one of {sorted(N.STORAGE)}, got {storage!r}")
        if storage != "f32" and prefilter == "auto":
            prefilter = None
        if prefilter not in N.PREFILTER or (prefilter and storage != "f32"):
            raise ValueError(f"prefilter must be None, 'bf16' or 'auto' (with fp32 storage), got {prefilter!r}")
        self._storage = storage
        self._prefilter = prefilter
        self.config = FlatIndexConfig(dimension=dimension, metric=metric, normalize=normalize, use_matrix=use_matrix,
                                      initial_capacity=initial_capacity)
        self._ids: List[str] = []
        self._id_to_index: Dict[str, int] = {}
        self._metadata: Dict[str, Dict[str, Any]] = {}
        self._capacity = max(1, int(initial_capacity))
        self._vectors = np.zeros((self._capacity, dimension), dtype=np.float32)
        self._size = 0
        self._is_trained = True
        self._build_start = time.time()
        self._store: Optional[N.RowStore] = None       # device mirror, created at the first search
        self._synced = 0                               # rows [0, _synced) are on the device

    # ------------------------------------------------------------ properties
    @property
    def index_type(self) -> IndexType:
        return IndexType.FLAT

    @property
    def size(self) -> int:
        return self._size

    # ------------------------------------------------------------ device mirror
    def _device(self) -> N.RowStore:
        """Device mirror with the pending host tail uploaded.  Call with ``self._lock`` held: two
        threads that both saw the same pending tail would append it twice."""
        if self._store is None:
            self._store = N.RowStore(self._dimension, storage=self._storage)
            if self._prefilter:
                self._store.set_prefilter(self._prefilter)
            self._synced = 0
        if self._synced < self._size:
            self._store.append(self._vectors[self._synced:self._size])
            self._synced = self._size
        return self._store

    def _mirror_stale(self) -> bool:
        return self._store is None or self._synced < self._size

    def _sync_mirror(self) -> None:
        self._device()

    def _metric_code(self) -> int:
        if self._metric not in _QUERY_METRICS:
            raise ValueError(f"Unknown metric: {self._metric}")          # distance/metrics.py:493 via flat.py:331
        return _QUERY_METRICS[self._metric]

    # ------------------------------------------------------------ add
    def _grow(self, required: int) -> None:
        while required > self._capacity:
            self._capacity = max(int(self._capacity * self.config.growth_factor), self._capacity + 1)
        if self._capacity > len(self._vectors):
            grown = np.zeros((self._capacity, self._dimension), dtype=np.float32)
            grown[:self._size] = self._vectors[:self._size]
            self._vectors = grown

    def add(self, id: str, vector, metadata: Optional[Dict[str, Any]] = None) -> None:
        vector = self._validate_vector(vector)
        if self.config.normalize:
            vector = unit_vector(vector)
        with self._lock:
            if id in self._id_to_index:
                raise ValueError(f"ID '{id}' already exists in index")
            self._grow(self._size + 1)
            self._vectors[self._size] = vector
            self._ids.append(id)
            self._id_to_index[id] = self._size
            self._metadata[id] = metadata or {}
            self._size += 1
            self._filters.touch()

    def add_batch(self, ids: List[str], vectors, metadata: Optional[List[Dict[str, Any]]] = None) -> int:
        vectors = self._validate_vectors(vectors)
        n = len(ids)
        if len(vectors) != n:
            raise ValueError(f"Number of ids ({n}) != number of vectors ({len(vectors)})")
        if metadata is not None and len(metadata) != n:
            raise ValueError(f"Number of ids ({n}) != number of metadata ({len(metadata)})")
        if self.config.normalize:
            vectors = unit_rows(vectors)
        with self._lock:
            seen = set()
            for id in ids:
                if id in self._id_to_index or id in seen:
                    raise ValueError(f"ID '{id}' already exists in index")
                seen.add(id)
            self._grow(self._size + n)
            self._vectors[self._size:self._size + n] = vectors
            for i, id in enumerate(ids):
                self._ids.append(id)
                self._id_to_index[id] = self._size + i
                self._metadata[id] = metadata[i] if metadata else {}
            self._size += n
            self._filters.touch()
        return n

    # ------------------------------------------------------------ search
    def _results(self, dist_row, rows_row, include_vectors: bool, include_metadata: bool) -> List[SearchResult]:
        out = []
        for d, r in zip(dist_row, rows_row):
            if r < 0:
                break
            id = self._ids[r]
            d = float(d)
            out.append(SearchResult(
                id=id, distance=d, score=1.0 / (1.0 + d),                                  # index/flat.py:370-373
                vector=self._vectors[r].copy() if include_vectors else None,
                metadata=self._metadata.get(id, {}).copy() if include_metadata else None))
        return out

    def _mask_words(self, filter_fn: Optional[FilterFunction], host_words: bool = False):
        return self._filter_for_device(filter_fn, self._ids, lambda i: self._metadata.get(i, {}), host_words)

    def search(self, query, k: int = 10, filter_fn: Optional[FilterFunction] = None, include_vectors: bool = False,
               include_metadata: bool = True) -> List[SearchResult]:
        if self._size == 0:
            return []
        query = self._validate_vector(query)
        if self.config.normalize:
            query = unit_vector(query)
        k = min(k, self._size)
        if k <= 0:
            return []
        code = self._metric_code()
        # device call and row -> id mapping under a shared hold of the index lock: a concurrent add or remove must
        # not move rows between the search and the mapping; other searches may run
        with self._searching():
            k = min(k, self._size)
            if k <= 0:
                return []
            dist, rows = self._store.search(query, k, code, self._mask_words(filter_fn))
            return self._results(dist[0], rows[0], include_vectors, include_metadata)

    def search_batch(self, queries, k: int = 10, filter_fn: Optional[FilterFunction] = None) -> List[List[SearchResult]]:
        """One device call for the whole batch (the reference loops over
        ``search``, index/flat.py:457-460)."""
        queries = self._validate_vectors(queries)
        if self._size == 0:
            return [[] for _ in range(len(queries))]
        if self.config.normalize:
            queries = unit_rows(queries)
        with self._searching():
            k = min(k, self._size)
            if k <= 0 or len(queries) == 0:
                return [[] for _ in range(len(queries))]
            dist, rows = self._store.search(queries, k, self._metric_code(), self._mask_words(filter_fn))
            return [self._results(dist[i], rows[i], False, True) for i in range(len(queries))]

    def search_by_id(self, id: str, k: int = 10, exclude_self: bool = True,
                     filter_fn: Optional[FilterFunction] = None) -> List[SearchResult]:
        if id not in self._id_to_index:
            raise KeyError(f"ID '{id}' not found in index")
        query = self._vectors[self._id_to_index[id]].copy()
        if exclude_self:
            inner = filter_fn
            filter_fn = lambda vid, meta: vid != id and (inner is None or inner(vid, meta))   # noqa: E731
        return self.search(query, k=k, filter_fn=filter_fn)

    def range_search(self, query, radius: float, filter_fn: Optional[FilterFunction] = None,
                     max_results: Optional[int] = None) -> List[SearchResult]:
        """index/flat.py:501-572: rows inside the radius that pass the filter;
        with ``max_results`` the FIRST that many in insertion order (:545-567),
        sorted by distance with equal distances in insertion order (:570)."""
        if self._size == 0:
            return []
        query = self._validate_vector(query)
        if self.config.normalize:
            query = unit_vector(query)
        with self._searching():
            if self._size == 0:
                return []
            dist, rows = self._store.range_search(query, radius, self._metric_code(), self._mask_words(filter_fn, host_words=True),
                                                     cap=max_results or self._size)
            return self._results(dist, rows, False, True)

    # ------------------------------------------------------------ remove / update
    def remove(self, id: str) -> bool:
        """Swap-with-last, as index/flat.py:578-611, mirrored on the device."""
        with self._lock:
            if id not in self._id_to_index:
                return False
            idx = self._id_to_index[id]
            last = self._size - 1
            synced = self._store is not None and self._synced == self._size
            if self._store is not None and not synced:
                self._device()
                synced = True
            if idx < last:
                last_id = self._ids[-1]
                self._vectors[idx] = self._vectors[last]
                self._id_to_index[last_id] = idx
                self._ids[idx] = last_id
                if synced:
                    self._store.set_row(idx, self._vectors[idx])
            if synced:
                self._store.truncate(last)
                self._synced = last
            del self._id_to_index[id]
            del self._metadata[id]
            self._ids.pop()
            self._size -= 1
            self._filters.touch()
            return True

    def remove_batch(self, ids: List[str]) -> int:
        return sum(1 for id in ids if self.remove(id))

    def update(self, id: str, vector=None, metadata: Optional[Dict[str, Any]] = None) -> bool:
        with self._lock:
            if id not in self._id_to_index:
                return False
            idx = self._id_to_index[id]
            if vector is not None:
                vector = self._validate_vector(vector)
                if self.config.normalize:
                    vector = unit_vector(vector)
                self._vectors[idx] = vector
                if self._store is not None and idx < self._synced:
                    self._store.set_row(idx, self._vectors[idx])
            if metadata is not None:
                self._metadata[id] = metadata
                self._filters.touch()
            return True

    # ------------------------------------------------------------ get / misc
    def get(self, id: str) -> Optional[Tuple[np.ndarray, Dict[str, Any]]]:
        if id not in self._id_to_index:
            return None
        return self._vectors[self._id_to_index[id]].copy(), self._metadata.get(id, {}).copy()

    def get_vectors(self, ids: List[str]) -> Dict[str, Optional[np.ndarray]]:
        return {id: (self._vectors[self._id_to_index[id]].copy() if id in self._id_to_index else None) for id in ids}

    def contains(self, id: str) -> bool:
        return id in self._id_to_index

    def clear(self) -> int:
        with self._lock:
            count = self._size
            self._ids.clear()
            self._id_to_index.clear()
            self._metadata.clear()
            self._capacity = max(1, int(self.config.initial_capacity))
            self._vectors = np.zeros((self._capacity, self._dimension), dtype=np.float32)
            self._size = 0
            if self._store is not None:
                self._store.truncate(0)
            self._synced = 0
            self._filters.touch()
            return count

    def stats(self) -> IndexStats:
        meta_mem = sum(len(str(m)) for m in self._metadata.values())
        id_mem = sum(len(i) for i in self._ids) * 2
        return IndexStats(index_type=self.index_type.value, dimension=self._dimension, metric=self._metric,
                          vector_count=self._size, memory_bytes=self._vectors.nbytes + meta_mem + id_mem,
                          is_trained=True, build_time_seconds=time.time() - self._build_start,
                          extra={"use_matrix": self.config.use_matrix, "normalize": self.config.normalize,
                                 "capacity": self._capacity,
                                 "device_bytes": self._synced * (self._dimension + 2) * 4})

    def to_dict(self) -> Dict[str, Any]:
        return {"index_type": self.index_type.value, "dimension": self._dimension, "metric": self._metric,
                "config": {"normalize": self.config.normalize, "use_matrix": self.config.use_matrix,
                           "initial_capacity": self.config.initial_capacity, "storage": self._storage,
                           "prefilter": self._prefilter},
                "ids": self._ids.copy(), "vectors": self._vectors[:self._size].tolist(),
                "metadata": {id: meta.copy() for id, meta in self._metadata.items()}}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "FlatIndex":
        cfg = data.get("config", {})
        ids = data.get("ids", [])
        index = cls(dimension=data["dimension"], metric=data["metric"], normalize=cfg.get("normalize", False),
                    use_matrix=cfg.get("use_matrix", True),
                    initial_capacity=max(cfg.get("initial_capacity", 1000), len(ids)),
                    storage=cfg.get("storage", "f32"), prefilter=cfg.get("prefilter", "auto"))
        if len(ids) > 0:
            meta = data.get("metadata", {})
            index.add_batch(ids, np.array(data.get("vectors", []), dtype=np.float32), [meta.get(i, {}) for i in ids])
        return index

    def iter_ids(self) -> Iterator[str]:
        return iter(self._ids)

    def iter_vectors(self) -> Iterator[Tuple[str, np.ndarray, Dict[str, Any]]]:
        for id in self._ids:
            yield id, self._vectors[self._id_to_index[id]].copy(), self._metadata.get(id, {})

    def get_all_vectors(self) -> np.ndarray:
        return self._vectors[:self._size].copy()

    def get_all_ids(self) -> List[str]:
        return self._ids.copy()

    def compute_distance_matrix(self) -> np.ndarray:
        """index/flat.py:880-901"""
        with self._lock:
            if self._size == 0:
                return np.zeros((0, 0), dtype=np.float32)
            return self._device().distances(self._vectors[:self._size], self._metric_code())

    def find_duplicates(self, threshold: float = 1e-6) -> List[Tuple[str, str, float]]:
        """index/flat.py:903-930: pairs (id_i, id_j, distance), i < j in insertion order, whose distance is
        at most `threshold`.  The reference asks query_distances for every pair; here blocks of rows go
        through the device's distance kernels against all rows.  Euclidean pairs are taken in the direct
        form sum((a-b)^2), which is exactly 0 for identical rows (the identity form the reference uses
        there cancels to rounding noise around sqrt(eps)*|a|; away from the threshold both agree to 1e-4)."""
        out: List[Tuple[str, str, float]] = []
        n = self._size
        if n < 2:
            return out
        code = self._metric_code()
        if code == N.EUCLIDEAN:
            code = N.EUCLIDEAN_DIRECT
        elif code == N.EUCLIDEAN_SQ:
            code = N.EUCLIDEAN_SQ_DIRECT
        with self._lock:
            n = self._size
            store = self._device()
            block = max(1, min(n, (64 << 20) // (4 * max(n, 1))))     # at most 64 MB of distances at a time
            for i0 in range(0, n, block):
                d = store.distances(self._vectors[i0:min(i0 + block, n)], code)
                ii, jj = np.nonzero(d <= threshold)
                keep = jj > ii + i0
                for a, b in zip(ii[keep], jj[keep]):                   # row-major: i ascending, then j ascending
                    out.append((self._ids[i0 + int(a)], self._ids[int(b)], float(d[a, b])))
        return out

    def sample(self, n: int, seed: Optional[int] = None) -> List[Tuple[str, np.ndarray, Dict[str, Any]]]:
        """index/flat.py:932-956: n rows drawn without replacement through NumPy's global generator
        (seeded when `seed` is given), as (id, vector, metadata)."""
        if seed is not None:
            np.random.seed(seed)
        n = min(n, self._size)
        indices = np.random.choice(self._size, n, replace=False)
        return [(self._ids[i], self._vectors[i].copy(), self._metadata.get(self._ids[i], {})) for i in indices]

    def __del__(self):
        store = getattr(self, "_store", None)
        if store is not None:
            store.close()
        filters = getattr(self, "_filters", None)
        if filters is not None:
            filters.drop()
