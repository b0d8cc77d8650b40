"""PQIndex (index/pq.py) on the MI355X search core.

Device work: sub-space k-means assignment and sums during training, encoding
(bit-identical codes), the per-query distance tables and the ADC scan over the
codes resident in HBM with the filter bit tested in the scan.  As in the
reference the ``metric`` argument is accepted and ignored: PQ distances are
always Euclidean (index/pq.py:268-297).
"""

from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

from .. import _native as N
from .base import BaseIndex, FilterFunction, IndexConfig, IndexStats, IndexType, SearchResult


@dataclass
class PQConfig(IndexConfig):                                # index/pq.py:26-48
    n_subvectors: int = 8
    n_bits: int = 8
    n_iter: int = 20
    n_training_samples: int = 10000
    seed: Optional[int] = None

    @property
    def n_centroids(self) -> int:
        return 2 ** self.n_bits


def train_subquantizers(vectors: np.ndarray, m: int, ksub: int, n_iter: int, rng: np.random.RandomState) -> np.ndarray:
    """The sub-quantizer training shared by PQIndex (pq.py:132-180) and HybridIndex (hybrid.py:167-203):
    per sub-space, start from ``rng.choice(n, k, replace=False)`` rows and run n_iter Lloyd steps.  The rows
    are uploaded once; a trainer per sub-space works on its columns on the device in the reference's
    arithmetic (fp32 in the first step, float64 afterwards, sums in sample order), and the float64
    centroids are rounded to fp32 when they enter the codebook, as ``codebook[m] = centroids`` does."""
    n, d = vectors.shape
    ds = d // m
    k = min(ksub, n)
    book = np.zeros((m, ksub, ds), dtype=np.float32)
    if n_iter <= 0:
        for j in range(m):
            book[j, :k] = vectors[rng.choice(n, k, replace=False), j * ds:(j + 1) * ds]
        return book
    store = N.RowStore(d)
    try:
        store.append(vectors)
        for j in range(m):
            trainer = N.KMeansTrainer(store, j * ds, ds, k)
            try:
                trainer.init_rows(rng.choice(n, k, replace=False))           # pq.py:150-151
                for _ in range(n_iter):
                    trainer.step(N.KMEANS_PQ)
                    trainer.commit()
                book[j, :k] = trainer.centroids()
            finally:
                trainer.close()
    finally:
        store.close()
    return book


class PQIndex(BaseIndex):
    """index/pq.py:51-436"""

    def __init__(self, dimension: int, metric: str = "euclidean", n_subvectors: int = 8, n_bits: int = 8,
                 seed: Optional[int] = None, **kwargs):
        super().__init__(dimension, metric, **kwargs)
        if dimension % n_subvectors != 0:
            raise ValueError(f"Dimension {dimension} must be divisible by n_subvectors {n_subvectors}")
        if not 1 <= n_bits <= 8:
            raise ValueError(f"n_bits must be in 1..8 (codes are stored as uint8), got {n_bits}")
        self.config = PQConfig(dimension=dimension, metric=metric, n_subvectors=n_subvectors, n_bits=n_bits, seed=seed)
        self._subvector_dim = dimension // n_subvectors
        self._n_centroids = 2 ** n_bits
        self._codebook: Optional[np.ndarray] = None
        self._ids: List[str] = []
        self._row_of: Dict[str, int] = {}
        self._codes_arr = np.zeros((1024, n_subvectors), dtype=np.uint8)
        self._metadata: Dict[str, Dict[str, Any]] = {}
        self._rng = np.random.RandomState(seed)
        self._build_start = time.time()
        self._dev: Optional[N.PqCodes] = None
        self._synced = 0          # rows [0, _synced) of _codes_arr are on the device

    @property
    def index_type(self) -> IndexType:
        return IndexType.PQ

    @property
    def size(self) -> int:
        return len(self._ids)

    @property
    def _codes(self) -> Dict[str, np.ndarray]:
        """id -> codes view, the shape of the reference's storage (pq.py:100)."""
        return {id: self._codes_arr[r] for id, r in self._row_of.items()}

    # ------------------------------------------------------------ training
    def train(self, vectors) -> None:
        """pq.py:115-180: per sub-space, k-means from ``rng.choice(n, k,
        replace=False)`` rows, n_iter Lloyd steps, empty clusters collapse to
        the zero vector, zero padding when n < 2**n_bits."""
        vectors = self._validate_vectors(vectors)
        self._set_codebook(train_subquantizers(vectors, self.config.n_subvectors, self._n_centroids, self.config.n_iter,
                                               self._rng))

    def _set_codebook(self, book: np.ndarray) -> None:
        self._codebook = N.f32(book)
        self._is_trained = True
        if self._dev is not None:
            self._dev.close()
        self._dev = N.PqCodes(self._dimension, self.config.n_subvectors, self._n_centroids)
        self._dev.set_codebook(self._codebook)
        self._synced = 0

    # ------------------------------------------------------------ add
    def _encode(self, vector) -> np.ndarray:
        return N.pq_encode(N.f32(vector)[None, :], self._codebook)[0]

    def _decode(self, codes) -> np.ndarray:
        return np.concatenate([self._codebook[j, codes[j]] for j in range(self.config.n_subvectors)]).astype(np.float32)

    def _reserve(self, n: int) -> None:
        if n > len(self._codes_arr):
            grown = np.zeros((max(n, 2 * len(self._codes_arr)), self.config.n_subvectors), dtype=np.uint8)
            grown[:self.size] = self._codes_arr[:self.size]
            self._codes_arr = grown

    def add(self, id: str, vector, metadata: Optional[Dict[str, Any]] = None) -> None:
        if not self._is_trained:
            raise RuntimeError("Index must be trained before adding vectors")
        vector = self._validate_vector(vector)
        if id in self._row_of:
            raise ValueError(f"ID '{id}' already exists")
        self.add_batch([id], vector[None, :], [metadata or {}])

    def add_batch(self, ids: List[str], vectors, metadata: Optional[List[Dict[str, Any]]] = None) -> int:
        """Encodes the whole batch in one kernel launch (the reference loops
        over ``add``, pq.py:242-243) and appends the codes on the device."""
        vectors = self._validate_vectors(vectors)
        if not self._is_trained:
            raise RuntimeError("Index must be trained before adding vectors")
        n = len(ids)
        if len(vectors) != n:
            raise ValueError(f"Number of ids ({n}) != vectors ({len(vectors)})")
        seen = set()
        for id in ids:
            if id in self._row_of or id in seen:
                raise ValueError(f"ID '{id}' already exists")
            seen.add(id)
        if n == 0:
            return 0
        with self._lock:
            self._sync()
            codes = self._dev.append_rows(vectors)
            base = self.size
            self._reserve(base + n)
            self._codes_arr[base:base + n] = codes
            for i, id in enumerate(ids):
                self._row_of[id] = base + i
                self._ids.append(id)
                self._metadata[id] = (metadata[i] if metadata else None) or {}
            self._synced = base + n
            self._filters.touch()
        return n

    def _sync(self) -> None:
        """Bring the device codes in line with the host array (after removals
        or ``from_dict``)."""
        if self._synced != self.size:
            self._dev.truncate(0)
            if self.size:
                self._dev.append_codes(self._codes_arr[:self.size])
            self._synced = self.size

    def _mirror_stale(self) -> bool:
        return self._synced != self.size

    def _sync_mirror(self) -> None:
        self._sync()

    # ------------------------------------------------------------ search
    def _mask_words(self, filter_fn: Optional[FilterFunction]):
        return self._filter_for_device(filter_fn, self._ids, lambda i: self._metadata.get(i, {}))

    def _results(self, dist_row, rows_row, include_vectors: bool, include_metadata: bool) -> List[SearchResult]:
        out = []
        for d, r in zip(dist_row, rows_row):
            if r < 0:
                break
            id = self._ids[r]
            out.append(SearchResult(id=id, distance=d, score=1.0 / (1.0 + d),          # np.float32, as pq.py:297,310-316
                                    vector=self._decode(self._codes_arr[r]) if include_vectors else None,
                                    metadata=self._metadata.get(id, {}) if include_metadata else None))
        return out

    def search(self, query, k: int = 10, filter_fn: Optional[FilterFunction] = None, include_vectors: bool = False,
               include_metadata: bool = True) -> List[SearchResult]:
        if not self._is_trained:
            raise RuntimeError("Index must be trained")
        if self.size == 0:
            return []
        query = self._validate_vector(query)
        # device call and row -> id mapping under a shared hold of the index lock (a concurrent add or remove
        # renumbers rows; other searches may run)
        with self._searching():
            k = min(k, self.size)
            if k <= 0:
                return []
            dist, rows = self._dev.search(query, k, self._mask_words(filter_fn))
            return self._results(dist[0], rows[0], include_vectors, include_metadata)

    def search_batch(self, queries, k: int = 10, filter_fn: Optional[FilterFunction] = None) -> List[List[SearchResult]]:
        if not self._is_trained:
            raise RuntimeError("Index must be trained")
        queries = self._validate_vectors(queries)
        if self.size == 0 or len(queries) == 0:
            return [[] for _ in range(len(queries))]
        with self._searching():
            k = min(k, self.size)
            if k <= 0:
                return [[] for _ in range(len(queries))]
            dist, rows = self._dev.search(queries, k, self._mask_words(filter_fn))
            return [self._results(dist[i], rows[i], False, True) for i in range(len(queries))]

    # ------------------------------------------------------------ remove / get
    def remove(self, id: str) -> bool:
        """Order-preserving removal (``list.remove``, pq.py:329-337): later
        rows move up by one, so the device codes are re-uploaded lazily."""
        with self._lock:
            if id not in self._row_of:
                return False
            r = self._row_of.pop(id)
            n = self.size
            self._codes_arr[r:n - 1] = self._codes_arr[r + 1:n]
            del self._ids[r]
            del self._metadata[id]
            for j in range(r, n - 1):
                self._row_of[self._ids[j]] = j
            self._synced = -1
            self._filters.touch()
            return True

    def remove_batch(self, ids: List[str]) -> int:
        return sum(1 for id in ids if self.remove(id))

    def get(self, id: str) -> Optional[Tuple[np.ndarray, Dict[str, Any]]]:
        if id not in self._row_of:
            return None
        return self._decode(self._codes_arr[self._row_of[id]]), self._metadata.get(id, {})

    def contains(self, id: str) -> bool:
        return id in self._row_of

    def clear(self) -> int:
        with self._lock:
            count = self.size
            self._ids.clear(); self._row_of.clear(); self._metadata.clear()
            self._synced = -1
            self._filters.touch()
            return count

    def stats(self) -> IndexStats:
        code_mem = self.size * self.config.n_subvectors
        book_mem = self.config.n_subvectors * self._n_centroids * self._subvector_dim * 4
        orig = self.size * self._dimension * 4
        return IndexStats(index_type="pq", dimension=self._dimension, metric=self._metric, vector_count=self.size,
                          memory_bytes=code_mem + book_mem, is_trained=self._is_trained,
                          build_time_seconds=time.time() - self._build_start,
                          extra={"n_subvectors": self.config.n_subvectors, "n_bits": self.config.n_bits,
                                 "n_centroids": self._n_centroids,
                                 "compression_ratio": orig / max(1, code_mem + book_mem),
                                 "original_memory_bytes": orig})

    def to_dict(self) -> Dict[str, Any]:
        return {"index_type": "pq", "dimension": self._dimension, "metric": self._metric,
                "config": {"n_subvectors": self.config.n_subvectors, "n_bits": self.config.n_bits},
                "codebook": self._codebook.tolist() if self._codebook is not None else None,
                "codes": {id: self._codes_arr[r].tolist() for id, r in self._row_of.items()},
                "metadata": self._metadata, "ids": self._ids}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "PQIndex":
        cfg = data.get("config", {})
        index = cls(dimension=data["dimension"], metric=data["metric"], n_subvectors=cfg.get("n_subvectors", 8),
                    n_bits=cfg.get("n_bits", 8))
        if data.get("codebook"):
            index._set_codebook(np.array(data["codebook"], dtype=np.float32))
        ids = list(data.get("ids", []))
        codes = data.get("codes", {})
        index._reserve(len(ids))
        if ids:
            index._codes_arr[:len(ids)] = np.array([codes[id] for id in ids], dtype=np.uint8).reshape(len(ids), -1)
        index._row_of = dict(zip(ids, range(len(ids))))
        index._ids = ids
        index._metadata = dict(data.get("metadata", {}))
        index._synced = -1
        return index

    def iter_ids(self) -> Iterator[str]:
        return iter(self._ids)

    def iter_vectors(self) -> Iterator[Tuple[str, np.ndarray, Dict[str, Any]]]:
        for id in self._ids:
            yield id, self._decode(self._codes_arr[self._row_of[id]]), self._metadata.get(id, {})

    def __del__(self):
        dev = getattr(self, "_dev", None)
        if dev is not None:
            dev.close()
