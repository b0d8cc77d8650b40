"""HybridIndex (IVF-PQ, index/hybrid.py) on the MI355X search core.

Device work: coarse k-means (the IVF KMeans), nearest-centroid assignment,
residuals and their PQ codes in one call, and the search -- coarse ranking, the
table of every (query, probed list) residual and the ADC scan of the list's
codes, with the filter bit tested in the selection.  As in the reference the
``metric`` argument is accepted and ignored: distances are Euclidean.
"""

from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

from .. import _native as N
from .base import BaseIndex, FilterFunction, IndexConfig, IndexStats, IndexType, SearchResult
from .ivf import KMeans
from .pq import train_subquantizers


@dataclass
class HybridConfig(IndexConfig):                            # index/hybrid.py:24-38
    n_clusters: int = 100
    n_probe: int = 10
    n_subvectors: int = 8
    n_bits: int = 8
    n_iter: int = 20
    seed: Optional[int] = None


class HybridIndex(BaseIndex):
    """index/hybrid.py:41-536"""

    def __init__(self, dimension: int, metric: str = "euclidean", n_clusters: int = 100, n_probe: int = 10,
                 n_subvectors: int = 8, n_bits: int = 8, seed: Optional[int] = None, **kwargs):
        super().__init__(dimension, metric, **kwargs)
        if dimension % n_subvectors != 0:
            raise ValueError(f"Dimension {dimension} must be divisible by n_subvectors {n_subvectors}")
        if not 1 <= n_bits <= 8:
            raise ValueError(f"n_bits must be in 1..8 (codes are stored as uint8), got {n_bits}")
        self.config = HybridConfig(dimension=dimension, metric=metric, n_clusters=n_clusters, n_probe=n_probe,
                                   n_subvectors=n_subvectors, n_bits=n_bits, seed=seed)
        self._subvector_dim = dimension // n_subvectors
        self._n_centroids_pq = 2 ** n_bits
        self._coarse_centroids: Optional[np.ndarray] = None
        self._pq_codebook: Optional[np.ndarray] = None
        # rows in insertion order; a removed row leaves a hole (list == -1) until the lists are rebuilt
        self._ids: List[str] = []
        self._row_of: Dict[str, int] = {}
        self._row_ids: List[Optional[str]] = []
        self._list_of_row = np.zeros(1024, dtype=np.int32)
        self._codes_arr = np.zeros((1024, n_subvectors), dtype=np.uint8)
        self._metadata: Dict[str, Dict[str, Any]] = {}
        self._size = 0
        self._rng = np.random.RandomState(seed)
        self._build_start = time.time()
        # Device mirror: a device row number is a host row.  `_dirty` asks for a full build of the lists; otherwise
        # host rows [_synced, len(_row_ids)) are waiting to be appended to the tails of their lists.
        self._dev: Optional[N.IvfPqLists] = None
        self._dirty = True
        self._synced = 0

    @property
    def index_type(self) -> IndexType:
        return IndexType.IVF_PQ

    @property
    def size(self) -> int:
        return self._size

    def set_n_probe(self, n_probe: int) -> None:
        self.config.n_probe = min(n_probe, self.config.n_clusters)

    @property
    def _id_to_cluster(self) -> Dict[str, int]:
        return {id: int(self._list_of_row[r]) for id, r in self._row_of.items()}

    # ------------------------------------------------------------ training
    def train(self, vectors) -> None:
        """hybrid.py:126-203: coarse k-means, residuals, one 2**n_bits-means per sub-space
        of the residuals (random initial rows, n_iter Lloyd steps, empty clusters become
        zero rows, zero padding when there are fewer rows than codewords)."""
        vectors = self._validate_vectors(vectors)
        n = len(vectors)
        k_coarse = min(self.config.n_clusters, n)
        cent = KMeans(n_clusters=k_coarse, n_iter=self.config.n_iter, seed=self.config.seed).fit(vectors)
        self.config.n_clusters = len(cent)
        assign, _ = N.assign(vectors, cent, N.EUCLIDEAN_SQ)
        residuals = vectors - cent[assign]
        book = train_subquantizers(np.ascontiguousarray(residuals, dtype=np.float32), self.config.n_subvectors,
                                   self._n_centroids_pq, self.config.n_iter, self._rng)
        self._set_quantizers(N.f32(cent), book)

    def _set_quantizers(self, centroids: np.ndarray, codebook: np.ndarray) -> None:
        self._coarse_centroids, self._pq_codebook = N.f32(centroids), N.f32(codebook)
        self._is_trained = True
        if self._dev is not None:
            self._dev.close()
        self._dev = N.IvfPqLists(self._dimension, self.config.n_subvectors, self._n_centroids_pq)
        self._dev.set_quantizers(self._coarse_centroids, self._pq_codebook)
        self._dirty = True

    # ------------------------------------------------------------ add / remove
    def _reserve(self, n: int) -> None:
        if n > len(self._list_of_row):
            cap = max(n, 2 * len(self._list_of_row))
            lists = np.full(cap, -1, dtype=np.int32)
            lists[:len(self._row_ids)] = self._list_of_row[:len(self._row_ids)]
            codes = np.zeros((cap, self.config.n_subvectors), dtype=np.uint8)
            codes[:len(self._row_ids)] = self._codes_arr[:len(self._row_ids)]
            self._list_of_row, self._codes_arr = lists, codes

    def add(self, id: str, vector, metadata: Optional[Dict[str, Any]] = None) -> None:
        if not self._is_trained:
            raise RuntimeError("Index must be trained")
        vector = self._validate_vector(vector)
        if id in self._row_of:
            raise ValueError(f"ID '{id}' already exists")
        self.add_batch([id], vector[None, :], [metadata or {}])

    def add_batch(self, ids: List[str], vectors, metadata: Optional[List[Dict[str, Any]]] = None) -> int:
        """One device call assigns, subtracts and encodes the whole batch (the reference
        loops over ``add``, hybrid.py:274-282)."""
        vectors = self._validate_vectors(vectors)
        if not self._is_trained:
            raise RuntimeError("Index must be trained")
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
        lists, codes = self._dev.encode(vectors)
        with self._lock:
            base = len(self._row_ids)
            self._reserve(base + n)
            self._list_of_row[base:base + n] = lists
            self._codes_arr[base:base + n] = codes
            for i, id in enumerate(ids):
                self._row_of[id] = base + i
                self._row_ids.append(id)
                self._ids.append(id)
                self._metadata[id] = (metadata[i] if metadata else None) or {}
            self._size += n
            self._filters.touch()
        return n

    def remove(self, id: str) -> bool:
        """hybrid.py:396-415: the entry leaves its list, the others keep their order."""
        with self._lock:
            if id not in self._row_of:
                return False
            r = self._row_of.pop(id)
            if not self._dirty and r < self._synced:
                # its place on the device: rows of a list sit there in insertion order
                l = int(self._list_of_row[r])
                self._dev.remove_at(l, int(np.count_nonzero(self._list_of_row[:r] == l)))
            self._list_of_row[r] = -1
            self._row_ids[r] = None
            self._ids.remove(id)
            self._metadata.pop(id, None)
            self._size -= 1
            self._filters.touch()
            return True

    def remove_batch(self, ids: List[str]) -> int:
        return sum(1 for id in ids if self.remove(id))

    def clear(self) -> int:
        count = self._size
        self._ids.clear(); self._row_of.clear(); self._row_ids.clear(); self._metadata.clear()
        self._size = 0
        self._dirty = True
        self._filters.touch()
        return count

    # ------------------------------------------------------------ device lists
    def _sync(self) -> None:
        n_rows = len(self._row_ids)
        lists = self._list_of_row[:n_rows]
        if self._dirty:
            live = np.flatnonzero(lists >= 0)
            order = live[np.argsort(lists[live], kind="stable")]                  # grouped by list, insertion order inside
            offs = np.searchsorted(lists[order], np.arange(self.config.n_clusters + 1))
            self._dev.build_numbered(self._codes_arr[order], order, offs)
            self._dirty = False
        elif self._synced < n_rows:
            # rows added since the last search go behind the last codes of their lists (hybrid.py:262); rows that
            # were removed again before they were ever uploaded are skipped
            new = np.arange(self._synced, n_rows, dtype=np.int64)
            new = new[lists[new] >= 0]
            if len(new):
                try:
                    self._dev.append(self._codes_arr[new], new, lists[new])
                except Exception:
                    self._dirty = True      # the device lists are as they were: the next search builds them afresh
                    raise
                live, _positions, dead = self._dev.layout()
                if dead > live + 4096:      # more abandoned places than rows: pack the lists again at the next search
                    self._dirty = True
        self._synced = n_rows

    def _mirror_stale(self) -> bool:
        return self._dirty or self._synced < len(self._row_ids)

    def _sync_mirror(self) -> None:
        self._sync()

    def _mask_words(self, filter_fn: Optional[FilterFunction]):
        return self._filter_for_device(filter_fn, self._row_ids, lambda i: self._metadata.get(i, {}))

    # ------------------------------------------------------------ search
    def _results(self, dist_row, rows_row, include_vectors: bool, include_metadata: bool) -> List[SearchResult]:
        out = []
        for d, r in zip(dist_row, rows_row):
            if r < 0:
                break
            id = self._row_ids[int(r)]
            out.append(SearchResult(id=id, distance=d, score=1.0 / (1.0 + d),                    # hybrid.py:342-348
                                    vector=self._reconstruct(id) if include_vectors else None,
                                    metadata=self._metadata.get(id, {}) if include_metadata else None))
        return out

    def search(self, query, k: int = 10, filter_fn: Optional[FilterFunction] = None, include_vectors: bool = False,
               include_metadata: bool = True, n_probe: Optional[int] = None) -> List[SearchResult]:
        if not self._is_trained:
            raise RuntimeError("Index must be trained")
        if self._size == 0:
            return []
        query = self._validate_vector(query)
        return self._search_rows(query[None, :], k, filter_fn, n_probe, include_vectors, include_metadata)[0]

    def search_batch(self, queries, k: int = 10, filter_fn: Optional[FilterFunction] = None,
                     n_probe: Optional[int] = None) -> List[List[SearchResult]]:
        """One device call for the batch (the reference loops over ``search``, hybrid.py:375-386)."""
        if not self._is_trained:
            raise RuntimeError("Index must be trained")
        queries = self._validate_vectors(queries)
        if self._size == 0:
            return [[] for _ in range(len(queries))]
        return self._search_rows(queries, k, filter_fn, n_probe, False, True)

    def _search_rows(self, queries, k, filter_fn, n_probe, include_vectors, include_metadata):
        n_probe = n_probe or self.config.n_probe
        with self._searching():
            k_eff = min(k, self._size)
            if k_eff <= 0:
                return [[] for _ in range(len(queries))]
            dist, rows = self._dev.search(queries, k_eff, n_probe, self._mask_words(filter_fn))
            return [self._results(dist[i], rows[i], include_vectors, include_metadata) for i in range(len(queries))]

    # ------------------------------------------------------------ accessors
    def _reconstruct(self, id: str) -> np.ndarray:
        r = self._row_of[id]
        codes = self._codes_arr[r]
        residual = np.concatenate([self._pq_codebook[j, codes[j]] for j in range(self.config.n_subvectors)])
        return (self._coarse_centroids[self._list_of_row[r]] + residual).astype(np.float32)

    def get(self, id: str) -> Optional[Tuple[np.ndarray, Dict[str, Any]]]:
        if id not in self._row_of:
            return None
        return self._reconstruct(id), self._metadata.get(id, {})

    def contains(self, id: str) -> bool:
        return id in self._row_of

    def stats(self) -> IndexStats:
        code_memory = self._size * self.config.n_subvectors
        codebook_memory = self.config.n_subvectors * self._n_centroids_pq * self._subvector_dim * 4
        centroid_memory = self.config.n_clusters * self._dimension * 4
        total = code_memory + codebook_memory + centroid_memory
        return IndexStats(index_type="ivf_pq", dimension=self._dimension, metric=self._metric, vector_count=self._size,
                          memory_bytes=total, is_trained=self._is_trained,
                          build_time_seconds=time.time() - self._build_start,
                          extra={"n_clusters": self.config.n_clusters, "n_probe": self.config.n_probe,
                                 "n_subvectors": self.config.n_subvectors, "n_bits": self.config.n_bits,
                                 "compression_ratio": self._size * self._dimension * 4 / max(1, total)})

    def to_dict(self) -> Dict[str, Any]:
        lists: Dict[str, list] = {str(l): [] for l in range(self.config.n_clusters if self._is_trained else 0)}
        for r, id in enumerate(self._row_ids):
            if id is not None:
                lists[str(int(self._list_of_row[r]))].append(
                    {"id": id, "codes": self._codes_arr[r].tolist(), "metadata": self._metadata.get(id, {})})
        return {"index_type": "ivf_pq", "dimension": self._dimension, "metric": self._metric,
                "config": {"n_clusters": self.config.n_clusters, "n_probe": self.config.n_probe,
                           "n_subvectors": self.config.n_subvectors, "n_bits": self.config.n_bits},
                "coarse_centroids": self._coarse_centroids.tolist() if self._coarse_centroids is not None else None,
                "pq_codebook": self._pq_codebook.tolist() if self._pq_codebook is not None else None,
                "inverted_lists": lists}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "HybridIndex":
        cfg = data.get("config", {})
        index = cls(dimension=data["dimension"], metric=data["metric"], n_clusters=cfg.get("n_clusters", 100),
                    n_probe=cfg.get("n_probe", 10), n_subvectors=cfg.get("n_subvectors", 8), n_bits=cfg.get("n_bits", 8))
        if data.get("coarse_centroids") and data.get("pq_codebook"):
            index._set_quantizers(np.array(data["coarse_centroids"], dtype=np.float32),
                                  np.array(data["pq_codebook"], dtype=np.float32))
        ids, meta, lists, codes = [], [], [], []
        for l_str, entries in data.get("inverted_lists", {}).items():
            for e in entries:
                ids.append(e["id"]); meta.append(e.get("metadata", {})); lists.append(int(l_str)); codes.append(e["codes"])
        n = len(ids)
        if n:
            index._reserve(n)
            index._list_of_row[:n] = np.asarray(lists, dtype=np.int32)
            index._codes_arr[:n] = np.array(codes, dtype=np.uint8).reshape(n, index.config.n_subvectors)
            index._row_of = dict(zip(ids, range(n)))
            index._row_ids = list(ids)
            index._ids = list(ids)
            index._metadata = dict(zip(ids, meta))
            index._size = n
        index._dirty = True
        return index

    def iter_ids(self) -> Iterator[str]:
        return iter(self._ids)

    def iter_vectors(self) -> Iterator[Tuple[str, np.ndarray, Dict[str, Any]]]:
        for id in self._ids:
            got = self.get(id)
            if got:
                yield id, got[0], got[1]

    def __del__(self):
        dev = getattr(self, "_dev", None)
        if dev is not None:
            dev.close()
