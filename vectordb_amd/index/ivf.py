"""KMeans and IVFIndex (index/ivf.py) on the MI355X search core.

Device work: nearest-centroid assignment (training, ``add``/``add_batch``),
the per-cluster sums of a Lloyd step, probe selection and the scan of the
probed inverted lists with the filter bit tested in the scan.  Host work: the
RandomState draw sequence of k-means++ and of empty-cluster re-seeding (kept
identical to the reference so equal seeds pick equal rows), id/metadata
bookkeeping, and the division sums/counts of k x d numbers.

Inverted lists are kept on the host as lists of slots into a growable fp32
matrix and are laid out contiguously in list order (CSR) in HBM when the
index is searched after a change.
"""

from __future__ import annotations

import threading
import time
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

from .. import _native as N
from ..distance.registry import canonical_metric, get_metric
from .base import (BaseIndex, FilterFunction, IndexConfig, IndexStats, IndexType, SearchResult, unit_rows,
                   unit_vector)

_CODES = {"euclidean": N.EUCLIDEAN, "euclidean_squared": N.EUCLIDEAN_SQ, "cosine": N.COSINE, "dot": N.DOT,
          "manhattan": N.MANHATTAN, "chebyshev": N.CHEBYSHEV, "hamming": N.HAMMING, "angular": N.ANGULAR}
# per-pair scalar forms behind get_metric_fn (distance/registry.py:95-177)
_PAIR_CODES = {"euclidean": N.EUCLIDEAN_DIRECT, "euclidean_squared": N.EUCLIDEAN_SQ_DIRECT, "cosine": N.COSINE_PAIR,
               "dot": N.DOT, "manhattan": N.MANHATTAN, "chebyshev": N.CHEBYSHEV, "hamming": N.HAMMING,
               "angular": N.ANGULAR}


@dataclass
class IVFConfig(IndexConfig):                              # index/ivf.py:64-99
    n_clusters: int = 100
    n_probe: int = 10
    n_iter: int = 20
    n_redo: int = 1
    max_points_per_centroid: int = 256
    min_points_per_centroid: int = 1
    normalize: bool = False
    seed: Optional[int] = None
    verbose: bool = False

    def __post_init__(self):
        super().validate()
        if self.n_clusters < 1:
            raise ValueError(f"n_clusters must be >= 1, got {self.n_clusters}")
        if self.n_probe < 1:
            raise ValueError(f"n_probe must be >= 1, got {self.n_probe}")
        if self.n_probe > self.n_clusters:
            self.n_probe = self.n_clusters


class KMeans:
    """index/ivf.py:102-237 (always Euclidean)."""

    def __init__(self, n_clusters: int, n_iter: int = 20, n_redo: int = 1, seed: Optional[int] = None,
                 verbose: bool = False):
        self.n_clusters = n_clusters
        self.n_iter = n_iter
        self.n_redo = n_redo
        self.seed = seed
        self.verbose = verbose
        self.centroids: Optional[np.ndarray] = None
        self._rng = np.random.RandomState(seed)

    def fit(self, vectors) -> np.ndarray:
        vectors = N.f32(vectors)
        n = len(vectors)
        if n < self.n_clusters:                               # ivf.py:138-141
            self.centroids = vectors.copy()
            return self.centroids
        # the training rows go to the device once; every step of every run works on them there
        self.phase_seconds = {"upload": 0.0, "init_pp": 0.0, "lloyd": 0.0, "lloyd_steps": 0}   # of the last fit
        store = N.RowStore(vectors.shape[1])
        try:
            t0 = time.perf_counter()
            store.append(vectors)
            self.phase_seconds["upload"] = time.perf_counter() - t0
            best, best_inertia = None, float("inf")
            for redo in range(self.n_redo):
                trainer = N.KMeansTrainer(store, 0, vectors.shape[1], self.n_clusters)
                try:
                    cent = self._fit_once(vectors, trainer)
                    if self.n_redo > 1 or self.verbose:
                        _, d = trainer.assignments(with_distances=True)       # ivf.py:150-153
                        inertia = float(np.sum(d.astype(np.float64) ** 2))
                    else:
                        inertia = 0.0                  # a single run is the best run; the reference's loop only ranks runs
                finally:
                    trainer.close()
                if inertia < best_inertia:
                    best, best_inertia = cent, inertia
                if self.verbose:
                    print(f"  K-Means redo {redo + 1}/{self.n_redo}: inertia={inertia:.2f}")
        finally:
            store.close()
        self.centroids = best
        return self.centroids

    def _with_trainer(self, vectors, trainer, body):
        """Run body(trainer) on a trainer over `vectors`: the caller's, or one opened (and closed) here."""
        if trainer is not None:
            return body(trainer)
        vectors = N.f32(vectors)
        store = N.RowStore(vectors.shape[1])
        try:
            store.append(vectors)
            own = N.KMeansTrainer(store, 0, vectors.shape[1], self.n_clusters)
            try:
                return body(own)
            finally:
                own.close()
        finally:
            store.close()

    def _fit_once(self, vectors, trainer: Optional["N.KMeansTrainer"] = None) -> np.ndarray:
        """ivf.py:165-202 on resident rows.  Host side: the RandomState draws, in the reference's order
        (randint for the first centroid, one uniform per rng.choice of k-means++, randint per empty
        cluster in ascending cluster order), and the stopping test on one number per step."""
        n = len(vectors)
        if not hasattr(self, "phase_seconds"):
            self.phase_seconds = {"upload": 0.0, "init_pp": 0.0, "lloyd": 0.0, "lloyd_steps": 0}

        def body(tr):
            t0 = time.perf_counter()
            self._init_centroids_pp(vectors, tr)
            t1 = time.perf_counter()
            self.phase_seconds["init_pp"] += t1 - t0
            try:
                for it in range(self.n_iter):
                    self.phase_seconds["lloyd_steps"] += 1
                    counts, diff = tr.step(N.KMEANS_IVF)
                    empty = np.flatnonzero(counts == 0)
                    if len(empty):                                             # ivf.py:185-190
                        diff = tr.reseed(empty, [self._rng.randint(n) for _ in empty])
                    tr.commit()
                    if self.verbose and (it + 1) % 5 == 0:
                        print(f"    Iteration {it + 1}/{self.n_iter}: diff={diff:.6f}")
                    if diff < 1e-6:
                        break
                return tr.centroids()
            finally:
                self.phase_seconds["lloyd"] += time.perf_counter() - t1

        return self._with_trainer(vectors, trainer, body)

    def _init_centroids_pp(self, vectors, trainer: Optional["N.KMeansTrainer"] = None) -> np.ndarray:
        """k-means++ (ivf.py:204-226) with the reference's draw order.  rng.choice(n, p=probs) draws one
        uniform number and takes the first row whose cumulative probability exceeds it, so the k - 1
        uniforms are drawn up front (same generator state afterwards) and the device keeps the running
        minimum of the squared distances and does every draw itself: no row of numbers comes back."""
        n = len(vectors)

        def body(tr):
            first = self._rng.randint(n)
            uniforms = self._rng.random_sample(self.n_clusters - 1)
            try:
                tr.init_pp(first, uniforms)
            except N.NativeError as e:
                if "probabilities contain NaN" in str(e):         # what rng.choice raises there (ivf.py:224)
                    raise ValueError("probabilities contain NaN") from e
                raise
            return tr.centroids()

        return self._with_trainer(vectors, trainer, body)

    def _assign(self, vectors, centroids) -> np.ndarray:
        idx, _ = N.assign(vectors, centroids, N.EUCLIDEAN)    # ivf.py:228-231
        return idx.astype(np.int64)

    def predict(self, vectors) -> np.ndarray:
        if self.centroids is None:
            raise RuntimeError("K-Means not fitted")
        return self._assign(vectors, self.centroids)


class IVFIndex(BaseIndex):
    """index/ivf.py:240-1061"""

    def __init__(self, dimension: int, metric: str = "euclidean", n_clusters: int = 100, n_probe: int = 10,
                 n_iter: int = 20, normalize: bool = False, seed: Optional[int] = None, verbose: bool = False,
                 prefilter: Optional[str] = "auto", **kwargs):
        """``prefilter`` is not a reference argument: batched Euclidean searches scan a bf16 copy of the member rows on
        the matrix cores and recompute the few remaining candidates in the reference's arithmetic; results are bit for
        bit those of the plain scan.  "auto" (default) keeps the copy while it leaves a tenth of the device memory
        free, "bf16" always, None never."""
        super().__init__(dimension, metric, **kwargs)
        if prefilter not in N.PREFILTER:
            raise ValueError(f"prefilter must be None, 'bf16' or 'auto', got {prefilter!r}")
        self._prefilter = prefilter
        self.config = IVFConfig(dimension=dimension, metric=metric, n_clusters=n_clusters, n_probe=n_probe,
                                n_iter=n_iter, normalize=normalize, seed=seed, verbose=verbose)
        get_metric(metric)                                     # KeyError for unknown metrics (ivf.py:332)
        self._canon = canonical_metric(metric)
        if self._canon not in _CODES:
            raise KeyError(f"Unknown metric: '{metric}'. Available: {list(_CODES)}")
        self._centroids: Optional[np.ndarray] = None
        self._lists: Dict[int, List[int]] = {}                 # cluster -> slots in scan order
        self._vectors = np.zeros((1024, dimension), dtype=np.float32)
        self._n_slots = 0
        self._slot_id: List[Optional[str]] = []
        self._slot_of: Dict[str, int] = {}
        self._metadata: Dict[str, Dict[str, Any]] = {}
        self._id_to_cluster: Dict[str, int] = {}
        self._id_to_position: Dict[str, int] = {}
        self._build_start = time.time()
        self._n_distance_computations = 0
        self._stats_mu = threading.Lock()
        self._rng = np.random.RandomState(seed)
        # Device mirror.  A device row number is a slot of the host table.  After the first build, adds go to the
        # tails of their lists (`_pending` until the next search) and removes swap inside one list, as the host
        # lists do; `_dev_dirty` asks for a full rebuild (new centroids, clear, or too many abandoned places).
        self._dev: Optional[N.IvfLists] = None
        self._dev_dirty = True
        self._pending: List[int] = []                          # slots added since the device was last brought up to date

    # ------------------------------------------------------------ properties
    @property
    def index_type(self) -> IndexType:
        return IndexType.IVF

    @property
    def size(self) -> int:
        return len(self._slot_of)

    @property
    def n_clusters(self) -> int:
        return self.config.n_clusters

    @property
    def n_probe(self) -> int:
        return self.config.n_probe

    @property
    def centroids(self) -> Optional[np.ndarray]:
        return self._centroids

    def set_n_probe(self, n_probe: int) -> None:
        if n_probe < 1:
            raise ValueError(f"n_probe must be >= 1, got {n_probe}")
        self.config.n_probe = min(n_probe, self.config.n_clusters)

    def is_trained(self) -> bool:                              # a method here, as ivf.py:432-434
        return self._is_trained and self._centroids is not None

    # ------------------------------------------------------------ training
    def train(self, vectors) -> None:
        vectors = self._validate_vectors(vectors)
        if self.config.normalize:
            vectors = unit_rows(vectors)
        n = len(vectors)
        actual = min(self.config.n_clusters, n)
        if self.config.verbose:
            print(f"Training IVF with {actual} clusters on {n} vectors...")
        km = KMeans(n_clusters=actual, n_iter=self.config.n_iter, n_redo=self.config.n_redo, seed=self.config.seed,
                    verbose=self.config.verbose)
        t0 = time.time()
        self._last_kmeans = km                                 # its phase_seconds describe this training
        self._set_centroids(km.fit(vectors))
        if self.config.verbose:
            print(f"Training completed in {time.time() - t0:.2f}s")

    def _set_centroids(self, centroids: np.ndarray) -> None:
        self._centroids = N.f32(centroids)
        self.config.n_clusters = len(self._centroids)
        self.config.n_probe = min(self.config.n_probe, self.config.n_clusters)
        self._lists = {i: [] for i in range(self.config.n_clusters)}
        self._is_trained = True
        self._dev_dirty = True
        self._filters.touch()
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    # ------------------------------------------------------------ add
    def _new_slot(self, vector: np.ndarray) -> int:
        if self._n_slots >= len(self._vectors):
            grown = np.zeros((max(2 * len(self._vectors), 1024), self._dimension), dtype=np.float32)
            grown[:self._n_slots] = self._vectors[:self._n_slots]
            self._vectors = grown
        self._vectors[self._n_slots] = vector
        self._n_slots += 1
        return self._n_slots - 1

    def _append(self, id: str, vector: np.ndarray, metadata, cluster: int) -> None:
        slot = self._new_slot(vector)
        self._slot_id.append(id)
        self._slot_of[id] = slot
        self._metadata[id] = metadata
        self._id_to_cluster[id] = cluster
        self._id_to_position[id] = len(self._lists[cluster])
        self._lists[cluster].append(slot)
        self._filters.touch()
        if not self._dev_dirty:
            self._pending.append(slot)

    def add(self, id: str, vector, metadata: Optional[Dict[str, Any]] = None) -> None:
        if not self.is_trained():
            raise RuntimeError("Index must be trained before adding vectors")
        vector = self._validate_vector(vector)
        if self.config.normalize:
            vector = unit_vector(vector)
        with self._lock:
            if id in self._slot_of:
                raise ValueError(f"ID '{id}' already exists in index")
            self._append(id, vector, metadata or {}, self._find_nearest_centroid(vector))

    def add_batch(self, ids: List[str], vectors, metadata: Optional[List[Dict[str, Any]]] = None) -> int:
        if not self.is_trained():
            raise RuntimeError("Index must be trained before adding vectors")
        vectors = self._validate_vectors(vectors)
        n = len(ids)
        if len(vectors) != n:
            raise ValueError(f"Number of ids ({n}) != vectors ({len(vectors)})")
        if metadata is not None and len(metadata) != n:
            raise ValueError(f"Number of ids ({n}) != metadata ({len(metadata)})")
        if self.config.normalize:
            vectors = unit_rows(vectors)
        with self._lock:
            seen = set()
            for id in ids:
                if id in self._slot_of or id in seen:
                    raise ValueError(f"ID '{id}' already exists in index")
                seen.add(id)
            clusters = self._find_nearest_centroids_batch(vectors)
            for i in range(n):
                self._append(ids[i], vectors[i], metadata[i] if metadata else {}, int(clusters[i]))
        return n

    def _find_nearest_centroid(self, vector: np.ndarray) -> int:
        """ivf.py:546-551: the per-pair scalar form of the metric."""
        idx, _ = N.assign(vector[None, :], self._centroids, _PAIR_CODES[self._canon])
        return int(idx[0])

    def _find_nearest_centroids_batch(self, vectors: np.ndarray) -> np.ndarray:
        """ivf.py:553-565: identity-form matrices for euclidean and cosine, the
        scalar form otherwise."""
        if len(vectors) == 0:
            return np.zeros(0, dtype=np.int64)
        code = _CODES[self._canon] if self._metric in ("euclidean", "cosine") else _PAIR_CODES[self._canon]
        idx, _ = N.assign(vectors, self._centroids, code)
        return idx.astype(np.int64)

    # ------------------------------------------------------------ device layout
    def _device(self) -> N.IvfLists:
        if self._dev is None:
            self._dev = N.IvfLists(self._dimension, _CODES[self._canon])
            self._dev.set_prefilter(self._prefilter)
            self._dev.set_centroids(self._centroids)
            self._dev_dirty = True
        if self._dev_dirty:
            k = self.config.n_clusters
            sizes = np.array([len(self._lists[c]) for c in range(k)], dtype=np.int64)
            offsets = np.zeros(k + 1, dtype=np.int64)
            np.cumsum(sizes, out=offsets[1:])
            order = np.fromiter((s for c in range(k) for s in self._lists[c]), dtype=np.int64, count=int(offsets[-1]))
            rows = self._vectors[order] if len(order) else np.zeros((0, self._dimension), np.float32)
            self._dev.build_numbered(rows, order, offsets)
            self._pending = []
            self._dev_dirty = False
        else:
            self._flush_pending()
        return self._dev

    def _flush_pending(self) -> None:
        """Rows added since the last search go behind the last rows of their lists on the device (ivf.py:478-542
        appends to the Python list); nothing else of the index is uploaded again."""
        if not self._pending or self._dev is None or self._dev_dirty:
            return
        slots = np.asarray(self._pending, dtype=np.int64)
        lists = np.fromiter((self._id_to_cluster[self._slot_id[s]] for s in self._pending), dtype=np.int32, count=len(slots))
        try:
            self._dev.append(self._vectors[slots], slots, lists)
        except Exception:
            self._dev_dirty = True          # the device lists are as they were: the next search builds them afresh
            self._pending = []
            raise
        self._pending = []
        live, _positions, dead = self._dev.layout()
        if dead > live + 4096:              # more abandoned places than rows: pack the lists again at the next search
            self._dev_dirty = True

    def _mirror_stale(self) -> bool:
        return self._dev is None or self._dev_dirty or bool(self._pending)

    def _sync_mirror(self) -> None:
        self._device()

    # ------------------------------------------------------------ search
    def _mask_words(self, filter_fn: Optional[FilterFunction]):
        return self._filter_for_device(filter_fn, self._slot_id, lambda i: self._metadata.get(i, {}))

    def _results(self, dist_row, rows_row, include_vectors: bool, include_metadata: bool) -> List[SearchResult]:
        out = []
        for d, r in zip(dist_row, rows_row):
            if r < 0:
                break
            id = self._slot_id[r]
            d = float(d)
            out.append(SearchResult(id=id, distance=d, score=1.0 / (1.0 + d),
                                    vector=self._vectors[self._slot_of[id]].copy() if include_vectors else None,
                                    metadata=self._metadata.get(id, {}).copy() if include_metadata else None))
        return out

    def search(self, query, k: int = 10, filter_fn: Optional[FilterFunction] = None, include_vectors: bool = False,
               include_metadata: bool = True, n_probe: Optional[int] = None) -> List[SearchResult]:
        if not self.is_trained():
            raise RuntimeError("Index must be trained before searching")
        if self.size == 0:
            return []
        query = self._validate_vector(query)
        if self.config.normalize:
            query = unit_vector(query)
        n_probe = min(n_probe or self.config.n_probe, self.config.n_clusters)      # ivf.py:607-608
        # device call and row -> id mapping under a shared hold of the index lock (a concurrent write swaps the
        # device row numbering; other searches may run)
        with self._searching():
            k = min(k, self.size)
            if k <= 0:
                return []
            dev = self._dev
            dist, rows, scanned = dev.search(query, k, n_probe, self._mask_words(filter_fn), with_scan_rows=True)
            with self._stats_mu:
                self._n_distance_computations += scanned
            return self._results(dist[0], rows[0], include_vectors, include_metadata)

    def search_batch(self, queries, k: int = 10, filter_fn: Optional[FilterFunction] = None,
                     n_probe: Optional[int] = None) -> List[List[SearchResult]]:
        """One device call for the batch (the reference loops, ivf.py:669-673)."""
        if not self.is_trained():
            raise RuntimeError("Index must be trained before searching")
        queries = self._validate_vectors(queries)
        if self.size == 0 or len(queries) == 0:
            return [[] for _ in range(len(queries))]
        if self.config.normalize:
            queries = unit_rows(queries)
        n_probe = min(n_probe or self.config.n_probe, self.config.n_clusters)
        with self._searching():
            k = min(k, self.size)
            if k <= 0:
                return [[] for _ in range(len(queries))]
            dev = self._dev
            dist, rows, scanned = dev.search(queries, k, n_probe, self._mask_words(filter_fn), with_scan_rows=True)
            with self._stats_mu:
                self._n_distance_computations += scanned
            return [self._results(dist[i], rows[i], False, True) for i in range(len(queries))]

    def search_with_scores(self, query, k: int = 10, n_probe: Optional[int] = None) -> Tuple[List[str], List[float]]:
        res = self.search(query, k=k, n_probe=n_probe, include_vectors=False, include_metadata=False)
        return [r.id for r in res], [r.distance for r in res]

    # ------------------------------------------------------------ remove / get
    def remove(self, id: str) -> bool:
        """Swap with the last entry of the list (ivf.py:708-745)."""
        with self._lock:
            if id not in self._slot_of:
                return False
            cluster, pos = self._id_to_cluster[id], self._id_to_position[id]
            lst = self._lists[cluster]
            if self._dev is not None and not self._dev_dirty:
                self._flush_pending()                  # the device list must hold the row before it can swap it out
                if not self._dev_dirty:
                    self._dev.remove_at(cluster, pos)
            if pos < len(lst) - 1:
                lst[pos] = lst[-1]
                self._id_to_position[self._slot_id[lst[pos]]] = pos
            lst.pop()
            self._slot_id[self._slot_of[id]] = None
            del self._slot_of[id], self._metadata[id], self._id_to_cluster[id], self._id_to_position[id]
            self._filters.touch()
            return True

    def remove_batch(self, ids: List[str]) -> int:
        return sum(1 for id in ids if self.remove(id))

    def get(self, id: str) -> Optional[Tuple[np.ndarray, Dict[str, Any]]]:
        if id not in self._slot_of:
            return None
        return self._vectors[self._slot_of[id]].copy(), self._metadata[id].copy()

    def contains(self, id: str) -> bool:
        return id in self._slot_of

    def clear(self) -> int:
        with self._lock:
            count = self.size
            for c in self._lists:
                self._lists[c] = []
            self._slot_of.clear(); self._metadata.clear(); self._id_to_cluster.clear(); self._id_to_position.clear()
            self._slot_id = []
            self._n_slots = 0
            self._dev_dirty = True
            self._filters.touch()
            return count

    def reset(self) -> None:
        self.clear()
        self._centroids = None
        self._is_trained = False
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    # ------------------------------------------------------------ stats
    def stats(self) -> IndexStats:
        sizes = [len(self._lists.get(i, [])) for i in range(self.config.n_clusters)]
        meta_mem = sum(len(str(m)) for m in self._metadata.values())
        return IndexStats(
            index_type=self.index_type.value, dimension=self._dimension, metric=self._metric, vector_count=self.size,
            memory_bytes=self.size * self._dimension * 4 + self.config.n_clusters * self._dimension * 4 + meta_mem,
            is_trained=self._is_trained, build_time_seconds=time.time() - self._build_start,
            extra={"n_clusters": self.config.n_clusters, "n_probe": self.config.n_probe,
                   "cluster_sizes": {"min": min(sizes) if sizes else 0, "max": max(sizes) if sizes else 0,
                                     "mean": np.mean(sizes) if sizes else 0, "std": np.std(sizes) if sizes else 0},
                   "empty_clusters": sum(1 for s in sizes if s == 0),
                   "distance_computations": self._n_distance_computations})

    def get_cluster_info(self) -> Dict[str, Any]:
        info = [{"cluster_id": c, "size": len(self._lists.get(c, []))} for c in range(self.config.n_clusters)]
        info.sort(key=lambda x: x["size"], reverse=True)
        sizes = [c["size"] for c in info]
        return {"n_clusters": self.config.n_clusters, "total_vectors": self.size, "cluster_sizes": info,
                "size_distribution": {"min": min(sizes) if sizes else 0, "max": max(sizes) if sizes else 0,
                                      "mean": np.mean(sizes) if sizes else 0, "std": np.std(sizes) if sizes else 0,
                                      "median": np.median(sizes) if sizes else 0},
                "empty_clusters": sum(1 for s in sizes if s == 0),
                "imbalance_ratio": max(sizes) / (np.mean(sizes) + 1e-8) if sizes else 0}

    def get_vectors_in_cluster(self, cluster_id: int) -> List[str]:
        if cluster_id < 0 or cluster_id >= self.config.n_clusters:
            raise ValueError(f"Invalid cluster_id: {cluster_id}")
        return [self._slot_id[s] for s in self._lists[cluster_id]]

    # ------------------------------------------------------------ serialisation
    def to_dict(self) -> Dict[str, Any]:
        lists = {str(c): [{"id": self._slot_id[s], "vector": self._vectors[s].tolist(),
                           "metadata": self._metadata[self._slot_id[s]]} for s in slots]
                 for c, slots in self._lists.items()}
        return {"index_type": self.index_type.value, "dimension": self._dimension, "metric": self._metric,
                "config": {"n_clusters": self.config.n_clusters, "n_probe": self.config.n_probe,
                           "n_iter": self.config.n_iter, "normalize": self.config.normalize, "seed": self.config.seed,
                           "prefilter": self._prefilter},
                "is_trained": self._is_trained,
                "centroids": self._centroids.tolist() if self._centroids is not None else None,
                "inverted_lists": lists}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "IVFIndex":
        cfg = data.get("config", {})
        index = cls(dimension=data["dimension"], metric=data["metric"], n_clusters=cfg.get("n_clusters", 100),
                    n_probe=cfg.get("n_probe", 10), n_iter=cfg.get("n_iter", 20), normalize=cfg.get("normalize", False),
                    seed=cfg.get("seed"), prefilter=cfg.get("prefilter", "auto"))
        if data.get("centroids") is not None:
            index._set_centroids(np.array(data["centroids"], dtype=np.float32))
            index._is_trained = data.get("is_trained", True)
        # every list's entries in one pass: one matrix for the rows, one call for the bookkeeping
        ids, meta, clusters, vecs = [], [], [], []
        for c_str, entries in data.get("inverted_lists", {}).items():
            for e in entries:
                ids.append(e["id"]); meta.append(e.get("metadata", {})); clusters.append(int(c_str)); vecs.append(e["vector"])
        if ids:
            index._bulk_load(ids, np.array(vecs, dtype=np.float32).reshape(len(ids), data["dimension"]), meta,
                             np.asarray(clusters, dtype=np.int64))
        return index

    def _bulk_load(self, ids: List[str], vectors: np.ndarray, metadata: List[Dict[str, Any]], clusters: np.ndarray) -> None:
        """Rows whose lists are known (a saved index): the host table takes the matrix as it is and every list gets
        its slots in the order given, without a pass through ``_append`` per row."""
        n = len(ids)
        with self._lock:
            base = self._n_slots
            if base + n > len(self._vectors):
                grown = np.zeros((max(2 * len(self._vectors), base + n), self._dimension), dtype=np.float32)
                grown[:base] = self._vectors[:base]
                self._vectors = grown
            self._vectors[base:base + n] = vectors
            self._n_slots = base + n
            self._slot_id.extend(ids)
            slots = range(base, base + n)
            self._slot_of.update(zip(ids, slots))
            self._metadata.update(zip(ids, metadata))
            cl = clusters.tolist()
            self._id_to_cluster.update(zip(ids, cl))
            for id, slot, c in zip(ids, slots, cl):
                lst = self._lists[c]
                self._id_to_position[id] = len(lst)
                lst.append(slot)
            self._dev_dirty = True
            self._pending = []
            self._filters.touch()

    def iter_ids(self) -> Iterator[str]:
        return iter(list(self._slot_of))

    def iter_vectors(self) -> Iterator[Tuple[str, np.ndarray, Dict[str, Any]]]:
        for c in range(self.config.n_clusters):
            for s in self._lists.get(c, []):
                id = self._slot_id[s]
                yield id, self._vectors[s].copy(), self._metadata[id].copy()

    # ------------------------------------------------------------ utilities
    def rebuild(self, vectors=None) -> None:
        """ivf.py:992-1027"""
        if self.size == 0 and vectors is None:
            return
        items = list(self.iter_vectors())
        all_vectors = np.array([v for _, v, _ in items], dtype=np.float32) if items else None
        self.reset()
        training = vectors if vectors is not None else all_vectors
        if training is not None and len(training) > 0:
            self.train(training)
        if all_vectors is not None and len(all_vectors) > 0:
            # vectors were normalised when first added
            keep, self.config.normalize = self.config.normalize, False
            try:
                self.add_batch([i for i, _, _ in items], all_vectors, [m for _, _, m in items])
            finally:
                self.config.normalize = keep

    def assign_cluster(self, vector) -> int:
        if not self.is_trained():
            raise RuntimeError("Index must be trained")
        return self._find_nearest_centroid(self._validate_vector(vector))

    def get_centroid(self, cluster_id: int) -> np.ndarray:
        if not self.is_trained():
            raise RuntimeError("Index must be trained")
        if cluster_id < 0 or cluster_id >= self.config.n_clusters:
            raise ValueError(f"Invalid cluster_id: {cluster_id}")
        return self._centroids[cluster_id].copy()

    def __del__(self):
        dev = getattr(self, "_dev", None)
        if dev is not None:
            dev.close()
        filters = getattr(self, "_filters", None)
        if filters is not None:
            filters.drop()
