"""``Collection.search(query, k, filter)`` -- entry point (A) of the reference's
hot path (core/collection.py:693-807) -- on the MI355X search core.

Only what the search path needs is here: records with ids and metadata, the
lazily rebuilt dense matrix (core/collection.py:809-828) mirrored in HBM, the
metadata filter turned into a row bitmask, and the reference's result and
score conventions.  CRUD bookkeeping beyond that, persistence and the query
engine stay with the reference.
"""

from __future__ import annotations

import time
import uuid
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Union

import numpy as np

from .. import _native as N
from ..distance.metrics import _QUERY_METRICS


class VectorExistsError(Exception):
    """core/exceptions.py: a vector with the given id already exists."""


@dataclass
class SearchResult:                                        # core/collection.py:103-123
    id: str
    distance: float
    score: float
    vector: Optional[np.ndarray] = None
    metadata: Dict[str, Any] = field(default_factory=dict)

    def to_dict(self, include_vector: bool = False) -> Dict[str, Any]:
        out = {"id": self.id, "distance": self.distance, "score": self.score, "metadata": self.metadata}
        if include_vector and self.vector is not None:
            out["vector"] = self.vector.tolist()
        return out


def _validate_vector(vector, expected_dimension: int) -> np.ndarray:
    """core/vector.py:364-414"""
    if not isinstance(vector, np.ndarray):
        vector = np.array(vector, dtype=np.float32)
    if vector.dtype != np.float32:
        vector = vector.astype(np.float32)
    if vector.ndim != 1:
        raise ValueError(f"Vector must be 1D, got {vector.ndim}D")
    if len(vector) == 0:
        raise ValueError("Vector cannot be empty")
    if len(vector) != expected_dimension:
        raise ValueError(f"Vector dimension {len(vector)} doesn't match expected {expected_dimension}")
    if not np.isfinite(vector).all():
        raise ValueError("Vector contains NaN or Inf values")
    return vector


def _normalize_vector(vector: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """utils/normalization.py:9-33"""
    vector = vector.copy()
    norm = np.linalg.norm(vector)
    if norm < eps:
        return vector
    vector /= norm
    return vector


class MetadataFilter:
    """Row predicate with the semantics of MetadataIndex.query
    (core/collection.py:183-282): AND over fields; a plain value (or list of
    values) means equality with any of them; operator dicts support $in, $nin,
    $exists, $gt, $gte, $lt, $lte, $ne, and the LAST operator of a dict decides
    (each branch there overwrites or extends one shared set); a record's list
    value matches through any of its elements; records lacking the field never
    match."""

    def __init__(self, filters: Dict[str, Any]):
        self.filters = filters

    @staticmethod
    def _values(v):
        return v if isinstance(v, list) else [v]

    @staticmethod
    def _cmp(a, op, b) -> bool:
        try:
            if op == "$gt":
                return a > b
            if op == "$gte":
                return a >= b
            if op == "$lt":
                return a < b
            if op == "$lte":
                return a <= b
            if op == "$ne":
                return a != b
        except TypeError:
            return False
        return False

    def _field_matches(self, have, cond) -> bool:
        vals = [tuple(v) if isinstance(v, list) else v for v in self._values(have)]
        if not isinstance(cond, dict):
            wanted = [tuple(v) if isinstance(v, list) else v for v in self._values(cond)]
            return any(v in wanted for v in vals)
        matched = False
        for op, arg in cond.items():
            if op == "$in":
                matched = matched or any(v in list(arg) for v in vals)
            elif op == "$nin":
                matched = not any(v in list(arg) for v in vals)
            elif op == "$exists":
                matched = matched or bool(arg)
            elif op in ("$gt", "$gte", "$lt", "$lte", "$ne"):
                matched = matched or any(self._cmp(v, op, arg) for v in vals)
        return matched

    def __call__(self, metadata: Dict[str, Any]) -> bool:
        if not self.filters:
            return False
        for key, cond in self.filters.items():
            if key not in metadata:
                return False
            if isinstance(metadata[key], list) and not metadata[key]:
                return False            # nothing is indexed for an empty list (core/collection.py:138-153): as if absent
            if not self._field_matches(metadata[key], cond):
                return False
        return True


def _value_key(v):
    """What a metadata value is compared by: lists and dicts in a hashable form (core/collection.py:284-290)."""
    if isinstance(v, list):
        v = tuple(v)
    elif isinstance(v, dict):
        v = tuple(sorted(v.items()))
    try:
        hash(v)
    except TypeError:
        v = repr(v)
    return v


class _FieldTable:
    """One metadata field over the rows of the device mirror: the distinct values, and for every row the values it
    holds (a list-valued field holds several, an absent field or an empty list none)."""

    def __init__(self, field_name: str, ids, metadata):
        code_of: Dict[Any, int] = {}
        rows: List[int] = []
        codes: List[int] = []
        for r, id in enumerate(ids):
            if id is None:
                continue
            have = metadata[id].get(field_name, _FieldTable)          # the class itself: "absent"
            if have is _FieldTable:
                continue
            for v in (have if isinstance(have, list) else (have,)):
                k = _value_key(v)
                c = code_of.get(k)
                if c is None:
                    c = code_of[k] = len(code_of)
                rows.append(r)
                codes.append(c)
        self.n_rows = len(ids)
        self.keys = list(code_of)
        self.code_of = code_of
        self.entry_row = np.asarray(rows, dtype=np.int64)
        self.entry_code = np.asarray(codes, dtype=np.int64)
        self.has_field = np.zeros(self.n_rows, dtype=bool)
        self.has_field[self.entry_row] = True

    def rows_with(self, key_ok: np.ndarray) -> np.ndarray:
        """Rows holding at least one value whose entry of ``key_ok`` (one flag per distinct value) is set."""
        out = np.zeros(self.n_rows, dtype=bool)
        if len(self.entry_row):
            out[self.entry_row[key_ok[self.entry_code]]] = True
        return out

    def keys_in(self, wanted) -> np.ndarray:
        ok = np.zeros(len(self.keys), dtype=bool)
        for w in wanted:
            c = self.code_of.get(_value_key(w))
            if c is not None:
                ok[c] = True
        return ok

    def keys_where(self, op: str, arg) -> np.ndarray:
        return np.fromiter((MetadataFilter._cmp(k, op, arg) for k in self.keys), dtype=bool, count=len(self.keys))


def metadata_row_mask(filters: Dict[str, Any], table_of) -> np.ndarray:
    """The rows a metadata filter passes, with the semantics of ``MetadataFilter`` (MetadataIndex.query,
    core/collection.py:183-282), computed from per-field tables: Python work is per DISTINCT value of a field,
    the per-row part is NumPy.  ``table_of(field)`` returns the field's ``_FieldTable``."""
    mask = None
    for name, cond in filters.items():
        t = table_of(name)
        if not isinstance(cond, dict):
            hit = t.rows_with(t.keys_in(cond if isinstance(cond, list) else [cond]))
        else:
            hit = np.zeros(t.n_rows, dtype=bool)
            for op, arg in cond.items():                       # in order: a later operator builds on or replaces the set
                if op == "$in":
                    hit |= t.rows_with(t.keys_in(list(arg)))
                elif op == "$nin":
                    hit = t.has_field & ~t.rows_with(t.keys_in(list(arg)))
                elif op == "$exists":
                    if arg:
                        hit |= t.has_field
                elif op in ("$gt", "$gte", "$lt", "$lte", "$ne"):
                    hit |= t.rows_with(t.keys_where(op, arg))
        mask = hit if mask is None else (mask & hit)
    return mask


@dataclass
class CollectionStats:                                     # core/collection.py:75-100
    name: str
    dimension: int
    metric: str
    vector_count: int
    index_type: str
    memory_usage_bytes: int
    created_at: float
    updated_at: float

    def to_dict(self) -> Dict[str, Any]:
        return {"name": self.name, "dimension": self.dimension, "metric": self.metric, "vector_count": self.vector_count,
                "index_type": self.index_type, "memory_usage_bytes": self.memory_usage_bytes,
                "memory_usage_mb": round(self.memory_usage_bytes / (1024 * 1024), 2),
                "created_at": self.created_at, "updated_at": self.updated_at}


class Collection:
    """core/collection.py:302-828 (search path).

    ``index_type`` (with ``index_params`` or plain keyword arguments such as
    ``n_clusters=...``) routes searches through one of the index classes instead
    of the exact scan.  The reference documents this argument
    (docs/indexing_guide.md:32-36,66-73; examples/semantic_search.py:75-81) but its
    ``Collection.__init__`` (core/collection.py:326-334) does not accept it.  An
    index that needs training is trained on the rows present at the first search
    after it was (re)built; later writes are applied to it one by one."""

    def __init__(self, name: str, dimension: int, metric: str = "euclidean", normalize: bool = False,
                 max_vectors: Optional[int] = None, enable_metadata_index: bool = True,
                 index_type: Optional[str] = None, index_params: Optional[Dict[str, Any]] = None, **index_kwargs):
        self.index_type = index_type.lower() if index_type else None
        # `prefilter` without an index type ("auto" by default, "bf16", or None): the exact scan's batches
        # (search_batch) run behind the bf16 filter pass of the row store, with the same results (not a reference
        # argument)
        self._prefilter = index_kwargs.pop("prefilter", "auto") if self.index_type is None else None
        if self._prefilter not in N.PREFILTER:
            raise ValueError(f"prefilter must be None, 'bf16' or 'auto', got {self._prefilter!r}")
        if self.index_type is None and (index_params or index_kwargs):
            raise TypeError("index parameters given without index_type")
        self._index_params = {**(index_params or {}), **index_kwargs}
        self._index = None
        self.name = name
        self.dimension = dimension
        self.metric = metric
        self.normalize = normalize
        self.max_vectors = max_vectors
        self.enable_metadata_index = enable_metadata_index
        self._vectors: Dict[str, np.ndarray] = {}          # insertion-ordered, like the reference's dict of records
        self._metadata: Dict[str, Dict[str, Any]] = {}
        # Device mirror of the reference's dense matrix (core/collection.py:809-828).  Row order = insertion order
        # of the dict.  A write does not rebuild it: a new id goes behind the last row (uploaded with the other
        # new rows at the next search), an overwritten id keeps its row, a deleted id leaves a dead row that every
        # search masks out.  The rows are packed again once a quarter of them is dead.
        self._matrix_dirty = True                           # full rebuild wanted
        self._ids: List[Optional[str]] = []                 # row -> id (None: dead)
        self._row_of: Dict[str, int] = {}
        self._synced = 0                                    # rows [0, _synced) are on the device
        self._n_dead = 0
        self._alive = np.zeros(0, dtype=bool)
        self._store: Optional[N.RowStore] = None
        self._bulk: Optional[np.ndarray] = None             # rows of a loaded collection, in dict order, not yet uploaded
        # evaluated filters (key: the filter dict as JSON text) stay on the device until the next write; the per-field
        # tables they are computed from are rebuilt lazily after a write
        from ..index.base import FilterCache
        self._filters = FilterCache(4)
        self._field_tables: Dict[str, _FieldTable] = {}
        self._field_tables_version = -1
        self._timestamps: Dict[str, float] = {}
        self._created_at = self._updated_at = time.time()

    def __len__(self) -> int:
        return len(self._vectors)

    def __contains__(self, id: str) -> bool:
        return id in self._vectors

    # ------------------------------------------------------------ writes
    def add(self, id: str, vector, metadata: Optional[Dict[str, Any]] = None, overwrite: bool = False) -> str:
        id = id if id else str(uuid.uuid4())
        vector = _validate_vector(vector, self.dimension)
        if self.normalize:
            vector = _normalize_vector(vector)
        if id in self._vectors and not overwrite:
            raise VectorExistsError(f"Vector with ID '{id}' already exists")
        if id not in self._vectors and self.max_vectors and len(self._vectors) >= self.max_vectors:
            raise ValueError(f"Collection at maximum capacity ({self.max_vectors})")
        existed = id in self._vectors
        self._bulk = None                                   # the loaded matrix no longer describes the rows
        self._filters.touch()
        self._vectors[id] = vector
        self._metadata[id] = dict(metadata or {})
        self._updated_at = self._timestamps[id] = time.time()
        if not self._matrix_dirty:
            if existed:                                     # the dict keeps the position of an existing key
                row = self._row_of[id]
                if row < self._synced:
                    self._store.set_row(row, vector)
            else:
                self._row_of[id] = len(self._ids)
                self._ids.append(id)
        if self._index is not None:
            try:
                if existed:
                    self._index.remove(id)
                self._index.add(id, vector, self._metadata[id])
            except Exception:
                self._drop_index()
        return id

    def add_batch(self, vectors: List[Dict[str, Any]], overwrite: bool = False, on_error: str = "raise") -> Dict[str, Any]:
        res = {"success_count": 0, "error_count": 0, "errors": []}
        for item in vectors:
            try:
                self.add(item.get("id", str(uuid.uuid4())), item["vector"], item.get("metadata", {}), overwrite)
                res["success_count"] += 1
            except Exception as e:
                res["error_count"] += 1
                res["errors"].append({"id": item.get("id"), "error": str(e)})
                if on_error == "raise":
                    raise
        return res

    def get(self, id: str, include_vector: bool = True) -> Optional[Dict[str, Any]]:
        if id not in self._vectors:
            return None
        out = {"id": id, "metadata": self._metadata[id].copy()}
        if include_vector:
            out["vector"] = self._vectors[id].copy()
        return out

    def delete(self, id: str) -> bool:
        if id not in self._vectors:
            return False
        del self._vectors[id], self._metadata[id]
        self._timestamps.pop(id, None)
        self._bulk = None
        self._filters.touch()
        if not self._matrix_dirty:
            row = self._row_of.pop(id)
            self._ids[row] = None
            self._n_dead += 1
            if row < len(self._alive):
                self._alive[row] = False
            if self._n_dead > max(1024, len(self._ids) // 4):
                self._matrix_dirty = True                   # pack the rows again at the next search
        if self._index is not None:
            try:
                self._index.remove(id)
            except Exception:
                self._drop_index()
        return True

    def clear(self) -> int:
        n = len(self._vectors)
        self._vectors.clear(); self._metadata.clear(); self._timestamps.clear()
        self._bulk = None
        self._filters.touch()
        self._matrix_dirty = True
        self._drop_index()
        return n

    # ------------------------------------------------------------ optional index
    def _drop_index(self) -> None:
        self._index = None

    def _ensure_index(self):
        """Build (and, where needed, train) the index over the rows present now."""
        if self._index is not None:
            return self._index
        from ..index import create_index
        ix = create_index(self.index_type, self.dimension, metric=self.metric, **self._index_params)
        ids = list(self._vectors.keys())
        if ids:
            mat = np.stack([self._vectors[i] for i in ids]).astype(np.float32)
            trained = ix.is_trained() if callable(ix.is_trained) else ix.is_trained
            if not trained:
                ix.train(mat)
            ix.add_batch(ids, mat, [self._metadata[i] for i in ids])
        self._index = ix
        return ix

    def _index_results(self, found, include_vector: bool, include_metadata: bool) -> List[SearchResult]:
        out = []
        dmax = np.float32(found[-1].distance) if found else np.float32(1.0)
        for r in found:
            d = np.float32(r.distance)
            score = 1.0 - (d / (dmax + 1e-8)) if dmax > 0 else 1.0             # core/collection.py:762-771
            out.append(SearchResult(id=r.id, distance=float(d), score=float(score),
                                    vector=self._vectors[r.id].copy() if include_vector else None,
                                    metadata=self._metadata[r.id].copy() if include_metadata else {}))
        return out

    def _filter_fn(self, filter: Optional[Dict[str, Any]]):
        if not (filter and self.enable_metadata_index):
            return None
        pred = MetadataFilter(filter)
        return lambda id, metadata: pred(metadata or {})

    # ------------------------------------------------------------ search
    def _ensure_matrix(self) -> None:
        """Bring the device mirror up to date: everything after a rebuild request, otherwise only the rows added
        since the last search."""
        if self._store is None:
            self._store = N.RowStore(self.dimension)
            if self._prefilter:
                self._store.set_prefilter(self._prefilter)
            self._matrix_dirty = True
        if self._matrix_dirty:
            self._store.truncate(0)
            self._ids = list(self._vectors.keys())
            self._row_of = {id: r for r, id in enumerate(self._ids)}
            self._synced = 0
            self._n_dead = 0
            self._alive = np.zeros(0, dtype=bool)
            self._matrix_dirty = False
        n = len(self._ids)
        if self._synced < n:
            # a row that was added and deleted between two searches goes up as zeros and stays masked
            if self._synced == 0 and self._bulk is not None and len(self._bulk) == n and self._n_dead == 0:
                tail = self._bulk                           # a loaded collection nobody has written to: its matrix as it is
            else:
                zero = np.zeros(self.dimension, dtype=np.float32)
                tail = np.stack([self._vectors[i] if i is not None else zero
                                 for i in self._ids[self._synced:]]).astype(np.float32)
            self._bulk = None
            self._store.append(tail)
            alive = np.ones(n, dtype=bool)
            alive[:len(self._alive)] = self._alive
            for r in range(self._synced, n):
                if self._ids[r] is None:
                    alive[r] = False
            self._alive = alive
            self._synced = n

    def _search_device(self, queries: np.ndarray, k: int, filter: Optional[Dict[str, Any]]):
        if self.metric not in _QUERY_METRICS:
            raise ValueError(f"Unknown metric: {self.metric}")
        self._ensure_matrix()
        words = None
        if filter and self.enable_metadata_index:
            _host, words, any_row = self._filters.lookup(self._filter_key(filter), lambda: self._filter_rows(filter))
            if words is None:
                words = _host
            if not any_row:
                return None
        elif self._n_dead:
            if not self._alive.any():
                return None
            words = N.pack_mask(self._alive)
        k = min(k, len(self._vectors))
        return self._store.search(queries, k, _QUERY_METRICS[self.metric], words)

    @staticmethod
    def _filter_key(filter: Dict[str, Any]) -> str:
        import json
        return json.dumps(filter, default=repr)             # key order kept: the operators of a field apply in order

    def _field_table(self, name: str) -> _FieldTable:
        if self._field_tables_version != self._filters.version:
            self._field_tables = {}
            self._field_tables_version = self._filters.version
        t = self._field_tables.get(name)
        if t is None:
            t = self._field_tables[name] = _FieldTable(name, self._ids, self._metadata)
        return t

    def _filter_rows(self, filter: Dict[str, Any]) -> np.ndarray:
        """Row mask of a metadata filter over the device mirror's rows (dead rows hold no values: never set)."""
        return metadata_row_mask(filter, self._field_table)

    def _results(self, dist_row, rows_row, include_vector: bool, include_metadata: bool) -> List[SearchResult]:
        keep = rows_row >= 0
        dist_row, rows_row = dist_row[keep], rows_row[keep]
        out = []
        dmax = dist_row[-1] if len(dist_row) else 1.0
        for d, r in zip(dist_row, rows_row):
            id = self._ids[r]
            score = 1.0 - (d / (dmax + 1e-8)) if dmax > 0 else 1.0           # core/collection.py:762-771 (fp32 scalars)
            out.append(SearchResult(id=id, distance=float(d), score=float(score),
                                    vector=self._vectors[id].copy() if include_vector else None,
                                    metadata=self._metadata[id].copy() if include_metadata else {}))
        return out

    def search(self, query, k: int = 10, filter: Optional[Dict[str, Any]] = None, include_vector: bool = False,
               include_metadata: bool = True) -> List[SearchResult]:
        if len(self._vectors) == 0:
            return []
        query = _validate_vector(query, self.dimension)
        if self.normalize:
            query = _normalize_vector(query)
        if k <= 0:
            return []
        if self.index_type is not None:
            found = self._ensure_index().search(query, k, filter_fn=self._filter_fn(filter))
            return self._index_results(found, include_vector, include_metadata)
        found = self._search_device(query[None, :], k, filter)
        if found is None:
            return []
        return self._results(found[0][0], found[1][0], include_vector, include_metadata)

    def search_batch(self, queries: Union[np.ndarray, List[List[float]]], k: int = 10,
                     filter: Optional[Dict[str, Any]] = None, include_vector: bool = False) -> List[List[SearchResult]]:
        """One device call for the batch (the reference loops over ``search``,
        core/collection.py:803-807)."""
        queries = np.atleast_2d(np.asarray(queries, dtype=np.float32))
        if len(self._vectors) == 0:
            return [[] for _ in range(len(queries))]
        qs = np.stack([_validate_vector(q, self.dimension) for q in queries])
        if self.normalize:
            qs = np.stack([_normalize_vector(q) for q in qs])
        if self.index_type is not None:
            per_query = self._ensure_index().search_batch(qs, k, filter_fn=self._filter_fn(filter))
            return [self._index_results(f, include_vector, True) for f in per_query]
        found = self._search_device(qs, k, filter)
        if found is None:
            return [[] for _ in range(len(qs))]
        return [self._results(found[0][i], found[1][i], include_vector, True) for i in range(len(qs))]

    # ------------------------------------------------------------ statistics and serialisation
    def stats(self) -> CollectionStats:
        """core/collection.py:953-976"""
        memory = 0
        if self._vectors:
            memory += len(self._vectors) * self.dimension * 4
            memory += sum(len(str(m)) for m in self._metadata.values())
            memory += sum(len(i) for i in self._vectors.keys())
        return CollectionStats(name=self.name, dimension=self.dimension, metric=self.metric, vector_count=len(self._vectors),
                               index_type=self.index_type or "flat", memory_usage_bytes=memory,
                               created_at=self._created_at, updated_at=self._updated_at)

    def to_dict(self, arrays: bool = False) -> Dict[str, Any]:
        """core/collection.py:998-1020: name, config, one record per row, timestamps.  With ``arrays=True`` the
        records carry no ``"vector"`` lists; the rows are one float32 matrix under ``"matrix"`` instead (same
        order), which is what a load hands to the device in one piece."""
        ids = list(self._vectors.keys())
        out = {"name": self.name,
               "config": {"dimension": self.dimension, "metric": self.metric, "index_type": self.index_type or "flat",
                          "normalize_vectors": self.normalize, "max_vectors": self.max_vectors,
                          "enable_metadata_index": self.enable_metadata_index, "index_params": dict(self._index_params),
                          "prefilter": self._prefilter},
               "created_at": self._created_at, "updated_at": self._updated_at}
        if arrays:
            out["matrix"] = (np.stack([self._vectors[i] for i in ids]).astype(np.float32) if ids
                             else np.zeros((0, self.dimension), dtype=np.float32))
            out["vectors"] = [{"id": i, "metadata": self._metadata[i], "timestamp": self._timestamps.get(i, self._created_at)}
                              for i in ids]
        else:
            out["vectors"] = [{"id": i, "vector": self._vectors[i].tolist(), "metadata": self._metadata[i],
                               "timestamp": self._timestamps.get(i, self._created_at)} for i in ids]
        return out

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "Collection":
        """core/collection.py:1022-1055.  The rows become one matrix; the first search uploads it whole."""
        config = data["config"]
        index_type = config.get("index_type")
        params = config.get("index_params") or {}
        col = cls(name=data["name"], dimension=config["dimension"], metric=config["metric"],
                  normalize=config.get("normalize_vectors", False), max_vectors=config.get("max_vectors"),
                  enable_metadata_index=config.get("enable_metadata_index", True),
                  index_type=index_type if (index_type and index_type != "flat") else None,
                  index_params=params if (index_type and index_type != "flat") else None)
        if col.index_type is None and "prefilter" in config:
            col._prefilter = config["prefilter"]
        records = data.get("vectors", [])
        if "matrix" in data:
            matrix = np.ascontiguousarray(data["matrix"], dtype=np.float32).reshape(len(records), config["dimension"])
        elif records:
            matrix = np.array([r["vector"] for r in records], dtype=np.float32).reshape(len(records), config["dimension"])
        else:
            matrix = np.zeros((0, config["dimension"]), dtype=np.float32)
        now = time.time()
        for i, r in enumerate(records):
            col._vectors[r["id"]] = matrix[i]
            col._metadata[r["id"]] = dict(r.get("metadata") or {})
            col._timestamps[r["id"]] = r.get("timestamp", now)
        if len(col._vectors) == len(records):               # no duplicate ids: dict order is matrix order
            col._bulk = matrix
        col._matrix_dirty = True
        col._created_at = data.get("created_at", now)
        col._updated_at = data.get("updated_at", now)
        return col

    def __repr__(self) -> str:
        return f"Collection(name='{self.name}', dimension={self.dimension}, count={len(self)}, metric='{self.metric}')"

    def __del__(self):
        store = getattr(self, "_store", None)
        if store is not None:
            store.close()
