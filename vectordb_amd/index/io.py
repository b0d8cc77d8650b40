"""Binary save / load of the index classes.

The reference persists indices as ``to_dict()`` output inside a pickle
(core/database.py:345-350; index/flat.py:791-840, ivf.py:897-972, pq.py:392-427,
hybrid.py:456-520): every vector becomes a Python list of floats.  Here the same
state goes to one ``.npz`` file: the large parts (vectors, codes, centroids, list
numbers) as arrays, the small parts (ids, metadata, config) as one JSON string.
Nothing is pickled, so ``numpy.load(..., allow_pickle=False)`` reads it back.
Loading goes through each class's own ``from_dict``, so a loaded index is in
exactly the state ``from_dict(to_dict())`` produces and the device mirror is
rebuilt lazily at the first search.
"""

from __future__ import annotations

import json
from typing import Any, Dict

import numpy as np

from .base import BaseIndex
from .flat import FlatIndex
from .hybrid import HybridIndex
from .ivf import IVFIndex
from .pq import PQIndex

FORMAT_VERSION = 1


def _pack_flat(ix: FlatIndex) -> Dict[str, Any]:
    small = {"index_type": ix.index_type.value, "dimension": ix._dimension, "metric": ix._metric,
             "config": {"normalize": ix.config.normalize, "use_matrix": ix.config.use_matrix,
                        "initial_capacity": ix.config.initial_capacity, "storage": ix._storage, "prefilter": ix._prefilter},
             "ids": list(ix._ids), "metadata": {i: m for i, m in ix._metadata.items()}}
    return {"small": small, "vectors": np.ascontiguousarray(ix._vectors[:ix._size], dtype=np.float32)}


def _unpack_flat(small: Dict[str, Any], z) -> FlatIndex:
    data = dict(small)
    data["vectors"] = z["vectors"]
    return FlatIndex.from_dict(data)


def _pack_ivf(ix: IVFIndex) -> Dict[str, Any]:
    ids, lists, slots = [], [], []
    for c in sorted(ix._lists):
        for s in ix._lists[c]:
            ids.append(ix._slot_id[s]); lists.append(c); slots.append(s)
    vectors = (np.ascontiguousarray(ix._vectors[np.asarray(slots, dtype=np.int64)], dtype=np.float32)
               if slots else np.zeros((0, ix._dimension), np.float32))
    small = {"index_type": ix.index_type.value, "dimension": ix._dimension, "metric": ix._metric,
             "config": {"n_clusters": ix.config.n_clusters, "n_probe": ix.config.n_probe, "n_iter": ix.config.n_iter,
                        "normalize": ix.config.normalize, "seed": ix.config.seed, "prefilter": ix._prefilter},
             "is_trained": bool(ix._is_trained), "ids": ids, "metadata": {i: ix._metadata[i] for i in ids}}
    out = {"small": small, "vectors": vectors, "lists": np.asarray(lists, dtype=np.int32)}
    if ix._centroids is not None:
        out["centroids"] = np.ascontiguousarray(ix._centroids, dtype=np.float32)
    return out


def _unpack_ivf(small: Dict[str, Any], z) -> IVFIndex:
    ids, meta = small["ids"], small["metadata"]
    vectors, lists = z["vectors"], z["lists"]
    inv: Dict[str, list] = {}
    for r, id in enumerate(ids):            # entries stay in the saved order inside every list
        inv.setdefault(str(int(lists[r])), []).append({"id": id, "vector": vectors[r], "metadata": meta.get(id, {})})
    data = {k: small[k] for k in ("index_type", "dimension", "metric", "config", "is_trained")}
    data["centroids"] = z["centroids"] if "centroids" in z.files else None
    data["inverted_lists"] = inv
    return IVFIndex.from_dict(data)


def _pack_pq(ix: PQIndex) -> Dict[str, Any]:
    ids = list(ix._ids)
    rows = np.asarray([ix._row_of[i] for i in ids], dtype=np.int64)
    m = ix.config.n_subvectors
    codes = np.ascontiguousarray(ix._codes_arr[rows], dtype=np.uint8) if len(ids) else np.zeros((0, m), np.uint8)
    small = {"index_type": "pq", "dimension": ix._dimension, "metric": ix._metric,
             "config": {"n_subvectors": ix.config.n_subvectors, "n_bits": ix.config.n_bits},
             "ids": ids, "metadata": ix._metadata}
    out = {"small": small, "codes": codes}
    if ix._codebook is not None:
        out["codebook"] = np.ascontiguousarray(ix._codebook, dtype=np.float32)
    return out


def _unpack_pq(small: Dict[str, Any], z) -> PQIndex:
    data = dict(small)
    data["codebook"] = z["codebook"].tolist() if "codebook" in z.files else None      # small: m * 2^bits * d/m floats
    codes = z["codes"]
    data["codes"] = {id: codes[r] for r, id in enumerate(small["ids"])}
    return PQIndex.from_dict(data)


def _pack_hybrid(ix: HybridIndex) -> Dict[str, Any]:
    rows = [r for r, id in enumerate(ix._row_ids) if id is not None]
    ids = [ix._row_ids[r] for r in rows]
    ra = np.asarray(rows, dtype=np.int64)
    m = ix.config.n_subvectors
    small = {"index_type": "ivf_pq", "dimension": ix._dimension, "metric": ix._metric,
             "config": {"n_clusters": ix.config.n_clusters, "n_probe": ix.config.n_probe,
                        "n_subvectors": ix.config.n_subvectors, "n_bits": ix.config.n_bits},
             "ids": ids, "metadata": {i: ix._metadata.get(i, {}) for i in ids}}
    out = {"small": small,
           "codes": np.ascontiguousarray(ix._codes_arr[ra], dtype=np.uint8) if rows else np.zeros((0, m), np.uint8),
           "lists": np.asarray(ix._list_of_row[ra], dtype=np.int32) if rows else np.zeros((0,), np.int32)}
    if ix._coarse_centroids is not None and ix._pq_codebook is not None:
        out["coarse_centroids"] = np.ascontiguousarray(ix._coarse_centroids, dtype=np.float32)
        out["pq_codebook"] = np.ascontiguousarray(ix._pq_codebook, dtype=np.float32)
    return out


def _unpack_hybrid(small: Dict[str, Any], z) -> HybridIndex:
    ids, meta = small["ids"], small["metadata"]
    codes, lists = z["codes"], z["lists"]
    trained = "coarse_centroids" in z.files
    inv: Dict[str, list] = {str(l): [] for l in range(small["config"]["n_clusters"] if trained else 0)}
    for r, id in enumerate(ids):
        inv.setdefault(str(int(lists[r])), []).append({"id": id, "codes": codes[r], "metadata": meta.get(id, {})})
    data = {k: small[k] for k in ("index_type", "dimension", "metric", "config")}
    data["coarse_centroids"] = z["coarse_centroids"].tolist() if trained else None
    data["pq_codebook"] = z["pq_codebook"].tolist() if trained else None
    data["inverted_lists"] = inv
    return HybridIndex.from_dict(data)


_PACK = {FlatIndex: _pack_flat, IVFIndex: _pack_ivf, PQIndex: _pack_pq, HybridIndex: _pack_hybrid}
_UNPACK = {"flat": _unpack_flat, "ivf": _unpack_ivf, "pq": _unpack_pq, "ivf_pq": _unpack_hybrid}


def save_index(index: BaseIndex, path: str) -> str:
    """Write ``index`` to ``path`` (``.npz`` is appended if missing).  Returns the file name."""
    for cls, pack in _PACK.items():
        if isinstance(index, cls):
            packed = pack(index)
            break
    else:
        raise TypeError(f"cannot save an index of type {type(index).__name__}")
    small = packed.pop("small")
    small["format_version"] = FORMAT_VERSION
    packed["small_json"] = np.frombuffer(json.dumps(small).encode("utf-8"), dtype=np.uint8)
    if not path.endswith(".npz"):
        path += ".npz"
    with open(path, "wb") as f:
        np.savez(f, **packed)
    return path


def load_index(path: str) -> BaseIndex:
    """Read an index written by :func:`save_index`."""
    if not path.endswith(".npz"):
        path += ".npz"
    with np.load(path, allow_pickle=False) as z:
        small = json.loads(bytes(z["small_json"]).decode("utf-8"))
        if small.get("format_version") != FORMAT_VERSION:
            raise ValueError(f"unsupported index file version {small.get('format_version')}")
        kind = small["index_type"]
        if kind not in _UNPACK:
            raise ValueError(f"unknown index type in file: {kind}")
        return _UNPACK[kind](small, z)
