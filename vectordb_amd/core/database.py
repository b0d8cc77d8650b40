"""On-disk form of a set of collections: ``VectorDB.save(path)`` / ``VectorDB.load(path)``.

The reference writes ``manifest.json`` and one pickle of ``Collection.to_dict()`` per collection, every row as a
Python list of floats (core/database.py:313-358).  Here a collection goes to ``<name>.npz``: its rows as one float32
matrix (what the first search hands to the device in a single upload), its timestamps as one float64 vector, and
ids, metadata and configuration as one JSON string.  Nothing is pickled, so ``numpy.load(allow_pickle=False)`` reads
it back.  ``load`` also reads a directory the reference wrote (``<name>.collection``, core/database.py:360-416):
those pickles hold dicts, lists, strings and numbers only, so they are read by an unpickler that resolves no names.

Only the file layout lives here.  Collection lifecycle beyond holding the named collections (auto-save, statistics,
per-collection shortcuts) stays with the reference.
"""

from __future__ import annotations

import io
import json
import os
import pickle
import time
from typing import Dict, Iterator, List, Optional

import numpy as np

from .collection import Collection

MANIFEST = "manifest.json"
LAYOUT = "npz-1"


class StorageError(Exception):
    """A directory that cannot be written or read (core/exceptions.py:61 has the reference's type of that name)."""


def _file_stem(name: str) -> str:
    """A collection name is used as a file name: it must stay inside the directory it is joined to."""
    if not isinstance(name, str) or not name or name in (".", "..") or os.path.basename(name) != name \
            or any(ch in name for ch in "/\\\0") or len(name) > 255:
        raise ValueError(f"collection name {name!r} cannot be used as a file name")
    return name


class _PlainUnpickler(pickle.Unpickler):
    """Reads pickles made of built-in containers and scalars; any reference to a class or function is refused."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"collection file refers to {module}.{name}; only plain data is read")


def _write_npz(path: str, col: Collection) -> None:
    d = col.to_dict(arrays=True)
    records = d.pop("vectors")
    matrix = d.pop("matrix")
    small = dict(d, ids=[r["id"] for r in records], metadata=[r["metadata"] for r in records])
    stamps = np.asarray([r["timestamp"] for r in records], dtype=np.float64)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        np.savez(f, matrix=matrix, timestamps=stamps,
                 small_json=np.frombuffer(json.dumps(small).encode("utf-8"), dtype=np.uint8))
    os.replace(tmp, path)                       # a crash in the middle leaves the previous file whole


def _read_npz(path: str) -> Collection:
    with np.load(path, allow_pickle=False) as z:
        small = json.loads(bytes(z["small_json"]).decode("utf-8"))
        matrix, stamps = z["matrix"], z["timestamps"]
    ids, meta = small.pop("ids"), small.pop("metadata")
    if not (len(ids) == len(meta) == len(stamps) == len(matrix)):
        raise ValueError(f"{path}: {len(ids)} ids, {len(meta)} metadata entries, {len(stamps)} timestamps, {len(matrix)} rows")
    small["vectors"] = [{"id": i, "metadata": m, "timestamp": float(t)} for i, m, t in zip(ids, meta, stamps)]
    small["matrix"] = matrix
    return Collection.from_dict(small)


def _read_reference_pickle(path: str) -> Collection:
    with open(path, "rb") as f:
        data = _PlainUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(data, dict) or "config" not in data:
        raise ValueError(f"{path} does not hold a collection")
    return Collection.from_dict(data)


class VectorDB:
    """Named collections with a directory form.  ``db[name]`` is the collection."""

    def __init__(self, data_dir: Optional[str] = None):
        self.data_dir = data_dir
        self._cols: Dict[str, Collection] = {}
        self._born = time.time()

    # -- the collections it holds
    def create_collection(self, name: str, dimension: int, metric: str = "euclidean", **kwargs) -> Collection:
        """Keyword arguments go to ``Collection`` (``index_type=...`` and index parameters included, as
        core/database.py:105-148 forwards them)."""
        _file_stem(name)
        if name in self._cols:
            raise ValueError(f"Collection '{name}' already exists")
        col = self._cols[name] = Collection(name=name, dimension=dimension, metric=metric, **kwargs)
        return col

    def get_collection(self, name: str) -> Collection:
        return self[name]

    def list_collections(self) -> List[str]:
        return list(self._cols)

    def __getitem__(self, name: str) -> Collection:
        try:
            return self._cols[name]
        except KeyError:
            raise KeyError(f"Collection '{name}' not found") from None

    def __contains__(self, name: str) -> bool:
        return name in self._cols

    def __iter__(self) -> Iterator[str]:
        return iter(self._cols)

    def __len__(self) -> int:
        return len(self._cols)

    # -- files
    def save(self, path: Optional[str] = None) -> None:
        """Write the manifest and one ``.npz`` per collection into ``path`` (default: ``data_dir``)."""
        target = path or self.data_dir
        if not target:
            raise StorageError("No save path specified and no data_dir configured")
        try:
            os.makedirs(target, exist_ok=True)
            for name, col in self._cols.items():
                _write_npz(os.path.join(target, _file_stem(name) + ".npz"), col)
            with open(os.path.join(target, MANIFEST), "w") as f:
                json.dump({"layout": LAYOUT, "created_at": self._born, "updated_at": time.time(),
                           "collections": list(self._cols)}, f, indent=2)
        except Exception as e:
            raise StorageError(f"Failed to save database: {e}") from e

    @classmethod
    def load(cls, path: str) -> "VectorDB":
        """Read a directory written by ``save`` or by the reference's ``VectorDB.save``; a manifest entry whose
        file is missing is skipped, as core/database.py:396-404 does."""
        manifest_path = os.path.join(path, MANIFEST)
        if not os.path.isdir(path):
            raise StorageError(f"Database path does not exist: {path}")
        if not os.path.isfile(manifest_path):
            raise StorageError(f"Manifest file not found in {path}")
        try:
            with open(manifest_path) as f:
                manifest = json.load(f)
            db = cls(data_dir=path)
            db._born = manifest.get("created_at", db._born)
            for name in manifest.get("collections", []):
                stem = os.path.join(path, _file_stem(name))
                if os.path.isfile(stem + ".npz"):
                    db._cols[name] = _read_npz(stem + ".npz")
                elif os.path.isfile(stem + ".collection"):
                    db._cols[name] = _read_reference_pickle(stem + ".collection")
            return db
        except Exception as e:
            raise StorageError(f"Failed to load database: {e}") from e

    def __repr__(self) -> str:
        return f"VectorDB({len(self._cols)} collections, data_dir={self.data_dir!r})"
