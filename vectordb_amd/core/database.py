"""``VectorDB``: named collections and their on-disk form (core/database.py:30-536 of the reference).

The files are the reference's: ``manifest.json`` plus one ``<name>.collection`` pickle of ``Collection.to_dict()``
per collection (core/database.py:313-358), so a directory written by either side loads on the other.  Loading puts
a collection's rows on the device in one call; nothing is replayed row by row."""

from __future__ import annotations

import json
import pickle
import re
import threading
import time
from pathlib import Path
from typing import Any, Dict, List, Optional

from .collection import Collection


class VectorDBError(Exception):                 # core/exceptions.py:6
    pass


class CollectionNotFoundError(VectorDBError):   # core/exceptions.py:16
    pass


class CollectionExistsError(VectorDBError):     # core/exceptions.py:21
    pass


class ValidationError(VectorDBError):           # core/exceptions.py:46
    pass


class StorageError(VectorDBError):              # core/exceptions.py:61
    pass


_ID_PATTERN = re.compile(r"^[a-zA-Z0-9_\-\.]+$")
_MAX_ID_LENGTH = 256


def validate_id(id: str) -> str:
    """utils/validation.py:27-60"""
    if not isinstance(id, str):
        raise ValidationError(f"ID must be a string, got {type(id).__name__}")
    if not id:
        raise ValidationError("ID cannot be empty")
    if len(id) > _MAX_ID_LENGTH:
        raise ValidationError(f"ID too long: {len(id)} characters (max {_MAX_ID_LENGTH})")
    if not _ID_PATTERN.match(id):
        raise ValidationError(f"Invalid ID '{id}': must contain only alphanumeric characters, underscores, hyphens, or dots")
    return id


class VectorDB:
    VERSION = "0.1.0"
    MANIFEST_FILE = "manifest.json"

    def __init__(self, data_dir: Optional[str] = None, auto_save: bool = False, auto_save_interval: int = 60):
        self._collections: Dict[str, Collection] = {}
        self._data_dir = Path(data_dir) if data_dir else None
        self._lock = threading.RLock()
        self._created_at = self._updated_at = time.time()
        self._auto_save = auto_save
        self._auto_save_interval = auto_save_interval
        self._auto_save_thread: Optional[threading.Thread] = None
        self._running = True
        if self._data_dir:
            self._data_dir.mkdir(parents=True, exist_ok=True)
        if auto_save and self._data_dir:
            self._start_auto_save()

    # ------------------------------------------------------------ collections (core/database.py:105-264)
    def create_collection(self, name: str, dimension: int, metric: str = "euclidean", normalize: bool = False,
                          max_vectors: Optional[int] = None, exist_ok: bool = False, **kwargs) -> Collection:
        name = validate_id(name)
        with self._lock:
            if name in self._collections:
                if exist_ok:
                    return self._collections[name]
                raise CollectionExistsError(f"Collection '{name}' already exists")
            collection = Collection(name=name, dimension=dimension, metric=metric, normalize=normalize,
                                    max_vectors=max_vectors, **kwargs)
            self._collections[name] = collection
            self._updated_at = time.time()
            return collection

    def get_collection(self, name: str) -> Collection:
        with self._lock:
            if name not in self._collections:
                raise CollectionNotFoundError(f"Collection '{name}' not found")
            return self._collections[name]

    def get_or_create_collection(self, name: str, dimension: int, metric: str = "euclidean", **kwargs) -> Collection:
        return self.create_collection(name=name, dimension=dimension, metric=metric, exist_ok=True, **kwargs)

    def delete_collection(self, name: str) -> bool:
        with self._lock:
            if name not in self._collections:
                return False
            del self._collections[name]
            self._updated_at = time.time()
            if self._data_dir:
                path = self._data_dir / f"{name}.collection"
                if path.exists():
                    path.unlink()
            return True

    def list_collections(self) -> List[str]:
        with self._lock:
            return list(self._collections.keys())

    def has_collection(self, name: str) -> bool:
        return name in self._collections

    def collection_stats(self, name: str) -> Dict[str, Any]:
        return self.get_collection(name).stats().to_dict()

    # ------------------------------------------------------------ shortcuts (core/database.py:270-307)
    def add(self, collection_name: str, id: str, vector: Any, metadata: Optional[Dict[str, Any]] = None) -> str:
        return self.get_collection(collection_name).add(id, vector, metadata)

    def search(self, collection_name: str, query: Any, k: int = 10, filter: Optional[Dict[str, Any]] = None, **kwargs):
        return self.get_collection(collection_name).search(query, k=k, filter=filter, **kwargs)

    def delete(self, collection_name: str, id: str) -> bool:
        return self.get_collection(collection_name).delete(id)

    # ------------------------------------------------------------ persistence (core/database.py:313-416)
    def save(self, path: Optional[str] = None, arrays: bool = False) -> None:
        """``arrays=True`` writes every collection's rows as one float32 matrix instead of a Python list of floats
        per row (``Collection.to_dict(arrays=True)``); both forms load here, only the default one loads in the
        reference."""
        save_dir = Path(path) if path else self._data_dir
        if not save_dir:
            raise StorageError("No save path specified and no data_dir configured")
        save_dir.mkdir(parents=True, exist_ok=True)
        try:
            with self._lock:
                manifest = {"version": self.VERSION, "created_at": self._created_at, "updated_at": self._updated_at,
                            "collections": list(self._collections.keys())}
                with open(save_dir / self.MANIFEST_FILE, "w") as f:
                    json.dump(manifest, f, indent=2)
                for name, collection in self._collections.items():
                    with open(save_dir / f"{name}.collection", "wb") as f:
                        pickle.dump(collection.to_dict(arrays=arrays), f)
        except Exception as e:
            raise StorageError(f"Failed to save database: {e}")

    @classmethod
    def load(cls, path: str, **kwargs) -> "VectorDB":
        load_dir = Path(path)
        if not load_dir.exists():
            raise StorageError(f"Database path does not exist: {path}")
        manifest_path = load_dir / cls.MANIFEST_FILE
        if not manifest_path.exists():
            raise StorageError(f"Manifest file not found in {path}")
        try:
            with open(manifest_path, "r") as f:
                manifest = json.load(f)
            db = cls(data_dir=path, **kwargs)
            db._created_at = manifest.get("created_at", time.time())
            db._updated_at = manifest.get("updated_at", time.time())
            for name in manifest.get("collections", []):
                collection_path = load_dir / f"{name}.collection"
                if collection_path.exists():
                    with open(collection_path, "rb") as f:
                        data = pickle.load(f)
                    db._collections[name] = Collection.from_dict(data)
            return db
        except Exception as e:
            raise StorageError(f"Failed to load database: {e}")

    def _start_auto_save(self) -> None:
        def loop():
            while self._running:
                time.sleep(self._auto_save_interval)
                if self._running:
                    try:
                        self.save()
                    except Exception:
                        pass
        self._auto_save_thread = threading.Thread(target=loop, daemon=True)
        self._auto_save_thread.start()

    # ------------------------------------------------------------ info and lifetime (core/database.py:441-536)
    def info(self) -> Dict[str, Any]:
        with self._lock:
            stats = {name: c.stats().to_dict() for name, c in self._collections.items()}
            return {"version": self.VERSION, "data_dir": str(self._data_dir) if self._data_dir else None,
                    "collection_count": len(self._collections),
                    "total_vectors": sum(len(c) for c in self._collections.values()),
                    "total_memory_mb": round(sum(s["memory_usage_bytes"] for s in stats.values()) / (1024 * 1024), 2),
                    "created_at": self._created_at, "updated_at": self._updated_at, "auto_save": self._auto_save,
                    "collections": stats}

    def close(self) -> None:
        self._running = False
        if self._auto_save_thread:
            self._auto_save_thread.join(timeout=5)
        if self._data_dir and self._auto_save:
            try:
                self.save()
            except Exception:
                pass

    def __enter__(self) -> "VectorDB":
        return self

    def __exit__(self, *args) -> None:
        self.close()

    def __getitem__(self, name: str) -> Collection:
        return self.get_collection(name)

    def __contains__(self, name: str) -> bool:
        return self.has_collection(name)

    def __iter__(self):
        return iter(self._collections.keys())

    def __len__(self) -> int:
        return len(self._collections)

    def __repr__(self) -> str:
        return f"VectorDB(collections={len(self._collections)}, data_dir='{self._data_dir}')"
