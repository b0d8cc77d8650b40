"""The search entry point of vectordb's ``core`` package on the device core."""

from .collection import Collection, CollectionStats, SearchResult, MetadataFilter, VectorExistsError   # noqa: F401
from .database import VectorDB, StorageError                                                            # noqa: F401
