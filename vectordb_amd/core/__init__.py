"""The search entry point of vectordb's ``core`` package on the device core."""

from .collection import Collection, SearchResult, MetadataFilter, VectorExistsError   # noqa: F401
