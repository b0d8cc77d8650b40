"""Metric registry with the reference's names, aliases and lookup errors
(distance/registry.py).  ``batch_function`` is the slot an accelerator
registers in (registry.py:58-72); the built-in metrics point it at the HIP
implementations."""

from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Callable, Dict, List, Optional

import numpy as np

from . import metrics as M

DistanceFunction = Callable[[np.ndarray, np.ndarray], float]
BatchDistanceFunction = Callable[..., np.ndarray]


class DistanceMetric(str, Enum):
    """registry.py:42-55"""
    EUCLIDEAN = "euclidean"
    EUCLIDEAN_SQUARED = "euclidean_squared"
    COSINE = "cosine"
    DOT = "dot"
    MANHATTAN = "manhattan"
    CHEBYSHEV = "chebyshev"
    HAMMING = "hamming"
    ANGULAR = "angular"

    def __str__(self) -> str:
        return self.value


@dataclass
class MetricInfo:
    """registry.py:58-72"""
    name: str
    function: DistanceFunction
    batch_function: Optional[BatchDistanceFunction]
    is_similarity: bool
    min_value: Optional[float]
    max_value: Optional[float]
    description: str
    requires_normalization: bool = False

    def __repr__(self) -> str:
        return f"MetricInfo(name='{self.name}', is_similarity={self.is_similarity})"


class MetricRegistry:
    """registry.py:79-276"""

    def __init__(self):
        self._metrics: Dict[str, MetricInfo] = {}
        self._aliases: Dict[str, str] = {}
        b = self.register
        b(MetricInfo("euclidean", M.euclidean, M.pairwise_euclidean, False, 0.0, None, "Euclidean (L2) distance"),
          aliases=["l2", "euclidean_distance"])
        b(MetricInfo("euclidean_squared", M.euclidean_squared, M.pairwise_euclidean_squared, False, 0.0, None,
                     "Squared Euclidean distance (faster, same ordering)"), aliases=["l2_squared", "sqeuclidean"])
        b(MetricInfo("cosine", M.cosine_distance, M.pairwise_cosine, False, 0.0, 2.0,
                     "Cosine distance (1 - cosine similarity)"), aliases=["cosine_distance"])
        b(MetricInfo("dot", M.negative_dot_product, M.pairwise_dot, False, None, None,
                     "Negative dot product (for normalized vectors)", requires_normalization=True),
          aliases=["inner_product", "ip"])
        b(MetricInfo("manhattan", M.manhattan, M.pairwise_manhattan, False, 0.0, None, "Manhattan (L1) distance"),
          aliases=["l1", "cityblock", "taxicab"])
        b(MetricInfo("chebyshev", M.chebyshev, None, False, 0.0, None, "Chebyshev (L-inf) distance"),
          aliases=["linf", "chessboard"])
        b(MetricInfo("hamming", M.hamming, None, False, 0.0, None, "Hamming distance (for binary vectors)"))   # registry.py:179-191
        b(MetricInfo("angular", M.angular_distance, None, False, 0.0, float(np.pi),
                     "Angular distance (angle in radians)"))                                            # registry.py:193-205

    def register(self, info: MetricInfo, aliases: Optional[List[str]] = None) -> None:
        self._metrics[info.name] = info
        for alias in aliases or []:
            self._aliases[alias] = info.name

    def canonical(self, name: str) -> str:
        return self._aliases.get(name, name)

    def get(self, name: str) -> MetricInfo:
        canonical = self._aliases.get(name, name)
        if canonical not in self._metrics:
            raise KeyError(f"Unknown metric: '{name}'. Available: {list(self._metrics.keys())}")   # registry.py:241-245
        return self._metrics[canonical]

    def get_function(self, name: str) -> DistanceFunction:
        return self.get(name).function

    def get_batch_function(self, name: str) -> Optional[BatchDistanceFunction]:
        return self.get(name).batch_function

    def list_metrics(self) -> List[str]:
        return list(self._metrics.keys())

    def list_all(self) -> Dict[str, MetricInfo]:
        return self._metrics.copy()

    def is_similarity(self, name: str) -> bool:
        return self.get(name).is_similarity

    def __contains__(self, name: str) -> bool:
        return self._aliases.get(name, name) in self._metrics

    def __getitem__(self, name: str) -> MetricInfo:
        return self.get(name)


_registry = MetricRegistry()


def get_metric(name: str) -> MetricInfo:
    return _registry.get(name)


def get_metric_fn(name: str) -> DistanceFunction:
    return _registry.get_function(name)


def canonical_metric(name: str) -> str:
    return _registry.canonical(name)


def register_metric(name: str, function: DistanceFunction, is_similarity: bool = False, description: str = "",
                    batch_function: Optional[BatchDistanceFunction] = None,
                    aliases: Optional[List[str]] = None) -> None:
    """registry.py:322-355"""
    _registry.register(MetricInfo(name, function, batch_function, is_similarity, 0.0, None,
                                  description or f"Custom metric: {name}"), aliases)


def list_metrics() -> List[str]:
    return _registry.list_metrics()


def is_similarity(name: str) -> bool:
    return _registry.is_similarity(name)


def metric_exists(name: str) -> bool:
    return name in _registry


class DistanceCalculator:
    """registry.py:398-462"""

    def __init__(self, metric: str = "euclidean"):
        self.metric = metric
        self.info = get_metric(metric)
        self._fn = self.info.function
        self._batch_fn = self.info.batch_function

    def distance(self, a, b) -> float:
        return self._fn(a, b)

    def batch_distances(self, query, collection) -> np.ndarray:
        return M.query_distances(query, collection, self.metric)

    def pairwise(self, X, Y=None) -> np.ndarray:
        if self._batch_fn is not None:
            return self._batch_fn(X, Y)
        Y = X if Y is None else Y
        code = M._QUERY_METRICS.get(canonical_metric(self.metric))
        if code is not None:
            return M._pairwise(X, Y, code)
        out = np.zeros((len(X), len(Y)))
        for i in range(len(X)):
            for j in range(len(Y)):
                out[i, j] = self._fn(X[i], Y[j])
        return out

    @property
    def requires_normalization(self) -> bool:
        return self.info.requires_normalization

    def __repr__(self) -> str:
        return f"DistanceCalculator(metric='{self.metric}')"
