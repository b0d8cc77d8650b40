"""vectordb's ``distance`` package on the MI355X search core."""

from .metrics import (  # noqa: F401
    query_distances, query_distances_matrix,
    pairwise_euclidean, pairwise_euclidean_squared, pairwise_cosine, pairwise_dot, pairwise_manhattan,
    euclidean, euclidean_squared, cosine_distance, cosine_similarity, dot_product, negative_dot_product,
    manhattan, chebyshev, normalize_vectors,
)
from .registry import (  # noqa: F401
    DistanceMetric, MetricInfo, MetricRegistry, DistanceCalculator,
    get_metric, get_metric_fn, register_metric, list_metrics, is_similarity, metric_exists, canonical_metric,
)
from .batch import (  # noqa: F401
    BatchDistanceCalculator, compute_all_distances, compute_top_k, compute_pairwise_matrix, find_nearest_neighbors,
)
