"""vectordb's ``distance`` package on the MI355X search core."""

from .metrics import (  # noqa: F401
    query_distances, query_distances_matrix,
    pairwise_euclidean, pairwise_euclidean_squared, pairwise_cosine, pairwise_dot, pairwise_manhattan,
    euclidean, euclidean_squared, cosine_distance, cosine_similarity, dot_product, negative_dot_product,
    manhattan, chebyshev, hamming, angular_distance, normalize_vectors,
)
from .registry import (  # noqa: F401
    DistanceMetric, MetricInfo, MetricRegistry, DistanceCalculator,
    get_metric, get_metric_fn, register_metric, list_metrics, is_similarity, metric_exists, canonical_metric,
)
from .simd import (  # noqa: F401
    get_optimized_distance_fn, simd_euclidean_batch, simd_dot_product_batch, simd_cosine_batch, simd_manhattan_batch,
)
from .batch import (  # noqa: F401
    BatchDistanceCalculator, compute_all_distances, compute_top_k, compute_pairwise_matrix, find_nearest_neighbors,
)
