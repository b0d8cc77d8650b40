/*
 * vdb_core.h -- C ABI of the MI355X (gfx950) search core for vectordb.
 *
 * One shared library (libvdbcore.so), plain pointers and sizes, no C++ or
 * framework types.  Each entry point names the reference interface it
 * replaces (paths relative to the reference checkout).  The reference is pure
 * Python/NumPy, so "replaces" means: the NumPy expression or Python loop at
 * that file:line is what the kernel behind the entry point computes.
 *
 * Conventions
 *   - every function returns VDB_OK (0) or a negative vdb_status; the text of
 *     the last failure on the calling thread is vdb_last_error();
 *   - vectors are fp32, row-major, contiguous (index/base.py:435-469 coerces
 *     every input to exactly that);
 *   - row numbers are int64 on the host side, distances fp32;
 *   - a filter is a little-endian packed bitmask: bit (i % 32) of word (i / 32)
 *     set means row i passes (the host evaluates filter_fn / the metadata
 *     filter once per row, index/base.py:125, core/collection.py:724);
 *   - result slots that cannot be filled (fewer than k rows pass) hold
 *     row = -1 and distance = +inf;
 *   - ordering of results: ascending (distance, row); equal distances go to
 *     the lower row number (= earlier insertion), the rule of every stable
 *     sort in the reference (core/collection.py:757, index/pq.py:300); the
 *     IVF search orders equal distances by scan position instead
 *     (index/ivf.py:619-631);
 *   - functions whose name ends in _dev take DEVICE pointers for queries,
 *     masks and outputs plus the hipStream_t to enqueue on, passed as void*
 *     (NULL = the HIP null stream); they enqueue work and do not synchronise.
 *     The plain forms take HOST pointers, copy, run on a stream of the library
 *     (a blocking stream, i.e. ordered with the null stream) and return when
 *     the results are in the caller's buffers.  Calls on one handle share its
 *     workspaces: order them yourself when they go to different streams.
 */
#ifndef VDB_CORE_H
#define VDB_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vdb_status {
    VDB_OK = 0,
    VDB_ERR_INVALID = -1,   /* bad argument: shape, metric, k            */
    VDB_ERR_HIP = -2,       /* a HIP runtime call failed                 */
    VDB_ERR_STATE = -3,     /* not initialised / not trained / empty     */
    VDB_ERR_NOMEM = -4      /* device or host allocation failed          */
} vdb_status;

/* distance/metrics.py:453-490 -- the six metrics query_distances knows */
typedef enum vdb_metric {
    VDB_EUCLIDEAN = 0,      /* sqrt(max(|q|^2 + |c|^2 - 2 c.q, 0))       :453-459 */
    VDB_EUCLIDEAN_SQ = 1,   /* max(|q|^2 + |c|^2 - 2 c.q, 0)             :461-466 */
    VDB_COSINE = 2,         /* 1 - clip(c.q / (|q| |c|), -1, 1)          :468-481 */
    VDB_DOT = 3,            /* -(c.q)                                    :483-484 */
    VDB_MANHATTAN = 4,      /* sum |c - q|                               :486-487 */
    VDB_CHEBYSHEV = 5,      /* max |c - q|                               :489-490 */
    /* direct forms used by the per-pair scalar functions (IVF search,
     * index/ivf.py:611-613,625) and by GPUDistanceCalculator
     * (distance/gpu.py:118-120): */
    VDB_EUCLIDEAN_DIRECT = 6,   /* sqrt(sum (q - c)^2), metrics.py:43    */
    VDB_COSINE_PAIR = 7,        /* 1 - c.q/(|q||c|), 0 if |q||c| < 1e-8, metrics.py:92-100,123 */
    VDB_EUCLIDEAN_SQ_DIRECT = 8,/* sum (q - c)^2, metrics.py:68-69       */
    /* the two remaining per-pair functions of the registry (distance/registry.py:179-205); reachable
     * through get_metric_fn, e.g. IVFIndex(metric="angular") (index/ivf.py:611-613,625): */
    VDB_HAMMING = 9,            /* number of positions where q != c, metrics.py:237 */
    VDB_ANGULAR = 10            /* arccos(clip(c.q/(|q||c|), -1, 1)), similarity 0 if |q||c| < 1e-8, metrics.py:535-538 */
} vdb_metric;

/* ---- library ---------------------------------------------------------- */

/* Select the GPU this process uses (one process per GPU).  Replaces the
 * device selection of distance/gpu.py:63-67.  The first call fixes the device:
 * a repeated call with the same device is a no-op, one with another device
 * returns VDB_ERR_STATE (handles, streams and buffers live on the first).  Every
 * entry point makes that device current on the calling thread, so handles may
 * be used from any thread; calls on ONE handle are serialised by the library,
 * calls on different handles run concurrently (each handle has its own stream). */
int vdb_init(int device);
const char* vdb_last_error(void);
/* distance/gpu.py:205-225 get_gpu_info(): name (<=255 chars), total and free bytes */
int vdb_device_info(char* name, int name_len, int64_t* mem_total, int64_t* mem_free);
/* block until everything enqueued on the device is done (every handle has a stream of its own) */
int vdb_synchronize(void);
/* What this GPU reaches on the two limits the roofline fractions are quoted against (no counterpart in the
 * reference): a 1 GiB device-to-device copy in GB/s (bytes read + bytes written) and register-only
 * loops of v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16 in TFLOP/s, and of the subtract / multiply / add
 * mix of the direct-form scans on the vector ALUs in 10^12 operations per second.  Allocates 2 GiB while it runs. */
int vdb_measure_ceilings(float* copy_gbs, float* mfma_f32_tflops, float* mfma_bf16_tflops, float* valu_direct_tops);
/* Kernel timing for roofline accounting (no counterpart in the reference; its
 * only timers are wall clocks around Collection.search, query/executor.py:255-272).
 * While enabled, every launch of a path's dominant kernel (the MFMA flat
 * kernel, the streaming scan, the PQ scan) is bracketed by HIP events on the
 * stream it is launched on.  vdb_profile_read waits for those events, returns
 * the summed kernel time in milliseconds and the number of launches since the
 * last read, and clears the record. */
int vdb_profile_enable(int on);
int vdb_profile_read(float* kernel_ms, int* launches);

/* ---- resident filter masks ----------------------------------------------
 * The reference evaluates filter_fn(id, metadata) for every row inside every filtered search
 * (index/base.py:125; index/flat.py:396-413, index/ivf.py:622, index/pq.py:287-297) and rebuilds the id
 * set of a metadata filter per call (core/collection.py:724).  Here a filter is evaluated once per write
 * version of an index, packed (bit i = row i passes) and kept in HBM as a vdb_mask; the *_search_masked
 * entry points name it instead of uploading the words again.  `words` holds (n_bits + 31) / 32 little-endian
 * words; vdb_mask_create returns when the copy is on the device.  A mask may be shorter-lived than the
 * index but must cover at least the index's rows when a search names it (VDB_ERR_INVALID otherwise).
 * vdb_mask_info returns the bit count and the device address of the words, for the *_dev entry points. */
typedef struct vdb_mask vdb_mask;
int vdb_mask_create(vdb_mask** out, const uint32_t* words, int64_t n_bits);
int vdb_mask_destroy(vdb_mask* m);
int vdb_mask_info(const vdb_mask* m, int64_t* n_bits, const uint32_t** d_words);

/* ---- resident fp32 row matrix (FlatIndex storage, Collection matrix) ---
 * index/flat.py:136-141,261,276-282 (growable fp32 matrix) and
 * core/collection.py:809-828 (_ensure_matrix).  Appending also computes and
 * keeps |c|^2 in NumPy's pairwise order (metrics.py:456) and |c| (:470), which
 * the reference recomputes on every search. */
typedef struct vdb_rows vdb_rows;
int vdb_rows_create(vdb_rows** out, int dim);
/* Same with a storage type.  VDB_STORE_BF16 keeps the rows (and rounds the queries) in
 * bfloat16, round-to-nearest-even, multiplies them on the bf16 matrix cores with fp32
 * accumulation, and takes the norm terms of the rounded values; it serves the
 * euclidean, euclidean_squared, cosine and dot metrics.  No counterpart in the
 * reference, which is fp32 throughout: results are those of the fp32 path on rounded
 * inputs (tolerance and recall: tests/test_gpu_cabi.py::test_flat_bf16_storage). */
enum vdb_storage { VDB_STORE_F32 = 0, VDB_STORE_BF16 = 1 };
int vdb_rows_create_typed(vdb_rows** out, int dim, int storage);
int vdb_rows_destroy(vdb_rows* h);
int vdb_rows_append(vdb_rows* h, const float* rows, int64_t n);           /* flat.py:261 */
int vdb_rows_append_dev(vdb_rows* h, const float* d_rows, int64_t n, int64_t ld, void* stream);
int vdb_rows_set(vdb_rows* h, int64_t row, const float* vec);             /* flat.py:599, :201 */
int vdb_rows_truncate(vdb_rows* h, int64_t n);                            /* flat.py:608-609 */
int vdb_rows_size(const vdb_rows* h, int64_t* n);
int vdb_rows_norms(vdb_rows* h, float* sqnorm_out, float* norm_out);      /* test hook: copies of the stored terms */
/* pairwise_cosine(X, Y, eps) (distance/metrics.py:336-373): norms below eps count as 1.  Applies to the
 * cosine distance MATRIX of this handle (vdb_flat_distances*); searches keep the fixed 1e-8 of
 * query_distances (metrics.py:469-477), which has no such argument.  Default 1e-8. */
int vdb_rows_set_cosine_eps(vdb_rows* h, float eps);
/* Batched searches of an fp32 handle (euclidean, euclidean_squared, cosine, dot; the contraction of
 * distance/metrics.py:453-484) behind a bf16 filter pass.  No counterpart in the reference, and no change of
 * results: with VDB_PREFILTER_BF16 the handle keeps a bf16 copy of its rows beside them (half their size again,
 * converted on the first search and after every write), the pass over all rows runs on the bf16 matrix cores with
 * its bound loosened by the largest error rounding to bf16 can cause, and the fp32 dot product of every row that
 * can still be among the k best is recomputed with the instruction sequence of the fp32 kernel.  Distances, rows
 * and their order are bit for bit those of VDB_PREFILTER_NONE (tests/test_gpu_prefilter.py).  Applies to batches
 * of at least 4 queries over at least 65536 rows with k <= 32, any dimension (the copy's rows are padded to whole
 * 128-byte steps); other calls, vdb_flat_search_static_dev and values whose products could overflow take the fp32
 * kernels.  VDB_PREFILTER_AUTO is the same with a condition: the copy is
 * taken (and grown) only while that leaves a tenth of the device memory, at least 2 GiB, free; a search that finds no
 * room runs on the fp32 kernels alone, with the same results.  The environment variable VDB_PREFILTER=1 makes
 * VDB_PREFILTER_BF16 the default of new fp32 handles. */
enum vdb_prefilter { VDB_PREFILTER_NONE = 0, VDB_PREFILTER_BF16 = 1, VDB_PREFILTER_AUTO = 2 };
int vdb_rows_set_prefilter(vdb_rows* h, int mode);

/* FlatIndex.search / search_batch (index/flat.py:288-460), the search part of
 * Collection.search (core/collection.py:733-758) and
 * BatchDistanceCalculator.top_k (distance/batch.py:136-220):
 * nq queries -> k best rows each.  out_dist / out_row are nq*k. */
int vdb_flat_search(vdb_rows* h, int metric, const float* queries, int nq, int k,
                    const uint32_t* mask, float* out_dist, int64_t* out_row);
int vdb_flat_search_dev(vdb_rows* h, int metric, const float* d_queries, int nq, int k,
                        const uint32_t* d_mask, float* d_out_dist, int64_t* d_out_row, void* stream);
/* vdb_flat_search with a resident mask (index/flat.py:386-431 _search_with_filter) */
int vdb_flat_search_masked(vdb_rows* h, int metric, const float* queries, int nq, int k,
                           const vdb_mask* mask, float* out_dist, int64_t* out_row);

/* vdb_flat_search_dev for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same
 * kernels, same results, but every workspace of the call (padded queries and their norms, the bound
 * pass's minima, the candidate queues and counters, candidate lists, plan tables) lives in ONE static
 * device array of VDB_STATIC_SCRATCH_BYTES that the library holds on its GPU from load to exit, and the
 * call makes no hipMalloc / hipFree / hipHostMalloc (kernel timing, vdb_profile_enable, must be off: it
 * creates events).  A call whose workspaces do not fit returns VDB_ERR_INVALID and names the size; 64 MiB
 * holds 1024 queries x 1M rows at k = 10.  Every call on a GPU shares the array and finds the last call's
 * bytes in it (what must start at zero is cleared by the call itself): two calls that overlap on the
 * device - different streams, or different handles from different threads - race; order them. */
#define VDB_STATIC_SCRATCH_BYTES (64u << 20)
int vdb_flat_search_static_dev(vdb_rows* h, int metric, const float* d_queries, int nq, int k,
                               const uint32_t* d_mask, float* d_out_dist, int64_t* d_out_row, void* stream);

/* Diagnostics of the last vdb_flat_search* call on this handle (no counterpart in the reference;
 * the analogue of IVFIndex's distance_computations counter, index/ivf.py:626,838):
 * *used_queue = 1 when the batch ran the candidate-queue kernel behind a sampled bound (2 when that was the
 * bf16 filter pass of vdb_rows_set_prefilter),
 * *overflowed = 1 when a queue was too small and the list kernel repeated the search.
 * Waits for the handle's stream. */
int vdb_flat_search_stats(vdb_rows* h, int* used_queue, int* overflowed);

/* query_distances (distance/metrics.py:431-493), pairwise_* (:272-424),
 * GPUDistanceCalculator.compute (distance/gpu.py:74-134): the full nq x n
 * distance matrix, out[i*n + j] = distance(query i, row j). */
int vdb_flat_distances(vdb_rows* h, int metric, const float* queries, int nq, float* out);
int vdb_flat_distances_dev(vdb_rows* h, int metric, const float* d_queries, int nq, float* d_out, void* stream);

/* FlatIndex.range_search (index/flat.py:501-572): rows with distance <= radius for one
 * query.  Like the reference, the FIRST `cap` such rows in row (= insertion) order are kept
 * (:545-567, max_results) and those are returned ascending by distance, equal distances in row
 * order (the stable sort of :570).  *count receives the number written (<= cap).  One pass over
 * the rows; compaction and the sort run on the device. */
int vdb_flat_range_search(vdb_rows* h, int metric, const float* query, float radius,
                          const uint32_t* mask, int64_t cap, float* out_dist, int64_t* out_row, int64_t* count);

/* ---- nearest-centroid assignment ---------------------------------------
 * KMeans._assign (index/ivf.py:228-231), IVFIndex._find_nearest_centroids_batch
 * (:553-565) and _find_nearest_centroid (:546-551): argmin over centroids,
 * taken on the FINAL distance value (after clamp and sqrt), first minimum.
 * metric selects the arithmetic form (VDB_EUCLIDEAN = identity form of
 * pairwise_euclidean; VDB_EUCLIDEAN_DIRECT = scalar form; ...). */
int vdb_assign(const float* x, int64_t n, const float* centroids, int n_centroids, int dim,
               int metric, int32_t* out_idx, float* out_dist);

/* One Lloyd update (index/ivf.py:177-187, index/pq.py:160-169): per-cluster
 * fp64 sums of the rows assigned to it and the member counts. */
int vdb_kmeans_accumulate(const float* x, int64_t n, int dim, const int32_t* assign, int n_clusters,
                          double* out_sums, int64_t* out_counts);

/* ---- k-means on resident rows ---------------------------------------------
 * KMeans (index/ivf.py:102-237) and the sub-quantizer training of PQIndex (index/pq.py:144-180) and
 * HybridIndex (index/hybrid.py:174-203) without a host round trip per step: the training rows are a
 * vdb_rows handle (uploaded once), the centroids, assignments and sums stay on the device, and only what
 * the host's RandomState and stopping rule need comes back (k counts and one number per step).
 * A trainer works on columns [col0, col0 + dsub) of the rows (a PQ sub-space) and borrows `train`, which
 * must outlive it and must not be written meanwhile.
 *
 * Arithmetic of a step, per form:
 *   VDB_KMEANS_IVF  assignment = argmin of pairwise_euclidean (identity form, clamp, sqrt; ivf.py:228-231);
 *                   sums fp32, one row at a time in sample order (:180-182); new = float(sum / float64(count))
 *                   (:187); a cluster without members is left for vdb_kmeans_reseed (:188-190).
 *   VDB_KMEANS_PQ   first step: argmin of |x|^2 + |c|^2 - 2 x.c in fp32 (pq.py:182-186), fp32 sums, new =
 *                   float64(sum) / max(count, 1) kept as FLOAT64 (:168-169); later steps: the same distances
 *                   and sums in float64, as NumPy promotes them there.  Empty clusters become zero vectors.
 * The sums are taken in the reference's order (a stable sort of the rows by cluster, then one sequential
 * pass per cluster and column), so equal assignments give bit-identical centroids. */
typedef struct vdb_kmeans vdb_kmeans;
enum vdb_kmeans_form { VDB_KMEANS_IVF = 0, VDB_KMEANS_PQ = 1 };
int vdb_kmeans_create(vdb_kmeans** out, vdb_rows* train, int col0, int dsub, int n_clusters);
int vdb_kmeans_destroy(vdb_kmeans* h);
/* centroids = these training rows (pq.py:150-151: rows drawn by rng.choice(n, k, replace=False)) */
int vdb_kmeans_init_rows(vdb_kmeans* h, const int64_t* rows, int n_rows);
/* k-means++ (ivf.py:204-226).  first_row = rng.randint(n); uniforms[j] = the (j+1)-th rng.random_sample(),
 * the one draw rng.choice(n, p=probs) makes for centroid j+1 (n_clusters - 1 values).  The running minimum of
 * the squared distances and every draw (first row whose cumulative share of the total exceeds the uniform
 * number) stay on the device; out_rows (n_clusters values, may be NULL) receives the chosen rows. */
int vdb_kmeans_init_pp(vdb_kmeans* h, int64_t first_row, const double* uniforms, int64_t* out_rows);
int vdb_kmeans_set_centroids(vdb_kmeans* h, const float* centroids);       /* start from given fp32 centroids */
/* One Lloyd step into a pending set of centroids: out_counts (n_clusters values) and
 * out_shift = sum((new - old)^2) (ivf.py:193) come back; nothing else leaves the device. */
int vdb_kmeans_step(vdb_kmeans* h, int form, int64_t* out_counts, double* out_shift);
/* ivf.py:188-190: pending centroid clusters[i] = training row rows[i] (rows drawn by rng.randint(n), in
 * ascending cluster order); out_shift is recomputed. */
int vdb_kmeans_reseed(vdb_kmeans* h, const int32_t* clusters, const int64_t* rows, int count, double* out_shift);
int vdb_kmeans_commit(vdb_kmeans* h);                                      /* pending centroids become current */
/* current centroids: fp32 (always; the float64 ones rounded, as `codebook[m] = centroids` rounds them,
 * pq.py:140) and, after a PQ-form step, the float64 values themselves (out64 may be NULL) */
int vdb_kmeans_centroids(vdb_kmeans* h, float* out32, double* out64);
/* assignments of the last step (or, with no step pending, KMeans.predict of the training rows, ivf.py:233-237)
 * and, for the fp32 forms, the distance to the assigned centroid */
int vdb_kmeans_assignments(vdb_kmeans* h, int32_t* out_assign, float* out_dist);

/* ---- IVF ------------------------------------------------------------------
 * Inverted lists stored contiguously in list order (CSR offsets), the
 * device-side form of index/ivf.py:320 (dict of Python lists of VectorEntry). */
typedef struct vdb_ivf vdb_ivf;
int vdb_ivf_create(vdb_ivf** out, int dim, int metric);
int vdb_ivf_destroy(vdb_ivf* h);
int vdb_ivf_set_centroids(vdb_ivf* h, const float* centroids, int n_lists);   /* ivf.py:417 */
/* rows in insertion order + the list each belongs to (ivf.py:526-542); replaces the content */
int vdb_ivf_build(vdb_ivf* h, const float* rows, int64_t n, const int32_t* list_of_row);
/* explicit list layout: row numbers of every list in scan order, concatenated,
 * and n_lists+1 offsets (covers swap-removed lists, ivf.py:730-738) */
int vdb_ivf_build_lists(vdb_ivf* h, const float* rows, int64_t n, const int64_t* list_rows, const int64_t* list_offsets);
/* the same layout with rows already in list order and a row number of the caller's choice for every one of
 * them (what searches return and what a mask is indexed by; any values below 2^31, e.g. slots of a host table
 * that has gaps where rows were removed) */
int vdb_ivf_build_numbered(vdb_ivf* h, const float* rows, int64_t n, const int64_t* row_numbers, const int64_t* list_offsets);
/* IVFIndex.add / add_batch on a built index (ivf.py:478-542: `self._inverted_lists[cluster].append(entry)`):
 * row i goes behind the last row of list lists[i].  Costs the new rows, plus, when a list has outgrown its
 * place, one device-side move of that list to a place twice as large; the rest of the index is not touched. */
int vdb_ivf_append(vdb_ivf* h, const float* rows, const int64_t* row_numbers, const int32_t* lists, int64_t count);
/* IVFIndex.remove (ivf.py:708-745): the row at `position` of `list` is replaced by the list's last row */
int vdb_ivf_remove_at(vdb_ivf* h, int32_t list, int64_t position);
/* rows in lists, positions of the row buffer in use, and positions lost to places that lists have left (a caller
 * rebuilds when the last is large) */
int vdb_ivf_layout(vdb_ivf* h, int64_t* live_rows, int64_t* positions, int64_t* dead_positions);
/* IVFIndex.search / search_batch (ivf.py:571-673).  mask is indexed by row
 * number (insertion position). */
int vdb_ivf_search(vdb_ivf* h, const float* queries, int nq, int k, int n_probe,
                   const uint32_t* mask, float* out_dist, int64_t* out_row);
int vdb_ivf_search_dev(vdb_ivf* h, const float* d_queries, int nq, int k, int n_probe,
                       const uint32_t* d_mask, float* d_out_dist, int64_t* d_out_row, void* stream);
/* vdb_ivf_search with a resident mask (index/ivf.py:619-628, the filter_fn test inside the list scan) */
int vdb_ivf_search_masked(vdb_ivf* h, const float* queries, int nq, int k, int n_probe,
                          const vdb_mask* mask, float* out_dist, int64_t* out_row);
/* The same search, plus for every result its place in the reference's visiting
 * order (ivf.py:619-628): order = probe rank << 32 | position in the list, -1 for
 * an unfilled slot.  Equal distances are ordered by it; a shard that holds only
 * some of the lists reports keys that compare correctly with other shards'. */
int vdb_ivf_search_ordered_dev(vdb_ivf* h, const float* d_queries, int nq, int k, int n_probe,
                               const uint32_t* d_mask, float* d_out_dist, int64_t* d_out_row,
                               int64_t* d_out_order, void* stream);
/* probe selection alone (ivf.py:611-614): n_probe nearest lists per query, ascending */
int vdb_ivf_probe(vdb_ivf* h, const float* queries, int nq, int n_probe, int32_t* out_lists, float* out_dist);
/* Batched Euclidean searches (index/ivf.py:619-632 with the per-pair form of distance/metrics.py:43) behind a
 * matrix-core scan: the handle keeps a bf16 copy of its member rows (half their size again, brought up to date
 * lazily after writes), the scan computes |q|^2 + |c|^2 - 2 q.c on the bf16 matrix cores for every (probed row,
 * query) pair, and only pairs within a proven margin of the query's k-th best are recomputed in the reference's
 * arithmetic (float32 differences, NumPy's summation order) and ranked.  Rows, their order and the distance bits
 * are those of the streaming scan (tests/test_gpu_ivf_mfma.py); no counterpart in the reference.  Modes as
 * vdb_rows_set_prefilter: VDB_PREFILTER_AUTO (the default: the copy is taken while it leaves a tenth of the
 * device free), VDB_PREFILTER_BF16 (always), VDB_PREFILTER_NONE (never; frees the copy).  Applies to batches of
 * at least 2 queries with k <= 32. */
int vdb_ivf_set_prefilter(vdb_ivf* h, int mode);
/* Diagnostics of the last search on this handle: *used_matrix_scan = 1 when it took the path above, *overflowed = 1
 * when a candidate queue was too small and the streaming scan repeated the batch.  Waits for the search. */
int vdb_ivf_search_stats(vdb_ivf* h, int* used_matrix_scan, int* overflowed);
/* bytes one search of this batch reads from the lists (sum of probed list lengths * dim * 4), for roofline accounting */
int vdb_ivf_last_scan_rows(vdb_ivf* h, int64_t* rows_scanned);

/* ---- PQ -------------------------------------------------------------------- */
/* PQIndex._encode for a batch (index/pq.py:188-200) */
int vdb_pq_encode(const float* x, int64_t n, const float* codebook, int m, int ksub, int dsub, uint8_t* out_codes);

typedef struct vdb_pq vdb_pq;
int vdb_pq_create(vdb_pq** out, int dim, int m, int ksub);
int vdb_pq_destroy(vdb_pq* h);
int vdb_pq_set_codebook(vdb_pq* h, const float* codebook);                  /* pq.py:140 */
int vdb_pq_append_codes(vdb_pq* h, const uint8_t* codes, int64_t n);         /* pq.py:229 */
int vdb_pq_append_codes_dev(vdb_pq* h, const uint8_t* d_codes, int64_t n, void* stream);
int vdb_pq_append_rows(vdb_pq* h, const float* rows, int64_t n, uint8_t* out_codes_or_null); /* pq.py:228-231 */
int vdb_pq_truncate(vdb_pq* h, int64_t n);
int vdb_pq_set_code(vdb_pq* h, int64_t row, const uint8_t* code);
int vdb_pq_size(const vdb_pq* h, int64_t* n);
/* distance tables of one query (pq.py:270-283): m*ksub floats */
int vdb_pq_lut(vdb_pq* h, const float* query, float* out_lut);
/* PQIndex.search / search_batch (pq.py:247-327): tables, ADC scan with the
 * bitmask tested in the scan, sqrt, top-k ascending (distance, row). */
int vdb_pq_search(vdb_pq* h, const float* queries, int nq, int k,
                  const uint32_t* mask, float* out_dist, int64_t* out_row);
int vdb_pq_search_dev(vdb_pq* h, const float* d_queries, int nq, int k,
                      const uint32_t* d_mask, float* d_out_dist, int64_t* d_out_row, void* stream);
/* vdb_pq_search with a resident mask (index/pq.py:287-297, the filter_fn test inside the ADC scan) */
int vdb_pq_search_masked(vdb_pq* h, const float* queries, int nq, int k,
                         const vdb_mask* mask, float* out_dist, int64_t* out_row);
/* Diagnostics of the last search on this handle.  Batches of two queries and more over large code sets (8-bit
 * codes, m in {4, 8, 16, 32}, k <= 32) pass over the codes eight queries at a time on one-byte tables that bound
 * the fp32 table entries from below, and take the reference's fp32 sums (index/pq.py:296-297) only for the rows
 * those bounds cannot exclude: same rows, same distance bits.  *used_byte_tables = 1 when the search went that way,
 * *overflowed = 1 when it could not answer the batch (no bound for a query, too many rows left) and the fp32 scan
 * repeated it.  Waits for the search.  No counterpart in the reference. */
int vdb_pq_search_stats(vdb_pq* h, int* used_byte_tables, int* overflowed);

/* ---- IVF-PQ (HybridIndex, index/hybrid.py:41-536) ----------------------------
 * Coarse k-means centroids + a PQ codebook trained on residuals.  Rows live as
 * m-byte codes in inverted lists.  Search (hybrid.py:276-352): coarse ranking by
 * the identity-form squared distance, then per probed list the table of the
 * residual query and the ADC scan (fp32 sum in sub-space order, sqrt), results
 * ascending with ties in visiting order (probe rank, then position in the list). */
typedef struct vdb_ivfpq vdb_ivfpq;
int vdb_ivfpq_create(vdb_ivfpq** out, int dim, int m, int ksub);
int vdb_ivfpq_destroy(vdb_ivfpq* h);
/* trained quantizers (hybrid.py:146,165); empties the lists */
int vdb_ivfpq_set_quantizers(vdb_ivfpq* h, const float* centroids, int n_lists, const float* codebook);
/* rows -> (list, code): nearest coarse centroid (hybrid.py:205-210), residual, PQ code (:212-224) */
int vdb_ivfpq_encode(vdb_ivfpq* h, const float* rows, int64_t n, int32_t* out_list, uint8_t* out_codes);
/* lists in scan order: codes[n][m] by row number, the row numbers of every list
 * concatenated, n_lists+1 offsets (hybrid.py:262 appends, :405-409 removes in place) */
int vdb_ivfpq_build(vdb_ivfpq* h, const uint8_t* codes, int64_t n, const int64_t* list_rows, const int64_t* list_offsets);
/* codes already in list order, a row number of the caller's choice for each (as vdb_ivf_build_numbered) */
int vdb_ivfpq_build_numbered(vdb_ivfpq* h, const uint8_t* codes, int64_t n, const int64_t* row_numbers, const int64_t* list_offsets);
/* HybridIndex.add on a built index (hybrid.py:262 appends to the list), list by list as vdb_ivf_append, and
 * remove (hybrid.py:396-400 filters the list: the codes behind the removed one move up in their order) */
int vdb_ivfpq_append(vdb_ivfpq* h, const uint8_t* codes, const int64_t* row_numbers, const int32_t* lists, int64_t count);
int vdb_ivfpq_remove_at(vdb_ivfpq* h, int32_t list, int64_t position);
int vdb_ivfpq_layout(vdb_ivfpq* h, int64_t* live_rows, int64_t* positions, int64_t* dead_positions);
int vdb_ivfpq_search(vdb_ivfpq* h, const float* queries, int nq, int k, int n_probe,
                     const uint32_t* mask, float* out_dist, int64_t* out_row);
int vdb_ivfpq_search_dev(vdb_ivfpq* h, const float* d_queries, int nq, int k, int n_probe,
                         const uint32_t* d_mask, float* d_out_dist, int64_t* d_out_row, void* stream);
/* vdb_ivfpq_search with a resident mask (index/hybrid.py:303-329, the filter_fn test inside the scan) */
int vdb_ivfpq_search_masked(vdb_ivfpq* h, const float* queries, int nq, int k, int n_probe,
                            const vdb_mask* mask, float* out_dist, int64_t* out_row);

/* ---- cross-shard merge ------------------------------------------------------
 * No counterpart in the reference (it is single-process).  After an
 * all-gather of every shard's (distance, global row) lists: n_lists lists of k
 * per query -> the k best per query under the (distance, row) order.
 * Layout of the inputs: [n_lists][nq][k].  Device pointers. */
int vdb_topk_merge_dev(const float* d_dist, const int64_t* d_row, int n_lists, int nq, int k,
                       float* d_out_dist, int64_t* d_out_row, void* stream);
/* Same for lists that do not lie back to back and carry shard-local rows: list
 * l of the distances starts at d_dist + l*list_stride_dist, of the rows at
 * d_row + l*list_stride_row (strides in elements, each >= nq*k), and
 * d_row_base[l] (device, n_lists values, or NULL) is added to the rows of list
 * l before they are compared and written.  Lets every shard send distances
 * and local rows in ONE all-gather of a packed block. */
int vdb_topk_merge_strided_dev(const float* d_dist, const int64_t* d_row, int64_t list_stride_dist, int64_t list_stride_row,
                               const int64_t* d_row_base, int n_lists, int nq, int k,
                               float* d_out_dist, int64_t* d_out_row, void* stream);
/* Same, ordered by (distance, key) with the rows carried along: the merge of
 * IVF shards, key = the visiting order of vdb_ivf_search_ordered_dev. */
int vdb_topk_merge_keyed_dev(const float* d_dist, const int64_t* d_key, const int64_t* d_row, int n_lists, int nq, int k,
                             float* d_out_dist, int64_t* d_out_row, void* stream);
/* The same over lists that lie list_stride_dist floats / list_stride_key_row int64 apart (keys and rows share
 * one stride): the blocks of ONE packed all-gather, [nq*k distances | pad to 8 bytes | nq*k keys | nq*k rows] per
 * rank, are merged where they arrived.  No counterpart in the reference (nothing there shards). */
int vdb_topk_merge_keyed_strided_dev(const float* d_dist, const int64_t* d_key, const int64_t* d_row,
                                     int64_t list_stride_dist, int64_t list_stride_key_row, int n_lists, int nq, int k,
                                     float* d_out_dist, int64_t* d_out_row, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDB_CORE_H */
