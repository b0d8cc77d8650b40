// Internal declarations shared by the kernels and the C-ABI layer of
// libvdbcore.so.  gfx950 (MI355X) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vdb_core.h"

namespace vdb {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);

#define VDB_HIP(call)                                                                \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            vdb::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
                           __FILE__, __LINE__);                                      \
            return (e__ == hipErrorOutOfMemory) ? VDB_ERR_NOMEM : VDB_ERR_HIP;       \
        }                                                                            \
    } while (0)

#define VDB_REQUIRE(cond, ...)                                                       \
    do {                                                                             \
        if (!(cond)) {                                                               \
            vdb::set_error(__VA_ARGS__);                                             \
            return VDB_ERR_INVALID;                                                  \
        }                                                                            \
    } while (0)

#define VDB_TRY(expr)                                                                \
    do {                                                                             \
        int s__ = (expr);                                                            \
        if (s__ != VDB_OK) return s__;                                               \
    } while (0)

// ---------------------------------------------------------------- device state
struct Context {
    int device = -1;                // the one GPU this process uses; fixed by the first vdb_init
    int num_cus = 256;
    float* zeros = nullptr;         // 256 zero bytes in device memory
    void* scratch = nullptr;        // device address of the static scratch array (no-allocation entry point)
    hipEvent_t scratch_done = nullptr;  // recorded behind the last call that used the array; the next one waits for it
    bool scratch_used = false;
    std::mutex mu;                  // guards initialisation only
};
Context& ctx();
int ensure_init();

// Base of every handle.  Calls on one handle are serialised by its mutex (they share its
// workspaces); different handles run concurrently.  The host-pointer entry points of a handle
// enqueue on its own stream, a blocking stream, i.e. ordered with the null stream.
struct Handle {
    std::mutex mu;
    hipStream_t stream = nullptr;
    int open();                     // creates the stream
    void close();                   // waits for the stream and destroys it
};
// Scope of one entry point on a handle: takes the handle's mutex and makes the library's device
// current on the calling thread (the HIP device is a per-thread setting).
struct Entry {
    std::lock_guard<std::mutex> g;
    explicit Entry(Handle* h) : g(h->mu) { (void)hipSetDevice(ctx().device); }
};

// HIP-event bracket around a dominant kernel launch (no-op unless profiling is on)
struct KernelTimer {
    hipStream_t s;
    hipEvent_t start = nullptr;
    explicit KernelTimer(hipStream_t stream, bool enabled = true);
    ~KernelTimer();
};

// A fixed block of device memory handed out front to back, never returned: the workspaces of a
// call that must not allocate are carved out of it.
struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
};

// A grow-only device buffer (workspace).  Bound to an arena it takes its memory from there
// instead of hipMalloc and fails with VDB_ERR_INVALID when the arena is exhausted.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    Arena* arena = nullptr;
    int reserve(size_t bytes);      // may reallocate (contents lost)
    int reserve_keep(size_t bytes, size_t keep_bytes, hipStream_t s);  // grows geometrically, keeps a prefix
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Dynamic LDS a kernel may be launched with (hipFuncAttributeMaxDynamicSharedMemorySize).  Only ever raised,
// under a lock, so concurrent launches of one kernel with different sizes cannot undercut each other.
int raise_dynamic_lds(const void* kernel, size_t bytes);

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- summation plan
// NumPy sums a contiguous fp32 run of length n pairwise: blocks of at most 128
// elements are reduced with 8 strided accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail; longer runs are
// split at (n/2 rounded down to a multiple of 8) and the halves added.  The
// plan lists the leaf blocks left to right and, after each leaf, how many
// pending partial sums are folded (right into left).  Kernels that must match
// np.sum bit for bit walk this plan.
struct SumPlan {
    static const int MAX_LEAVES = 64;      // dim <= 8192
    int n_leaves;
    short leaf_len[MAX_LEAVES];
    unsigned char folds[MAX_LEAVES];
};
int make_sum_plan(int n, SumPlan* plan);

// ---------------------------------------------------------------- row store
struct Rows {
    int dim = 0;
    int ld = 0;             // floats per stored row: dim rounded up to 4, zero padded
    int64_t n = 0;
    int64_t cap = 0;
    float* data = nullptr;  // [cap][ld]   (fp32 storage)
    bool bf16 = false;      // bf16 storage: data16 holds the rows, data is unused
    int ld16 = 0;           // bf16 elements per stored row: dim rounded up to 8, zero padded
    uint16_t* data16 = nullptr;   // [cap][ld16]
    float* sq = nullptr;    // [cap]  np.sum(row**2)
    float* nrm = nullptr;   // [cap]  sqrt(sq)
    float cos_eps = 1e-8f;  // norms below this count as 1 in the cosine distance matrix (pairwise_cosine's eps)
    SumPlan plan;
};

// ---------------------------------------------------------------- kernel launchers (device pointers)
enum { KMAX_FUSED = 32 };   // largest k the fused selection keeps per thread in LDS

// copy n rows of `dim` floats (source leading dimension src_ld) into padded rows of dst_ld floats
int launch_pad_rows(const float* src, int64_t src_ld, int dim, int64_t n, float* dst, int dst_ld, hipStream_t s);

// MFMA path: fused distance + per-slab top-k for the four dot-product metrics.
// part_* hold n_cand candidate lists of k per query: layout [nq][n_cand][k].
int flat_gemm_num_cand(int64_t n, int nq, int k, int* n_cand);
int flat_gemm_num_cand_tiles(int n_tiles, int nq, int* n_cand);
// strided row sample used to warm-start the selection threshold: number of
// 128-row tiles to visit (0 = not worth it) and the stride between them
int flat_gemm_sample_plan(int64_t n, int* sample_tiles, int* tile_stride);
int launch_flat_gemm_topk(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm,
                          int nq, int k, const uint32_t* mask, const float* lb_d, const int64_t* lb_k,
                          const float* thr_init, int sample_tiles, int tile_stride,
                          float* part_d, int* part_i, int n_cand, hipStream_t s, const int* run_if = nullptr,
                          const int* q_count = nullptr);
// `q_count` (device): only the first *q_count of the nq queries are searched (the grid is laid out for nq).
// MFMA path with a known per-query bound of the k-th best distance (`thr`): survivors of the
// bound go to queues of `cap` entries per (query, slab) and the k best of them come out directly.
// `*overflow` is set to 1 when a queue was too small (the caller then repeats the search with
// launch_flat_gemm_topk(..., run_if = overflow)); it must be zero on entry.
// Bound pass for stores the branch-free kernel can read (flat_queue_fast_ok): per-lane minima over
// `sample_tiles` tiles (every tile_stride-th), then thr[q] = k-th smallest of query q's minima.
// `mins` needs nq * flat_min_sample_lists(...) floats.  A non-null `clear_flag` is set to 0 on the way
// (the overflow flag of the queue pass that follows).
bool flat_queue_fast_ok(const Rows& r);
int flat_min_sample_lists(int sample_tiles, int nq, int k, int* n_slabs);
int launch_flat_min_sample(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm, int nq, int k,
                           const uint32_t* mask, int sample_tiles, int tile_stride, float* mins, float* thr,
                           int* clear_flag, hipStream_t s, int* clear_marks = nullptr);
int flat_queue_plan(const Rows& r, int nq, int k, int sample_stride, int* n_slabs, int* cap, bool loosened = false);
// Nearest row of every query (k = 1; VDB_EUCLIDEAN or VDB_EUCLIDEAN_SQ; fp32 rows that flat_queue_fast_ok accepts; no
// filter): out_d / out_i / out_i32 [q * out_stride] get the list kernel's result wherever the smallest key is clear
// of the second smallest; the other queries are appended to list[] (at most lim of them; *count, zero on entry, counts
// all) and must be searched again by the list kernel.  arg_buf: nq * flat_argmin_lists(r, nq) * 16 bytes.
int flat_argmin_lists(const Rows& r, int nq);
int launch_flat_argmin(const Rows& r, int metric, const void* q, const float* q_sq, int nq, void* arg_buf,
                       float* out_d, int64_t* out_i, int* out_i32, int out_stride, int* list, int* count, int lim, hipStream_t s);
// The listed queries' padded rows and norm terms to compact arrays (many blocks); *q_count = their number, or 0 with
// *all_flag = 1 when more were wanted than the list holds.
int launch_redo_gather_list(const int* list, const int* count, int lim, const float* q, int ldq, const float* q_sq,
                            const float* q_nrm, float* q_out, float* q_sq_out, float* q_nrm_out, int* q_count, int* all_flag,
                            hipStream_t s);
// bf16 rows with whole 128-byte column steps and whole readable 256-row tiles take the 256 x 256 kernel
bool queue_use_wide(const Rows& r);
// Exact search behind a bf16 filter: `r` and `q` of launch_flat_queue are then bf16 copies (rounded to nearest) of
// the fp32 rows and queries below, with the norm terms of the fp32 values.  The queue pass runs on the copies with
// its bound loosened by the largest difference rounding can make (slack[query], in dot-product units), and the
// selection recomputes the fp32 dot product of every queued row that can still be among the k best, with the same
// instruction sequence as the fp32 kernels, so distances and order are those of the fp32 search.
struct PrefilterArgs {
    const float* rows32; int ld32;      // fp32 rows, pitch in floats
    const float* q32; int ldq32;        // fp32 queries of this launch, padded like the rows
    const float* slack;                 // [nq]
};
// slack[q] from the query norms and the largest row norm; sets *overflow (the flag of the queue pass, which makes the
// fp32 list kernel repeat the search) when a dot product could overflow or is not a number.  With `widen_thr`
// (per-query bounds of the k-th best distance found on the bf16 copies) every bound is raised to what the exact
// distances of the same k rows cannot exceed.
int launch_prefilter_slack(int metric, const float* q_nrm, const float* max_row_nrm, int nq, int dim, float* slack,
                           float* widen_thr, int* overflow, hipStream_t s, int* ovf_mark = nullptr);
// max over row_nrm[n0, n1) folded into *max_nrm (a device float that starts at 0; a NaN norm wins)
int launch_max_norm(const float* row_nrm, int64_t n0, int64_t n1, float* max_nrm, hipStream_t s);
int launch_flat_queue(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm, int nq, int k,
                      const uint32_t* mask, const float* thr, int n_slabs, int cap, void* queue, int* counts,
                      int* overflow, float* out_d, int64_t* out_i, int* out_i32, hipStream_t s,
                      const PrefilterArgs* pf = nullptr, int* ovf_mark = nullptr);
// The queries whose queues overflowed (ovf_mark[q] != 0), searched again alone.  launch_redo_gather (one block) lists
// them in ascending order, writes their number to *count and copies their padded rows, norm terms and bounds to
// compact arrays; the list kernel then runs over those (q_count = count, a grid laid out for `limit` queries so that
// it fills the GPU), the selection writes compact results, and launch_redo_scatter moves result row j to query
// list[j].  More than `limit` marked queries: *count = 0 and *all_flag = 1, the caller's whole-batch repeat runs behind
// all_flag instead.  Nothing runs while *flag is zero (*count = 0, *all_flag = 0).
int flat_gemm_query_tile();
int launch_redo_gather(const int* flag, const int* ovf_mark, int nq, const float* q, int ldq, const float* q_sq, const float* q_nrm,
                       const float* thr, int* list, int* count, float* q_out, float* q_sq_out, float* q_nrm_out, float* thr_out,
                       int limit, int* all_flag, hipStream_t s);
int launch_redo_scatter(const int* list, const int* count, int nq, int k, const float* d, const int64_t* i64, const int* i32,
                        float* out_d, int64_t* out_i, int* out_i32, hipStream_t s);
// MFMA path, full distance matrix out[nq][n]
int launch_flat_gemm_dist(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm,
                          int nq, float* out, hipStream_t s);

// streaming pipeline: pairwise-ordered sums for (row chunk, query group) items -> candidate buffer
enum ScanOp { OP_DOT = 0, OP_SQDIFF = 1, OP_ABSDIFF = 2, OP_MAXABS = 3, OP_SQ = 4, OP_NEQ = 5 };
enum StreamPost { POST_NONE = 0, POST_SQRT = 1, POST_NEG = 2 };
static inline int op_of_metric(int metric) {
    switch (metric) {
        case VDB_EUCLIDEAN: case VDB_EUCLIDEAN_SQ: case VDB_COSINE: case VDB_DOT: case VDB_COSINE_PAIR: case VDB_ANGULAR: return OP_DOT;
        case VDB_EUCLIDEAN_DIRECT: case VDB_EUCLIDEAN_SQ_DIRECT: return OP_SQDIFF;
        case VDB_MANHATTAN: return OP_ABSDIFF;
        case VDB_CHEBYSHEV: return OP_MAXABS;
        case VDB_HAMMING: return OP_NEQ;
        default: return OP_SQ;
    }
}
static inline int post_of_metric(int metric) {
    return metric == VDB_EUCLIDEAN_DIRECT ? POST_SQRT : (metric == VDB_DOT ? POST_NEG : POST_NONE);
}
struct StreamItem {                 // one (row chunk, query group) unit of work
    int64_t row0;                   // first stored row of the chunk
    int64_t out_rel;                // chunk's first row relative to the first row of its segment
    int rows;                       // rows in the chunk (1..256)
    int qbeg; int qcnt;             // the group's entries in sq_qid / sq_out
    int pad;
};
struct StreamPlanBufs {             // device arrays describing the items of one launch
    int64_t* seg_row_off; int* seg_q_off; int64_t* seg_item_off;   // [n_seg + 1] each
    int* sq_qid; int64_t* sq_out;                                  // one entry per (segment, query) pair
    StreamItem* items; int64_t items_cap;                          // item table and its capacity
};
struct StreamJob {
    const float* data; int ld; int dim; SumPlan plan;
    int op; int post;
    const float* q; int ldq;        // query rows; readable up to 8 floats past the last row
    int group;                      // queries per item: 1, 8, 16 or 128 (stream_batch_group)
    int n_seg;
    const int64_t* seg_row_off; const int* seg_q_off; const int64_t* seg_item_off;
    const int* sq_qid; const int64_t* sq_out;
    const StreamItem* items; int64_t items_cap;
    float* out;
    bool timed;
    const int* run_if;              // optional device flag: the launch does nothing while it is zero
    // > 0: the number of items of the table; the blocks then take them back to front.  A scan that is repeated over a
    // row set larger than the memory-side cache alternates direction, so each pass starts on the rows the last one
    // ended on (k-means++ runs thousands of such passes)
    int64_t reverse_items;
    // bounded form (group 128, flat row sets): pairs within bound[query] whose row passes `mask` go to per-query
    // queues of queue_cap (distance, row) entries instead of `out`; *overflow is set when a queue is full
    const float* bound; bool bound_sqrt; const uint32_t* mask;
    int* queue_len; float* queue_d; int* queue_i; int queue_cap; int* overflow;
};
int launch_stream_sums(const StreamJob& j, hipStream_t s);
// queries per item a batch job (nq > 1) over rows of this shape is planned with: 8 or 16, or 128 where the
// row-resident kernel applies (rows of 32, 64 or 128 floats that are one leaf of the summation plan, and enough
// row chunks to give every block several items; n_rows < 0: an IVF list scan)
int stream_batch_group(int dim, int ld, const SumPlan& plan, int op, int nq, int64_t n_rows);
// row norms in NumPy order: sq[i] = np.sum(row_i**2), nrm[i] = sqrt(sq[i]); b needs ceil(n/256) items
int launch_row_norms(const float* data, int ld, int dim, int64_t n, const SumPlan& plan, const StreamPlanBufs& b,
                     float* sq, float* nrm, hipStream_t s);
int launch_flat_plan(int64_t n, int nq, int group, int64_t out_stride, const StreamPlanBufs& b, hipStream_t s);
int launch_ivf_slot_offsets(const int* probes, int nq, int n_probe, const int64_t* list_beg, const int64_t* list_end, int n_lists,
                            int64_t* cand_off, int* tile_off, hipStream_t s);
int launch_ivf_units_by_list(const int* probes, int nq, int n_probe, const int64_t* list_beg, const int64_t* list_end, int n_lists,
                             int64_t cand_stride, int* list_cnt, int* cursor, int64_t* cand_off, int* tile_off,
                             const StreamPlanBufs& b, int* sq_list, int blocks, int* first, hipStream_t s);
int launch_ivf_plan(const int* probes, int nq, int n_probe, const int64_t* list_beg, const int64_t* list_end, int n_lists, int group,
                    int64_t cand_stride, int* list_cnt, int* cursor, int64_t* cand_off, int* tile_off,
                    const StreamPlanBufs& b, hipStream_t s);
// in-place distance finish of a [nq][n] matrix of streamed sums (identity-form L2 and cosine only)
int launch_finish_dist(int metric, const float* q_sq, const float* q_nrm, const float* row_sq, const float* row_nrm,
                       float cos_eps, float* v, int nq, int64_t n, hipStream_t s);
// per query: finish + filter + k best of its slice of the candidate buffer -> part lists
// [nq][splits * topk_cand_lists_per_split()][k]
int topk_cand_lists_per_split();
struct CandJob {
    const float* cand; int64_t cand_stride; int nq;
    int64_t n;                                        // flat: candidates per query (slot = stored row)
    const int* probes; int n_probe; int n_lists; const int64_t* list_beg; const int64_t* list_end;   // lists (probes == null: flat)
    const int64_t* cand_off; const int* tile_off;     // lists: [nq][n_probe+1] from launch_ivf_plan
    const int* row_ids; const uint32_t* mask;
    int metric; const float* q_sq; const float* q_nrm; const float* row_sq; const float* row_nrm;
    const float* lb_d; const int64_t* lb_k; float radius; int use_radius;
    int k; int splits; float* part_d; int* part_i;
    bool sqrt_here;                                   // VDB_EUCLIDEAN_DIRECT whose scan stored the sums without the root
    // lists only: one launch for selection, merge and row lookup (the last block of a query merges); ticket = one
    // zeroed int per query, k_out = results per query (>= k; the rest is filled with INFINITY / -1)
    int* ticket; int k_out; float* fin_d; int64_t* fin_rows; int64_t* fin_order;
    const int* run_if;                                // optional device flag: the launch does nothing while it is zero
    // flat candidates whose count differs per query (the slots of an IVF query are one contiguous run when no filter
    // needs the rows behind them): query q has n_of_query[q * n_stride] candidates
    const int64_t* n_of_query; int n_stride;
    bool block_list;                                  // one list per block instead of one per wave: part_* is [nq][splits][k]
};
// one query against an IVF index: norms, centroid ranking, slot/tile offsets and scan items in one block
int launch_ivf_one_plan(const float* sums, int metric, const float* cent_sq, const float* cent_nrm, float cos_eps, const float* q,
                        const SumPlan& plan, int n_lists, int n_probe, const int64_t* list_beg, const int64_t* list_end,
                        float* q_sq, float* q_nrm, int* probes, float* probe_d, int64_t* cand_off, int* tile_off,
                        const StreamPlanBufs& b, hipStream_t s);
int launch_topk_cand(const CandJob& j, hipStream_t s);
// range search over the streamed sums of ONE query (candidate slot = stored row): count + scan of the rows inside
// the radius, then the first `cap` of them in row order as sorted (distance, row) results
struct RangeJob {
    const float* cand; int64_t n; const uint32_t* mask;
    int metric; const float* q_sq; const float* q_nrm; const float* row_sq; const float* row_nrm;
    float radius;
};
int64_t range_blocks(int64_t n);                  // entries block_counts / block_off need
int64_t range_sort_capacity(int64_t m);           // keys the sort of m results needs
int launch_range_count(const RangeJob& j, int* block_counts, int64_t* block_off, int64_t* total, hipStream_t s);
int launch_range_collect(const RangeJob& j, const int64_t* block_off, int64_t cap, int64_t m, uint64_t* keys,
                         float* out_d, int64_t* out_row, hipStream_t s);
int launch_ivf_rows(const int* slots, int nq, int k, int out_stride, int out_off, const int* probes, int n_probe, int n_lists,
                    const int64_t* cand_off, const int64_t* list_beg, const int* row_ids, int64_t* out_rows,
                    int64_t* out_order, hipStream_t s, const int* run_if = nullptr);

// final selection: per query, the k best of n_cand ascending candidate lists of k,
// layout [nq][n_cand][k], under the (distance, tie key) order with
// tie key = (candidate list / tie_group) << 32 | idx.  idx_map (optional)
// translates a selected idx to a row number; idx_offset is added afterwards.
// Batched IVF list scan on the matrix cores behind an exact recomputation (ivf_mfma.hip): bound pass, threshold,
// scan pass into per-query queues, exact selection.  `plan` is the list-major plan of launch_ivf_plan (items of at
// most 32 queries) with sq_rank beside sq_qid / sq_out; outputs are written for k results per query at stride k_out.
struct IvfMfmaJob {
    const uint16_t* rows16; int ld16; const float* row_sq;     // bf16 copies of the stored rows and |c|^2 of the fp32 rows
    const float* rows; int ld; int dim; SumPlan sum_plan;      // the fp32 rows
    const float* q; int ldq; const uint16_t* q16; const float* q_sq; const float* q_nrm; int nq;
    StreamPlanBufs plan; int64_t cand_stride;
    const int* probes; int n_probe; int n_lists; const int64_t* cand_off; const int64_t* list_beg;
    const int* row_ids; const uint32_t* mask;
    const float* max_row_nrm;                                  // device float: the largest |c| over the stored rows
    float* mins; float* thr; float* slack; int4* seg; int* seg_len; int seg_cap;                         // workspaces
    int* queue_slot; float* queue_d; int* queue_pos; int* queue_len; int queue_cap; int* overflow;      // queue_d holds the keys
    int* slots;                                                // workspace [nq][k_out]: the selected candidate slots
    int k, k_out; bool take_sqrt;
    float* out_d; int64_t* out_rows; int64_t* out_order;
    bool timed;
};
int launch_ivf_mfma_search(const IvfMfmaJob& j, hipStream_t s);
// The lists a batch of queries probes, by the same two steps: keys of every (query, centroid) pair on the matrix cores,
// then per query the few centroids whose keys can be among the n_probe nearest recomputed in the reference's
// arithmetic and ranked by (distance, centroid number).
struct IvfProbeJob {
    const uint16_t* cent16; const float* cent_sq; const float* max_cent_nrm;     // row-major bf16 copy, |c|^2, largest |c| (device)
    const float* cent; int ld; int dim; SumPlan sum_plan; int n_lists;           // the fp32 centroids
    const float* q; int ldq; const uint16_t* q16; int ld16; const float* q_sq; const float* q_nrm; int nq;
    int n_probe; bool take_sqrt;
    float* keys;                                                                  // workspace [nq][n_lists]
    int* probes; float* probe_d;                                                  // [nq][n_probe]
};
int launch_ivf_mfma_probe(const IvfProbeJob& j, hipStream_t s);
// the k <= 64 smallest of every row of a [nq][n] matrix under (value, column): one block per row (rows that hold
// over a thousand equal smallest values are overwritten in the process)
int launch_rank_rows(float* dist, int nq, int n, int k, float* out_d, int* out_i, hipStream_t s);
// the k <= 64 best of every query's run of candidate distances (cand + q * cand_stride, cand_off[q][n_probe] of them, no
// filter) under (distance, slot): distances and slots at stride k_out
int launch_cand_rank(const float* cand, int64_t cand_stride, const int64_t* cand_off, int n_probe, int nq, int k, int k_out,
                     float* out_d, int* out_slots, hipStream_t s, int parts = 1, float* part_d = nullptr, int* part_i = nullptr,
                     bool squares = false);   // squares: the run holds squares of the distances
int ivf_mfma_pitch(int dim);            // bf16 elements per row of the copy
int ivf_mfma_subsets();                 // floats of `mins` per query
int ivf_mfma_len_stride();              // ints of `queue_len` per query
int ivf_mfma_scan_waves();              // segments of `seg` / entries of `seg_len`
int launch_ivf_round_rows(const float* src, int64_t src_ld, int dim, int64_t n, uint16_t* dst16, int ld16, hipStream_t s);
// rows [pos0, pos0 + n) of a store (src = its row pos0) into the tiled bf16 copy `dst16` (the copy's base address)
int launch_ivf_round_rows_tiled(const float* src, int64_t src_ld, int dim, int64_t pos0, int64_t n, uint16_t* dst16, int ld16,
                                hipStream_t s);

struct SelectOut {
    float* out_d; int64_t* out_i; int* out_i32;   // any of the index outputs may be null
    int out_stride; int out_off;                  // result r of query q goes to [q*out_stride + out_off + r]
    float* last_d; int64_t* last_key;             // (distance, tie key) of the last result, for a follow-up pass
    const int* run_if;                            // optional device flag: the launch does nothing while it is zero
};
// `mid` = four workspace buffers of the caller (distances and ids of two intermediate levels), used when more
// than one block's worth of lists has to be merged hierarchically.
int launch_select(const float* part_d, const int* part_i, int nq, int n_cand, int k, int tie_group,
                  const int* idx_map, int64_t idx_offset, const SelectOut& o, hipStream_t s, DevBuf* mid);
// a few queries: the k best of every query's packed candidate area (queue_len[q] of `area` entries, any order),
// one block per query; sets queue_len[q] back to zero (query q's length at queue_len[q * len_stride]).
// `redo` (optional: one device int that is zero before the call, k <= 16): a short kernel answers the queries with up
// to 512 entries per wave first and sets the flag for the others, which the general kernel then answers.
int launch_select_packed(const float* part_d, const int* part_i, int nq, int area, int k, const SelectOut& o, hipStream_t s,
                         int* queue_len, int len_stride = 1, int waves = 16, int* redo = nullptr);
// the same over whole areas (every entry present or marked empty by its id): no lengths
int launch_select_area(const float* part_d, const int* part_i, int nq, int area, int k, const SelectOut& o, hipStream_t s);
// lists laid out [n_lists][nq][k] with int64 ids (negative = empty), ordered by (distance, id);
// with a payload of the same layout the payload of every selected entry is written instead of its id.
// List l of the distances starts at d + l * stride_d, of the ids and the payload at i + l * stride_i
// (elements); with list_base, list_base[l] is added to the ids of list l before comparing and writing.
int launch_select_i64(const float* d, const int64_t* i, const int64_t* payload, int64_t stride_d, int64_t stride_i,
                      const int64_t* list_base, int n_lists, int nq, int k, float* out_d, int64_t* out_i, hipStream_t s);

// sort every row of a [nq][n] matrix (n <= 16384) ascending by (value, index); writes the first k of each
int launch_row_sort_topk(const float* dist, int nq, int n, int k, float* out_d, int* out_i, hipStream_t s);

// PQ
int launch_pq_lut(const float* q, int ldq, int nq, const float* book, int m, int ksub, int dsub,
                  const SumPlan& plan, float* lut, hipStream_t s, int* zero_counts = nullptr, float* mins = nullptr,
                  int* zero_more = nullptr, int zero_more_n = 0);
int launch_pq_encode(const float* x, int64_t n, int ldx, const float* book, int m, int ksub, int dsub,
                     const SumPlan& plan, uint8_t* codes, hipStream_t s);
// candidate lists come out as [nq][splits * pq_scan_lists_per_split()][k]
int pq_scan_lists_per_split();
int launch_pq_scan(const uint8_t* codes, int64_t n, int m, int ksub, const float* lut, int nq, int k,
                   const uint32_t* mask, const float* lb_d, const int64_t* lb_k, int splits,
                   float* part_d, int* part_i, float* dist_out, hipStream_t s, const float* bound_d = nullptr,
                   int64_t bound_stride = 0, bool timed = true, int* queue_len = nullptr, const int* run_if = nullptr,
                   int max_q = 4);
// Batches of two queries and more: eight queries per pass on one-byte tables (lower bounds derived from bound_d),
// then the fp32 sums of the rows that pass, into per-query candidate areas [nq][area] of queue_len[q * len_stride]
// entries.  `overflow` (one int, zero on entry) is set when the pass cannot answer the batch.  pq_scan8.hip
struct PqScan8Job {
    const uint8_t* codes; int64_t n; int m;
    const float* lut; int nq;                      // fp32 tables [nq][m][256]
    const float* lut_mins;                         // [nq][m]: the smallest entry of every table row (launch_pq_lut)
    const float* bound_d; int64_t bound_stride;    // per query: a distance that k rows of the set do not exceed
    const uint32_t* mask;
    int splits;                                    // pq_scan8_splits(n, nq)
    void* t8;                                      // pq_scan8_table_bytes(nq, m)
    float* thr;                                    // [nq]
    float* bound_keep;                             // [nq]: copy of the bounds (bound_d may lie in the output buffer)
    void* seg; int* seg_len; int seg_cap;          // groups * splits * 4 segments of seg_cap (row, query bits) pairs
    float* area_d; int* area_i; int area;
    int* queue_len; int len_stride;
    int* overflow;
    bool timed;
};
bool pq_scan8_supports(int m, int ksub);
int pq_scan8_groups(int nq);
int pq_scan8_splits(int64_t n, int nq);
size_t pq_scan8_table_bytes(int nq, int m);
int launch_pq_scan8(const PqScan8Job& j, hipStream_t s);
// IVF-PQ (index/hybrid.py): residuals, and the per-(query, probe) table + list scan into candidate slots
int launch_residual(const float* x, int64_t ldx, const float* cent, int ldc, const int* list, int dim, int64_t n,
                    float* out, hipStream_t s);
struct IvfPqUnits { const int* list; const int* query; const int64_t* out; const int* first; };   // launch_ivf_units_by_list
int ivfpq_scan_blocks();             // blocks of the two-table scan: first[0..blocks] of the units in list order
bool ivfpq_scan_takes_units(int m, int ksub, int dsub, int dim, int nq, int n_probe);
int launch_ivfpq_scan(const float* q, int ldq, int dim, int nq, const float* cent, int ldc, const float* book, int m, int ksub,
                      int dsub, const SumPlan& plan, const uint8_t* codes, const int64_t* list_beg, const int64_t* list_end, int n_lists,
                      const int* probes, int n_probe, const int64_t* cand_off, int64_t cand_stride, float* cand, hipStream_t s,
                      const struct IvfPqUnits* units = nullptr);
// measured ceilings of this GPU (ceilings.hip): device-to-device copy GB/s (read + write), register-only MFMA loops
int measure_ceilings(hipStream_t s, float* copy_gbs, float* mfma_f32_tflops, float* mfma_bf16_tflops, float* valu_direct_tops);
// k-means
int launch_kmeans_accumulate(const float* x, int64_t n, int dim, int ldx, const int* assign, int n_clusters,
                             double* sums, unsigned long long* counts, hipStream_t s);

// k-means on resident rows (kmeans.hip): stable sort of the rows by cluster, sequential per-cluster sums, the
// float64 assignment of the later PQ steps, and the k-means++ draw
int km_sort_chunk(int64_t n, int64_t* chunk, int* n_blocks);
int launch_km_sort(const int* assign, int64_t n, int k, int* hist, int* totals, int* cluster_off, int* members, hipStream_t s);
int launch_km_sums(int form, const float* x, int ldx, int dsub, const int* members, const int* cluster_off, int k,
                   float* next32, double* next64, long long* counts, hipStream_t s);
int launch_km_shift(const float* a, const float* b, int64_t total, double* out, hipStream_t s);
int launch_km_gather(const float* x, int ldx, int dsub, const long long* rows, const int* dst_rows, int count,
                     float* dst32, double* dst64, hipStream_t s);
int launch_km_assign_f64(const float* x, int ldx, int dsub, int64_t n, const float* xsq, const double* cent, double* csq,
                         int k, int* assign, hipStream_t s);
int64_t km_pp_blocks(int64_t n);
int launch_km_gather_query(const float* x, int ldx, int dsub, const long long* row, const float* x_sq, const float* x_nrm,
                           float* q, float* q_sq, float* q_nrm, hipStream_t s, float* q16f = nullptr, int q16f_n = 0);
// the selection also copies the chosen row out as the next round's query, with the row store's norms of it
struct KmNextQuery { const float* x = nullptr; int ldx = 0; int dsub = 0; const float* x_sq = nullptr; const float* x_nrm = nullptr;
                     float* q = nullptr; float* q_sq = nullptr; float* q_nrm = nullptr;
                     float* q16f = nullptr; int q16f_n = 0; };      // ... and its elements rounded to bf16, zero-padded to q16f_n
// k-means++ round over a bf16 copy of the rows (kmeans.hip): distances, running minima and block sums in one pass
struct KmBf16Pass {
    const uint16_t* x16; int pieces;
    const float* x; int ldx; int dim;
    const float* x_sq; const float* max_nrm;
    const float* q; const float* q16f;
    const float* q_sq; const float* q_nrm;
    int64_t n;
    float* nearest; double* block_sums;
};
int launch_km_pp_bf16(const KmBf16Pass& p, hipStream_t s);
int launch_km_pp_step(const float* dist, int64_t n, bool first, float* nearest, double* block_sums, double u,
                      long long* chosen, long long* bad_weights, hipStream_t s,
                      const float* q_sq = nullptr, const float* row_sq = nullptr, const KmNextQuery* next = nullptr);
int launch_km_pp_select(const float* nearest, int64_t n, const double* block_sums, double u, long long* chosen,
                        long long* bad_weights, const KmNextQuery& nx, hipStream_t s);

}  // namespace vdb
