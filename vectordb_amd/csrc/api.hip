// C ABI of libvdbcore.so (see include/vdb_core.h): handles, workspaces and
// the orchestration of the kernels.  No arithmetic on the distances happens on
// the host; this file only moves buffers and launches kernels.
#include "vdb_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <unordered_map>

namespace vdb {

// ============================================================== plumbing
static thread_local std::string g_err;

int raise_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, size_t>> seen;
    std::lock_guard<std::mutex> g(mu);
    for (auto& e : seen) {
        if (e.first != kernel) continue;
        if (bytes <= e.second) return VDB_OK;
        VDB_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        e.second = bytes;
        return VDB_OK;
    }
    VDB_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    seen.emplace_back(kernel, bytes);
    return VDB_OK;
}

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

Context& ctx() {
    static Context c;
    return c;
}

int ensure_init() {
    if (ctx().device >= 0) return VDB_OK;
    return vdb_init(0);
}

// a stream of its own for an entry point that has no handle (one-shot conversions of host arrays)
struct ScopedHandle : Handle {
    ~ScopedHandle() { close(); }
};

// ---- kernel timing ---------------------------------------------------------
static bool g_profile = false;
static std::mutex g_prof_mu;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;

// launches under a ProfileMute stay out of the timing (the side kernels of a path whose main kernel is the one rated)
static thread_local int g_prof_mute = 0;
struct ProfileMute {
    bool on;
    explicit ProfileMute(bool on_) : on(on_) { if (on) ++g_prof_mute; }
    ~ProfileMute() { if (on) --g_prof_mute; }
};

KernelTimer::KernelTimer(hipStream_t stream, bool enabled) : s(stream) {
    if (!g_profile || !enabled || g_prof_mute > 0) return;
    if (hipEventCreate(&start) != hipSuccess) { start = nullptr; return; }
    (void)hipEventRecord(start, s);
}

KernelTimer::~KernelTimer() {
    if (!start) return;
    hipEvent_t stop = nullptr;
    if (hipEventCreate(&stop) != hipSuccess) { (void)hipEventDestroy(start); return; }
    (void)hipEventRecord(stop, s);
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof_events.emplace_back(start, stop);
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return VDB_OK;
    if (arena) {
        // carve the block out of the arena (a block grown within one call leaves its old place unused)
        const size_t at = (arena->used + 255) / 256 * 256;
        if (at + bytes > arena->size) {
            set_error("the static scratch of %zu bytes is too small for this call (%zu bytes in use, %zu more asked): "
                      "fewer queries or rows per call, or the allocating entry point", arena->size, arena->used, bytes);
            return VDB_ERR_INVALID;
        }
        p = arena->base + at;
        cap = bytes;
        arena->used = at + bytes;
        return VDB_OK;
    }
    if (p) { VDB_HIP(hipFree(p)); p = nullptr; cap = 0; }
    const size_t want = std::max(bytes, (size_t)256);
    VDB_HIP(hipMalloc(&p, want));
    cap = want;
    return VDB_OK;
}

int DevBuf::reserve_keep(size_t bytes, size_t keep_bytes, hipStream_t s) {
    if (bytes <= cap) return VDB_OK;
    if (arena) { set_error("a buffer bound to the static scratch cannot grow in place"); return VDB_ERR_STATE; }
    size_t want = std::max(bytes, cap + cap / 2);
    want = std::max(want, (size_t)4096);
    void* np = nullptr;
    VDB_HIP(hipMalloc(&np, want));
    if (p && keep_bytes) VDB_HIP(hipMemcpyAsync(np, p, keep_bytes, hipMemcpyDeviceToDevice, s));
    if (p) { VDB_HIP(hipStreamSynchronize(s)); VDB_HIP(hipFree(p)); }
    p = np;
    cap = want;
    return VDB_OK;
}

void DevBuf::release() {
    if (p && !arena) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int Handle::open() {
    VDB_TRY(ensure_init());
    (void)hipSetDevice(ctx().device);
    // a blocking stream: ordered with work the caller put on the null stream
    if (!stream) VDB_HIP(hipStreamCreateWithFlags(&stream, hipStreamDefault));
    return VDB_OK;
}

void Handle::close() {
    if (!stream) return;
    (void)hipStreamSynchronize(stream);
    (void)hipStreamDestroy(stream);
    stream = nullptr;
}

static void plan_rec(int n, std::vector<int>& leaves, std::vector<int>& folds) {
    if (n <= 128) { leaves.push_back(n); folds.push_back(0); return; }
    int h = n / 2;
    h -= h % 8;
    plan_rec(h, leaves, folds);
    plan_rec(n - h, leaves, folds);
    folds.back() += 1;
}

int make_sum_plan(int n, SumPlan* plan) {
    std::vector<int> leaves, folds;
    plan_rec(std::max(n, 1), leaves, folds);
    VDB_REQUIRE((int)leaves.size() <= SumPlan::MAX_LEAVES, "dimension %d is too large (max 8192)", n);
    memset(plan, 0, sizeof(*plan));
    plan->n_leaves = (int)leaves.size();
    for (size_t i = 0; i < leaves.size(); ++i) { plan->leaf_len[i] = (short)leaves[i]; plan->folds[i] = (unsigned char)folds[i]; }
    return VDB_OK;
}

// *_dev entry points enqueue on the caller's stream; NULL is the HIP null stream
static hipStream_t pick_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ============================================================== small kernels
__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t src_ld, int dim, int64_t n,
                                const int* __restrict__ perm, float* __restrict__ dst, int dst_ld) {
    const int64_t total = n * dst_ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dst_ld;
        const int c = (int)(i - r * dst_ld);
        const int64_t sr = perm ? perm[r] : r;
        dst[i] = c < dim ? src[sr * src_ld + c] : 0.0f;
    }
}

// fp32 rows -> bf16 rows (round to nearest even), plus the rounded values back in fp32 so
// that the norm terms describe exactly what the matrix cores multiply
__global__ void round_bf16_kernel(const float* __restrict__ src, int64_t src_ld, int dim, int64_t n,
                                  uint16_t* __restrict__ dst16, int ld16, float* __restrict__ dst32, int ld32) {
    const int w = max(ld16, ld32);
    const int64_t total = n * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / w;
        const int c = (int)(i - r * w);
        uint32_t b = 0;
        if (c < dim) {
            const uint32_t x = __float_as_uint(src[r * src_ld + c]);
            b = ((x & 0x7fffffffu) > 0x7f800000u) ? ((x >> 16) | 0x40u)        // NaN stays NaN
                                                  : ((x + 0x7fffu + ((x >> 16) & 1u)) >> 16);
        }
        if (c < ld16) dst16[r * ld16 + c] = (uint16_t)b;
        if (c < ld32) dst32[r * ld32 + c] = __uint_as_float(b << 16);
    }
}

__global__ void fill_results_kernel(float* d, int64_t* i, int64_t total) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (d) d[t] = INFINITY;
        if (i) i[t] = -1;
    }
}

static int grid_for(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, 16384); }

int launch_pad_rows(const float* src, int64_t src_ld, int dim, int64_t n, float* dst, int dst_ld, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(n * dst_ld)), dim3(256), 0, s, src, src_ld, dim, n,
                       (const int*)nullptr, dst, dst_ld);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

static int fill_results(float* d, int64_t* i, int64_t total, hipStream_t s) {
    if (total <= 0) return VDB_OK;
    hipLaunchKernelGGL(fill_results_kernel, dim3(grid_for(total)), dim3(256), 0, s, d, i, total);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// ============================================================== workspaces
// page-locked host memory: copies to and from it are asynchronous and need no staging by the runtime
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return VDB_OK;
        release();
        const size_t want = std::max<size_t>(bytes, 64 * 1024);
        VDB_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return VDB_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Workspace {
    PinBuf pin_in, pin_out;             // host ends of the small transfers of the host-pointer entry points
    DevBuf q_buf, q_sq, q_nrm, part_d, part_i, lb_d, lb_k, thr_d, thr_k, mask, out_d, out_i, dist, probes, probe_d, lut, stage;
    DevBuf queue, queue_cnt, flag, mins, seg, seg_len;
    DevBuf redo_mark, redo_list, redo_q, redo_terms, redo_out;    // queries whose queues overflowed, searched again alone
    DevBuf ticket;                      // one int per query, zero between launches (topk_cand's last-block merge)
    void* ticket_zeroed = nullptr;      // the allocation that has been cleared
    DevBuf mid[4];                      // intermediate levels of a hierarchical list merge (launch_select)
    // item table of the last flat scan, kept between calls: it depends on (rows, queries, group) only
    DevBuf fp_seg_row_off, fp_seg_q_off, fp_seg_item_off, fp_sq_qid, fp_sq_out, fp_items;
    int64_t fp_n = -1; int fp_nq = -1, fp_group = -1;
    DevBuf round32, q16, cand, seg_row_off, seg_q_off, seg_item_off, sq_qid, sq_out, cand_off, tile_off, list_cnt, cursor, slots, items, slot_list, unit_first;
    template <class F> void each(F&& f) {
        DevBuf* all[] = {&q_buf, &q_sq, &q_nrm, &part_d, &part_i, &lb_d, &lb_k, &thr_d, &thr_k, &mask, &out_d, &out_i, &dist,
                         &probes, &probe_d, &lut, &stage, &round32, &q16, &cand, &seg_row_off, &seg_q_off, &seg_item_off, &sq_qid, &sq_out,
                         &cand_off, &tile_off, &list_cnt, &cursor, &slots, &items, &slot_list, &unit_first,
                         &queue, &queue_cnt, &flag, &mins, &seg, &seg_len, &redo_mark, &redo_list, &redo_q, &redo_terms, &redo_out, &ticket, &mid[0], &mid[1], &mid[2], &mid[3],
                         &fp_seg_row_off, &fp_seg_q_off, &fp_seg_item_off, &fp_sq_qid, &fp_sq_out, &fp_items};
        for (DevBuf* b : all) f(b);
    }
    void release() {
        fp_n = -1;
        ticket_zeroed = nullptr;
        each([](DevBuf* b) { b->release(); });
        pin_in.release(); pin_out.release();
    }
    // every device buffer of this workspace comes out of `a` from now on (a fresh, empty workspace only)
    void bind(Arena* a) { each([a](DevBuf* b) { b->arena = a; }); }
};

static bool is_dot_metric(int m) {
    return m == VDB_EUCLIDEAN || m == VDB_EUCLIDEAN_SQ || m == VDB_COSINE || m == VDB_DOT;
}
static bool valid_metric(int m) { return m >= VDB_EUCLIDEAN && m <= VDB_ANGULAR; }
static const int QUEUE_Q_BLOCK = 1024;  // queries per launch of the candidate-queue kernel (bounds its workspace)
static const int GEMM_MIN_NQ = 4;     // below this the streaming scan reads the rows once per query and wins

static int reserve_stream_plan(Workspace& ws, int n_seg, int64_t n_pairs, int64_t items_cap, StreamPlanBufs* b) {
    VDB_TRY(ws.seg_row_off.reserve((size_t)(n_seg + 1) * sizeof(int64_t)));
    VDB_TRY(ws.seg_q_off.reserve((size_t)(n_seg + 1) * sizeof(int)));
    VDB_TRY(ws.seg_item_off.reserve((size_t)(n_seg + 1) * sizeof(int64_t)));
    VDB_TRY(ws.sq_qid.reserve((size_t)std::max<int64_t>(n_pairs, 1) * sizeof(int)));
    VDB_TRY(ws.sq_out.reserve((size_t)std::max<int64_t>(n_pairs, 1) * sizeof(int64_t)));
    VDB_TRY(ws.items.reserve((size_t)std::max<int64_t>(items_cap, 1) * sizeof(StreamItem)));
    b->items = ws.items.as<StreamItem>(); b->items_cap = items_cap;
    b->seg_row_off = ws.seg_row_off.as<int64_t>(); b->seg_q_off = ws.seg_q_off.as<int>();
    b->seg_item_off = ws.seg_item_off.as<int64_t>(); b->sq_qid = ws.sq_qid.as<int>(); b->sq_out = ws.sq_out.as<int64_t>();
    return VDB_OK;
}


static int row_norms(Workspace& ws, const float* data, int ld, int dim, int64_t n, const SumPlan& plan,
                     float* sq, float* nrm, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    StreamPlanBufs b{};
    VDB_TRY(reserve_stream_plan(ws, 1, 1, (n + 255) / 256, &b));
    return launch_row_norms(data, ld, dim, n, plan, b, sq, nrm, s);
}

// queries -> padded device rows + their |q|^2, |q| in NumPy order
static int prepare_queries(Workspace& ws, const float* d_q, int nq, int dim, int ld, const SumPlan& plan,
                           const float** q_out, hipStream_t s) {
    if (ld == dim && dim % 8 == 0) {
        *q_out = d_q;
    } else {
        // the streaming kernel reads whole groups of 8 floats: keep 8 readable floats behind the last row
        VDB_TRY(ws.q_buf.reserve((size_t)nq * ld * sizeof(float) + 64));
        VDB_TRY(launch_pad_rows(d_q, dim, dim, nq, ws.q_buf.as<float>(), ld, s));
        *q_out = ws.q_buf.as<float>();
    }
    VDB_TRY(ws.q_sq.reserve((size_t)nq * sizeof(float)));
    VDB_TRY(ws.q_nrm.reserve((size_t)nq * sizeof(float)));
    VDB_TRY(row_norms(ws, *q_out, ld, dim, nq, plan, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), s));
    return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

// ============================================================== row store
struct vdb_rows : Handle {
    Rows r;
    DevBuf data, data16, sq, nrm;
    Workspace ws;
    // Exact search behind a bf16 filter (vdb_rows_set_prefilter): a bf16 copy of the fp32 rows, kept in step with
    // them lazily (rows [0, shadow_n) are converted), and the largest row norm seen (a device float)
    int prefilter = 0;
    DevBuf shadow16, max_nrm, slack;
    int64_t shadow_n = 0, shadow_cap = 0;
    bool last_used_prefilter = false;
    bool last_used_queue = false;       // diagnostics of the last flat search
    hipStream_t last_stream = nullptr;
    const int* last_flag = nullptr;     // overflow flag of that search (device)
};

static int rows_reserve(vdb_rows* h, int64_t n_total, hipStream_t s) {
    Rows& r = h->r;
    if (n_total <= r.cap) return VDB_OK;
    int64_t cap = std::max<int64_t>(n_total, r.cap + r.cap / 2);
    // a 256-row chunk that starts at any stored row stays readable: the scans read whole chunks and drop what
    // lies past the end of a list or of the store
    cap = round_up64(std::max<int64_t>(cap, 1024) + 256, 256);
    // the view is refreshed after every buffer that moved, and the capacity only once all of them have room: a
    // failure half way leaves pointers that are valid and a capacity that every buffer still covers
    int st = r.bf16 ? h->data16.reserve_keep((size_t)cap * r.ld16 * 2, (size_t)r.n * r.ld16 * 2, s)
                    : h->data.reserve_keep((size_t)cap * r.ld * sizeof(float), (size_t)r.n * r.ld * sizeof(float), s);
    r.data = h->data.as<float>();
    r.data16 = h->data16.as<uint16_t>();
    if (st == VDB_OK) st = h->sq.reserve_keep((size_t)cap * sizeof(float), (size_t)r.n * sizeof(float), s);
    r.sq = h->sq.as<float>();
    if (st == VDB_OK) st = h->nrm.reserve_keep((size_t)cap * sizeof(float), (size_t)r.n * sizeof(float), s);
    r.nrm = h->nrm.as<float>();
    if (st != VDB_OK) return st;
    // norm terms of positions nobody has written read as zero (inverted lists keep free room between their rows, and
    // the largest norm over a range of positions must not pick up what the allocator left there)
    VDB_HIP(hipMemsetAsync(r.sq + r.n, 0, (size_t)(cap - r.n) * sizeof(float), s));
    VDB_HIP(hipMemsetAsync(r.nrm + r.n, 0, (size_t)(cap - r.n) * sizeof(float), s));
    r.cap = cap;
    return VDB_OK;
}

static int rows_append_dev_impl(vdb_rows* h, const float* d_rows, int64_t n, int64_t src_ld, hipStream_t s) {
    Rows& r = h->r;
    if (n <= 0) return VDB_OK;
    h->last_stream = s;
    h->shadow_n = std::min(h->shadow_n, r.n);       // rows past a truncation are new rows
    VDB_TRY(rows_reserve(h, r.n + n, s));
    if (r.bf16) {
        // in chunks: round, store bf16, norms of the rounded values
        const int64_t chunk = std::max<int64_t>(1024, (int64_t)(256 << 20) / (r.ld * (int64_t)sizeof(float)));
        for (int64_t o = 0; o < n; o += chunk) {
            const int64_t c = std::min(chunk, n - o);
            VDB_TRY(h->ws.round32.reserve((size_t)c * r.ld * sizeof(float)));
            hipLaunchKernelGGL(round_bf16_kernel, dim3(grid_for(c * r.ld16)), dim3(256), 0, s, d_rows + o * src_ld, src_ld, r.dim,
                               c, r.data16 + (r.n + o) * r.ld16, r.ld16, h->ws.round32.as<float>(), r.ld);
            VDB_HIP(hipGetLastError());
            VDB_TRY(row_norms(h->ws, h->ws.round32.as<float>(), r.ld, r.dim, c, r.plan, r.sq + r.n + o, r.nrm + r.n + o, s));
        }
        r.n += n;
        return VDB_OK;
    }
    float* dst = r.data + r.n * r.ld;
    if (src_ld == r.ld && r.ld == r.dim) {
        VDB_HIP(hipMemcpyAsync(dst, d_rows, (size_t)n * r.ld * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(n * r.ld)), dim3(256), 0, s, d_rows, src_ld, r.dim, n,
                           (const int*)nullptr, dst, r.ld);
        VDB_HIP(hipGetLastError());
    }
    VDB_TRY(row_norms(h->ws, dst, r.ld, r.dim, n, r.plan, r.sq + r.n, r.nrm + r.n, s));
    r.n += n;
    return VDB_OK;
}


// bf16 storage: queries rounded to bf16 (what the matrix cores multiply) + norms of the rounded values
static int prepare_queries_bf16(Workspace& ws, const Rows& r, const float* d_q, int nq, const void** q_out, hipStream_t s) {
    VDB_TRY(ws.round32.reserve((size_t)nq * r.ld * sizeof(float)));
    VDB_TRY(ws.q16.reserve((size_t)nq * r.ld16 * 2));
    hipLaunchKernelGGL(round_bf16_kernel, dim3(grid_for((int64_t)nq * r.ld16)), dim3(256), 0, s, d_q, (int64_t)r.dim, r.dim,
                       (int64_t)nq, ws.q16.as<uint16_t>(), r.ld16, ws.round32.as<float>(), r.ld);
    VDB_HIP(hipGetLastError());
    VDB_TRY(ws.q_sq.reserve((size_t)nq * sizeof(float)));
    VDB_TRY(ws.q_nrm.reserve((size_t)nq * sizeof(float)));
    VDB_TRY(row_norms(ws, ws.round32.as<float>(), r.ld, r.dim, nq, r.plan, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), s));
    *q_out = ws.q16.p;
    return VDB_OK;
}

// ---- streaming scan of all rows for a block of queries: candidate buffer [nq][n] of streamed sums
static const int64_t CAND_BUDGET = (int64_t)1 << 30;      // bytes of candidate buffer per pass

// blocks per query of the candidate selection: enough to fill the chip several times over
// and to keep a thread's serial walk short (each step waits on dependent loads)
static int cand_splits(int64_t tiles, int nq) {
    static const int forced = getenv("VDB_CAND_SPLITS") ? atoi(getenv("VDB_CAND_SPLITS")) : 0;
    if (forced > 0) return (int)std::min<int64_t>(tiles, forced);
    const int64_t by_chip = (4 * (int64_t)ctx().num_cus + nq - 1) / nq;
    const int64_t by_walk = (tiles + 63) / 64;
    int64_t splits = std::max<int64_t>(1, std::min<int64_t>(tiles, std::max(by_chip, by_walk)));
    // a handful of queries: the selection is latency, not bandwidth.  At most 64 splits (256 lists:
    // one merge launch; 256 splits and two merge levels past a quarter of a million candidates) of
    // at least 16 tiles each, so a small candidate set goes through one block.
    if (nq <= 8) splits = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(splits, tiles > 1024 ? 256 : 64), tiles / 16));
    return (int)splits;
}

// leave_sqrt: direct-form Euclidean sums are stored without their square root (the selection that follows takes it)
// `more`: the fields of a StreamJob beyond the plain sums (run_if, the bounded form), or null.  own_plan: plan this
// scan in the shared plan buffers instead of the cached flat plan (a scan over a prefix of the rows must not evict the
// plan of the whole set).
static int stream_flat_sums(Workspace& ws, const Rows& r, int metric, const float* q, int nq, float* out, hipStream_t s,
                            bool leave_sqrt = false, const StreamJob* more = nullptr, bool own_plan = false) {
    StreamPlanBufs b{};
    // queries per row chunk: 16 for real batches (each staged chunk serves twice as many queries), 8 for a handful
    const int group = nq == 1 ? 1 : stream_batch_group(r.dim, r.ld, r.plan, op_of_metric(metric), nq, r.n);
    const int64_t items_cap = ((r.n + 255) / 256) * ((nq + group - 1) / group);
    // the plan buffers of a flat scan are its own (row norms and the IVF plans use the shared ones),
    // so a repeated search of the same shape skips the two planning launches
    DevBuf* bufs[] = {&ws.fp_seg_row_off, &ws.fp_seg_q_off, &ws.fp_seg_item_off, &ws.fp_sq_qid, &ws.fp_sq_out, &ws.fp_items};
    const size_t need[] = {2 * sizeof(int64_t), 2 * sizeof(int), 2 * sizeof(int64_t), (size_t)std::max(nq, 1) * sizeof(int),
                           (size_t)std::max(nq, 1) * sizeof(int64_t), (size_t)std::max<int64_t>(items_cap, 1) * sizeof(StreamItem)};
    if (own_plan) {
        VDB_TRY(reserve_stream_plan(ws, 1, nq, items_cap, &b));
        VDB_TRY(launch_flat_plan(r.n, nq, group, r.n, b, s));
    } else {
        bool moved = false;
        for (int i = 0; i < 6; ++i) {
            void* before = bufs[i]->p;
            VDB_TRY(bufs[i]->reserve(need[i]));
            moved = moved || bufs[i]->p != before;
        }
        b.seg_row_off = ws.fp_seg_row_off.as<int64_t>(); b.seg_q_off = ws.fp_seg_q_off.as<int>();
        b.seg_item_off = ws.fp_seg_item_off.as<int64_t>(); b.sq_qid = ws.fp_sq_qid.as<int>(); b.sq_out = ws.fp_sq_out.as<int64_t>();
        b.items = ws.fp_items.as<StreamItem>(); b.items_cap = items_cap;
        if (moved || ws.fp_n != r.n || ws.fp_nq != nq || ws.fp_group != group) {
            VDB_TRY(launch_flat_plan(r.n, nq, group, r.n, b, s));
            ws.fp_n = r.n; ws.fp_nq = nq; ws.fp_group = group;
        }
    }
    StreamJob j{};
    if (more) j = *more;
    j.data = r.data; j.ld = r.ld; j.dim = r.dim; j.plan = r.plan;
    j.op = op_of_metric(metric); j.post = post_of_metric(metric);
    if (leave_sqrt && j.post == POST_SQRT) j.post = POST_NONE;
    j.q = q; j.ldq = r.ld; j.group = group; j.n_seg = 1;
    j.seg_row_off = b.seg_row_off; j.seg_q_off = b.seg_q_off; j.seg_item_off = b.seg_item_off;
    j.sq_qid = b.sq_qid; j.sq_out = b.sq_out; j.items = b.items; j.items_cap = b.items_cap; j.out = out; j.timed = true;
    return launch_stream_sums(j, s);
}

static constexpr int FLAT_BOUND_QUEUE = 4096;     // (distance, row) entries per query in the bounded form

// streamed search of a flat row set (any metric): per block of queries, sums -> candidate buffer -> k best
static int stream_flat_search(vdb_rows* h, Workspace& ws, int metric, const float* q, int nq, int k, int64_t k_eff,
                              const uint32_t* d_mask, float* d_out_d, int64_t* d_out_i, int* d_out_i32, hipStream_t s) {
    Rows& r = h->r;
    // queries per pass: what the candidate buffer [queries][rows] may take; from the static scratch, half of what is left
    int64_t budget = CAND_BUDGET;
    if (ws.cand.arena) budget = std::min<int64_t>(budget, (int64_t)(ws.cand.arena->size - std::min(ws.cand.arena->size, ws.cand.arena->used)) / 2);
    const int q_block = (int)std::max<int64_t>(1, std::min<int64_t>(nq, budget / (r.n * (int64_t)sizeof(float))));
    const int64_t ntiles = (r.n + 255) / 256;
    const int kmax = (int)std::min<int64_t>(KMAX_FUSED, k_eff);
    const bool multi = k_eff > KMAX_FUSED;
    VDB_TRY(ws.cand.reserve((size_t)q_block * r.n * sizeof(float)));
    if (multi) {
        VDB_TRY(ws.lb_d.reserve((size_t)q_block * sizeof(float)));
        VDB_TRY(ws.lb_k.reserve((size_t)q_block * sizeof(int64_t)));
    }
    // Bounded form (batches over a large row set with the row-resident kernel, direct-form metrics, k results in one
    // pass): the k best of the first rows of the set go through the plain pipeline; their k-th distance is not below
    // the query's k-th best over all rows, so the scan of the whole set keeps only the (row, query) pairs within it,
    // in per-query queues, and one block per query picks the k best of its queue.  The [queries][rows] candidate
    // buffer is neither written nor read.  A queue that fills up (rows ordered so that the first ones are the
    // farthest) sets a device flag, and the plain pipeline below then runs behind that flag; otherwise its launches
    // return at once.  Either way the results are those of the plain pipeline.
    static const int64_t bound_min_rows = getenv("VDB_FLAT_BOUND_MIN_ROWS") ? atoll(getenv("VDB_FLAT_BOUND_MIN_ROWS")) : 262144;
    const int op = op_of_metric(metric);
    // (decided once for the call: a last, smaller block of queries must get the row-resident kernel too)
    const int last_nb = nq % q_block ? nq % q_block : std::min(nq, q_block);
    const bool bounded = nq >= 2 && last_nb >= 2 && !multi && r.n >= bound_min_rows &&
                         (op == OP_SQDIFF || op == OP_ABSDIFF || op == OP_MAXABS) &&
                         stream_batch_group(r.dim, r.ld, r.plan, op, std::min(nq, q_block), r.n) > 16 &&
                         stream_batch_group(r.dim, r.ld, r.plan, op, last_nb, r.n) > 16;
    const int64_t n_s = round_up64(std::max<int64_t>(r.n / 64, bound_min_rows / 16), 256);
    const int* run_if = nullptr;
    if (bounded) {
        VDB_TRY(ws.thr_d.reserve((size_t)q_block * sizeof(float)));
        VDB_TRY(ws.thr_k.reserve((size_t)q_block * sizeof(int64_t)));
        VDB_TRY(ws.queue.reserve((size_t)q_block * FLAT_BOUND_QUEUE * (sizeof(float) + sizeof(int))));
        VDB_TRY(ws.queue_cnt.reserve((size_t)q_block * sizeof(int)));
        VDB_TRY(ws.flag.reserve(sizeof(int)));
        VDB_HIP(hipMemsetAsync(ws.flag.p, 0, sizeof(int), s));
        run_if = ws.flag.as<int>();
    }
    for (int q0 = 0; q0 < nq; q0 += q_block) {
        const int nb = std::min(q_block, nq - q0);
        // enough blocks to fill the chip, at least 8 tiles each
        const int splits = cand_splits(ntiles, nb);
        // a handful of queries: the lists are merged by one block per query, which is a chain of inserts per list; the
        // selection blocks hand over one list each instead of one per wave
        const bool block_list = nb <= 8;
        const int n_cand = splits * (block_list ? 1 : topk_cand_lists_per_split());
        VDB_TRY(ws.part_d.reserve((size_t)nb * n_cand * kmax * sizeof(float)));
        VDB_TRY(ws.part_i.reserve((size_t)nb * n_cand * kmax * sizeof(int)));
        if (bounded) {
            // 1: the k-th best of the first n_s rows -> ws.thr_d
            Rows rs = r;
            rs.n = n_s;
            const int64_t s_tiles = (n_s + 255) / 256;
            const int s_splits = cand_splits(s_tiles, nb);
            const int s_cand = s_splits * topk_cand_lists_per_split();
            VDB_TRY(ws.part_d.reserve((size_t)nb * s_cand * kmax * sizeof(float)));
            VDB_TRY(ws.part_i.reserve((size_t)nb * s_cand * kmax * sizeof(int)));
            VDB_TRY(stream_flat_sums(ws, rs, metric, q + (int64_t)q0 * r.ld, nb, ws.cand.as<float>(), s, true, nullptr, true));
            CandJob c{};
            c.sqrt_here = true;
            c.cand = ws.cand.as<float>(); c.cand_stride = n_s; c.nq = nb; c.n = n_s; c.mask = d_mask;
            c.metric = metric; c.q_sq = ws.q_sq.as<float>() + q0; c.q_nrm = ws.q_nrm.as<float>() + q0;
            c.row_sq = r.sq; c.row_nrm = r.nrm;
            c.k = kmax; c.splits = s_splits; c.part_d = ws.part_d.as<float>(); c.part_i = ws.part_i.as<int>();
            VDB_TRY(launch_topk_cand(c, s));
            SelectOut ob{nullptr, nullptr, nullptr, kmax, 0, ws.thr_d.as<float>(), ws.thr_k.as<int64_t>()};
            VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nb, s_cand, kmax, 1 << 30, nullptr, 0, ob, s, ws.mid));
            // 2: all rows, pairs within the bound -> queues
            VDB_HIP(hipMemsetAsync(ws.queue_cnt.p, 0, (size_t)nb * sizeof(int), s));
            StreamJob more{};
            more.bound = ws.thr_d.as<float>(); more.bound_sqrt = metric == VDB_EUCLIDEAN_DIRECT; more.mask = d_mask;
            more.queue_len = ws.queue_cnt.as<int>(); more.queue_d = ws.queue.as<float>();
            more.queue_i = reinterpret_cast<int*>(ws.queue.as<float>() + (size_t)q_block * FLAT_BOUND_QUEUE);
            more.queue_cap = FLAT_BOUND_QUEUE; more.overflow = ws.flag.as<int>();
            VDB_TRY(stream_flat_sums(ws, r, metric, q + (int64_t)q0 * r.ld, nb, nullptr, s, true, &more));
            // 3: the k best of every queue
            SelectOut o{d_out_d ? d_out_d + (int64_t)q0 * k : nullptr, d_out_i ? d_out_i + (int64_t)q0 * k : nullptr,
                        d_out_i32 ? d_out_i32 + (int64_t)q0 * k : nullptr, k, 0, nullptr, nullptr};
            VDB_TRY(launch_select_packed(more.queue_d, more.queue_i, nb, FLAT_BOUND_QUEUE, kmax, o, s, more.queue_len));
        }
        StreamJob guard{};
        guard.run_if = run_if;
        VDB_TRY(stream_flat_sums(ws, r, metric, q + (int64_t)q0 * r.ld, nb, ws.cand.as<float>(), s, true, bounded ? &guard : nullptr));
        for (int64_t done = 0; done < k_eff; done += KMAX_FUSED) {
            const int kk = (int)std::min<int64_t>(KMAX_FUSED, k_eff - done);
            CandJob c{};
            c.run_if = run_if;
            c.sqrt_here = true;
            c.cand = ws.cand.as<float>(); c.cand_stride = r.n; c.nq = nb; c.n = r.n; c.mask = d_mask;
            c.metric = metric; c.q_sq = ws.q_sq.as<float>() + q0; c.q_nrm = ws.q_nrm.as<float>() + q0;
            c.row_sq = r.sq; c.row_nrm = r.nrm;
            c.lb_d = done > 0 ? ws.lb_d.as<float>() : nullptr; c.lb_k = done > 0 ? ws.lb_k.as<int64_t>() : nullptr;
            c.k = kk; c.splits = splits; c.part_d = ws.part_d.as<float>(); c.part_i = ws.part_i.as<int>();
            c.block_list = block_list;
            VDB_TRY(launch_topk_cand(c, s));
            SelectOut o{d_out_d ? d_out_d + (int64_t)q0 * k : nullptr, d_out_i ? d_out_i + (int64_t)q0 * k : nullptr,
                        d_out_i32 ? d_out_i32 + (int64_t)q0 * k : nullptr, k, (int)done,
                        multi ? ws.lb_d.as<float>() : nullptr, multi ? ws.lb_k.as<int64_t>() : nullptr, run_if};
            VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nb, n_cand, kk, 1 << 30, nullptr, 0, o, s, ws.mid));
        }
    }
    return VDB_OK;
}

// bf16 copy of the fp32 rows for the filter pass: rows [shadow_n, n) are converted now
static int shadow_pitch(const Rows& r) { return round_up(r.dim, 64); }     // whole 128-byte column steps for any dimension

static int shadow_sync(vdb_rows* h, hipStream_t s) {
    Rows& r = h->r;
    const int ld16 = shadow_pitch(r);
    h->shadow_n = std::min(h->shadow_n, r.n);
    if (!h->max_nrm.p) {
        VDB_TRY(h->max_nrm.reserve(sizeof(float)));
        VDB_HIP(hipMemsetAsync(h->max_nrm.p, 0, sizeof(float), s));
        h->shadow_n = 0;
    }
    if (h->shadow_cap < r.cap) {
        if (h->prefilter == VDB_PREFILTER_AUTO) {
            // the copy is taken only while it leaves a tenth of the device (at least 2 GiB) free for everything else
            size_t free_b = 0, total_b = 0;
            VDB_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t want = (size_t)r.cap * ld16 * 2 + std::max<size_t>(total_b / 10, (size_t)2 << 30);
            if (free_b + h->shadow16.cap < want) return VDB_ERR_NOMEM;
        }
        VDB_TRY(h->shadow16.reserve_keep((size_t)r.cap * ld16 * 2, (size_t)h->shadow_n * ld16 * 2, s));
        h->shadow_cap = r.cap;
    }
    if (h->shadow_n < r.n) {
        const int64_t c = r.n - h->shadow_n;
        hipLaunchKernelGGL(round_bf16_kernel, dim3(grid_for(c * ld16)), dim3(256), 0, s, r.data + h->shadow_n * r.ld, (int64_t)r.ld,
                           r.dim, c, h->shadow16.as<uint16_t>() + h->shadow_n * ld16, ld16, (float*)nullptr, 0);
        VDB_HIP(hipGetLastError());
        // the largest norm over all rows, taken afresh (4 bytes per row): an overwritten outlier stops loosening the slack
        VDB_HIP(hipMemsetAsync(h->max_nrm.p, 0, sizeof(float), s));
        VDB_TRY(launch_max_norm(r.nrm, 0, r.n, h->max_nrm.as<float>(), s));
        h->shadow_n = r.n;
    }
    return VDB_OK;
}

// Nearest row of every query of a large batch (k = 1, Euclidean forms, fp32 rows, no filter): k-means assignment and
// index builds.  The branch-free kernel keeps a running (smallest, second smallest, row) per lane instead of lists
// (launch_flat_argmin); the few queries whose two best keys are too close for the clamp and square root of the list
// kernel to tell apart are searched again by the list kernel itself, so every result is the list kernel's.
static int flat_nearest_row(vdb_rows* h, Workspace& ws, int metric, const float* q, int nq, float* d_out_d, int64_t* d_out_i,
                            int* d_out_i32, hipStream_t s) {
    Rows& r = h->r;
    const int lists = flat_argmin_lists(r, nq);
    const int lim = std::min(nq, std::max(128, nq / 64));
    int n_cand = 0, n_cand_redo = 0;
    VDB_TRY(flat_gemm_num_cand(r.n, nq, 1, &n_cand));
    VDB_TRY(flat_gemm_num_cand(r.n, lim, 1, &n_cand_redo));
    const size_t parts = std::max((size_t)nq * n_cand, (size_t)lim * n_cand_redo);
    const size_t d_bytes = ((size_t)lim * sizeof(float) + 7) / 8 * 8;
    VDB_TRY(ws.mins.reserve((size_t)nq * lists * sizeof(float4)));
    VDB_TRY(ws.redo_list.reserve(((size_t)lim + 4) * sizeof(int)));
    VDB_TRY(ws.redo_q.reserve((size_t)lim * r.ld * sizeof(float)));
    VDB_TRY(ws.redo_terms.reserve((size_t)lim * 2 * sizeof(float)));
    VDB_TRY(ws.redo_out.reserve(d_bytes + (size_t)lim * (sizeof(int64_t) + sizeof(int))));
    VDB_TRY(ws.part_d.reserve(parts * sizeof(float)));
    VDB_TRY(ws.part_i.reserve(parts * sizeof(int)));
    int* list = ws.redo_list.as<int>();
    int* wanted = list + lim;
    int* q_count = wanted + 1;
    int* all_flag = wanted + 2;
    float* t_sq = ws.redo_terms.as<float>();
    float* t_nrm = t_sq + lim;
    float* c_d = ws.redo_out.as<float>();
    int64_t* c_i = reinterpret_cast<int64_t*>(ws.redo_out.as<char>() + d_bytes);
    int* c_i32 = reinterpret_cast<int*>(c_i + lim);
    VDB_HIP(hipMemsetAsync(wanted, 0, 3 * sizeof(int), s));
    VDB_TRY(launch_flat_argmin(r, metric, q, ws.q_sq.as<float>(), nq, ws.mins.p, d_out_d, d_out_i, d_out_i32, 1, list, wanted, lim, s));
    VDB_TRY(launch_redo_gather_list(list, wanted, lim, q, r.ld, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), ws.redo_q.as<float>(),
                                    t_sq, t_nrm, q_count, all_flag, s));
    VDB_TRY(launch_flat_gemm_topk(r, metric, ws.redo_q.as<float>(), t_sq, t_nrm, lim, 1, nullptr, nullptr, nullptr, nullptr, 0, 1,
                                  ws.part_d.as<float>(), ws.part_i.as<int>(), n_cand_redo, s, nullptr, q_count));
    {
        SelectOut o{c_d, d_out_i ? c_i : nullptr, d_out_i32 ? c_i32 : nullptr, 1, 0, nullptr, nullptr, nullptr};
        VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), lim, n_cand_redo, 1, 1 << 30, nullptr, 0, o, s, ws.mid));
    }
    VDB_TRY(launch_redo_scatter(list, q_count, lim, 1, c_d, c_i, c_i32, d_out_d, d_out_i, d_out_i32, s));
    // more unsure queries than the list holds (duplicated rows, say): the whole batch through the list kernel
    VDB_TRY(launch_flat_gemm_topk(r, metric, q, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), nq, 1, nullptr, nullptr, nullptr, nullptr,
                                  0, 1, ws.part_d.as<float>(), ws.part_i.as<int>(), n_cand, s, all_flag));
    SelectOut o{d_out_d, d_out_i, d_out_i32, 1, 0, nullptr, nullptr, all_flag};
    return launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, n_cand, 1, 1 << 30, nullptr, 0, o, s, ws.mid);
}

// ---- flat search on device pointers ------------------------------------
// `ws_in` replaces the handle's own workspace (the no-allocation entry point passes one bound to the static scratch)
static int flat_search_impl(vdb_rows* h, int metric, const float* d_q, int nq, int k, const uint32_t* d_mask,
                            float* d_out_d, int64_t* d_out_i, int* d_out_i32, hipStream_t s, Workspace* ws_in = nullptr) {
    Rows& r = h->r;
    Workspace& ws = ws_in ? *ws_in : h->ws;
    h->last_used_queue = false;
    h->last_used_prefilter = false;
    h->last_stream = s;
    h->last_flag = nullptr;
    VDB_REQUIRE(valid_metric(metric), "unknown metric code %d", metric);
    VDB_REQUIRE(nq >= 0 && k >= 1, "need nq >= 0 and k >= 1 (got nq=%d, k=%d)", nq, k);
    if (nq == 0) return VDB_OK;
    const int64_t k_eff = std::min<int64_t>(k, r.n);
    if (k_eff < k) {
        VDB_REQUIRE(d_out_i32 == nullptr || r.n > 0, "empty row set");
        VDB_TRY(fill_results(d_out_d, d_out_i, (int64_t)nq * k, s));
    }
    if (r.n == 0) return VDB_OK;

    const void* q = nullptr;
    if (r.bf16) {
        VDB_REQUIRE(is_dot_metric(metric), "bf16 storage serves euclidean, euclidean_squared, cosine and dot (metric %d asked)", metric);
        VDB_TRY(prepare_queries_bf16(ws, r, d_q, nq, &q, s));
    } else {
        const float* qf = nullptr;
        VDB_TRY(prepare_queries(ws, d_q, nq, r.dim, r.ld, r.plan, &qf, s));
        const bool gemm = is_dot_metric(metric) && nq >= GEMM_MIN_NQ;
        if (!gemm) return stream_flat_search(h, ws, metric, qf, nq, k, k_eff, d_mask, d_out_d, d_out_i, d_out_i32, s);
        q = qf;
        const bool no_argmin = getenv("VDB_NO_ARGMIN") != nullptr;           // read per call: tests compare both paths
        if (k == 1 && !d_mask && (metric == VDB_EUCLIDEAN || metric == VDB_EUCLIDEAN_SQ) && nq >= 1024 && r.n >= 2 &&
            flat_queue_fast_ok(r) && ws_in == nullptr && !no_argmin)
            return flat_nearest_row(h, ws, metric, qf, nq, d_out_d, d_out_i, d_out_i32, s);
    }

    int n_cand = 0, sample_tiles = 0, sample_stride = 1, n_cand_sample = 0;
    VDB_TRY(flat_gemm_num_cand(r.n, nq, k, &n_cand));
    VDB_TRY(flat_gemm_sample_plan(r.n, &sample_tiles, &sample_stride));
    if (h->prefilter != VDB_PREFILTER_NONE && !r.bf16 && ws_in == nullptr && sample_tiles > 0) {
        // Behind the bf16 filter every survivor of the bound costs more (a looser test, an fp32 recomputation for some)
        // and the bound pass a quarter of the fp32 one: a denser sample pays while rows are short.  Cosine 1M x 384,
        // batch 1024: 1.01 ms at 1/64 of the tiles, 0.96 at 1/32, 0.93 at 1/16, 0.96 at 1/8; euclidean 1M x 128: 0.71,
        // 0.64, 0.61, 0.59; dot 1.25M x 768: 2.10 at all of 1/64 ... 1/16, and 10M x 768 14.2 ms at 1/64 against 14.8.
        static const int forced = getenv("VDB_PREFILTER_SAMPLE") ? std::max(1, atoi(getenv("VDB_PREFILTER_SAMPLE"))) : 0;
        const int pitch = shadow_pitch(r);
        const int dens = forced ? forced : (pitch <= 384 ? 4 : (pitch <= 512 ? 2 : 1));
        const int stride = std::max(1, sample_stride / dens);
        const int n_tiles128 = (int)((r.n + 127) / 128);
        sample_tiles = std::max(sample_tiles, n_tiles128 / stride);
        sample_stride = std::max(1, n_tiles128 / sample_tiles);
    }
    if (sample_tiles > 0) VDB_TRY(flat_gemm_num_cand_tiles(sample_tiles, nq, &n_cand_sample));
    const int kmax = (int)std::min<int64_t>(KMAX_FUSED, k_eff);
    const int cand_max = std::max(n_cand, n_cand_sample);
    VDB_TRY(ws.part_d.reserve((size_t)nq * cand_max * kmax * sizeof(float)));
    VDB_TRY(ws.part_i.reserve((size_t)nq * cand_max * kmax * sizeof(int)));
    const bool multi = k_eff > KMAX_FUSED;
    if (multi) {
        VDB_TRY(ws.lb_d.reserve((size_t)nq * sizeof(float)));
        VDB_TRY(ws.lb_k.reserve((size_t)nq * sizeof(int64_t)));
    }
    const float* thr_init = nullptr;
    static const bool no_queue = getenv("VDB_NO_QUEUE") != nullptr;
    const bool will_queue = sample_tiles > 0 && !multi && !no_queue;
    bool flag_cleared = false;
    // Queries whose queues overflow are marked one by one and searched again alone by the list kernel (fp32 rows):
    // one unlucky query does not cost the batch a second search.  bf16 storage keeps the whole-batch repeat.
    const bool redo_some = will_queue && !r.bf16;
    if (will_queue) VDB_TRY(ws.flag.reserve(sizeof(int)));
    if (redo_some) VDB_TRY(ws.redo_mark.reserve((size_t)nq * sizeof(int)));
    int* marks = redo_some ? ws.redo_mark.as<int>() : nullptr;
    // Filter pass on bf16 copies, exact selection in fp32 (vdb_rows_set_prefilter): only where the 256 x 256 bf16
    // kernel can read the copy, and never from the entry point that must not allocate (the copy is allocated here)
    Rows rf = r;
    rf.bf16 = true;
    rf.ld16 = shadow_pitch(r);
    bool prefilter = will_queue && h->prefilter != VDB_PREFILTER_NONE && !r.bf16 && ws_in == nullptr && queue_use_wide(rf);
    bool thr_from_copies = false;
    if (prefilter) {
        // no room for the copy: the search runs on the fp32 kernels alone, as it would without the filter
        const int st = shadow_sync(h, s);
        if (st == VDB_ERR_NOMEM) {
            (void)hipGetLastError();
            h->shadow16.release(); h->shadow_n = 0; h->shadow_cap = 0;
            prefilter = false;
        } else {
            VDB_TRY(st);
        }
    }
    if (prefilter) {
        rf.data16 = h->shadow16.as<uint16_t>();
        VDB_TRY(ws.q16.reserve((size_t)nq * rf.ld16 * 2));
        hipLaunchKernelGGL(round_bf16_kernel, dim3(grid_for((int64_t)nq * rf.ld16)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(q), (int64_t)r.ld, r.dim, (int64_t)nq, ws.q16.as<uint16_t>(), rf.ld16,
                           (float*)nullptr, 0);
        VDB_HIP(hipGetLastError());
        VDB_TRY(h->slack.reserve((size_t)nq * sizeof(float)));
    }
    h->last_used_prefilter = prefilter;
    if (sample_tiles > 0) {
        // k-th best distance over a strided sample of the rows: an upper bound of the
        // k-th best over all rows, used as the initial filter threshold of the full pass
        VDB_TRY(ws.thr_d.reserve((size_t)nq * sizeof(float)));
        VDB_TRY(ws.thr_k.reserve((size_t)nq * sizeof(int64_t)));
        static const bool no_min = getenv("VDB_NO_MINPASS") != nullptr;
        // behind the bf16 filter the bound pass reads the copies too; the bound it finds is widened below by what
        // rounding can have moved the k distances it stands for
        static const bool f32_bound = getenv("VDB_PREFILTER_F32_BOUND") != nullptr;
        const bool bound_on_copies = prefilter && !f32_bound && flat_queue_fast_ok(rf);
        const Rows& rb = bound_on_copies ? rf : r;
        if (flat_queue_fast_ok(rb) && !no_min) {
            // per-lane minima over the sample, then the k-th smallest of them (blocks of queries keep
            // the number of lanes per query well above k)
            const char* qb = bound_on_copies ? ws.q16.as<char>() : reinterpret_cast<const char*>(q);
            const size_t q_pitch = rb.bf16 ? (size_t)rb.ld16 * 2 : (size_t)r.ld * 4;
            thr_from_copies = bound_on_copies;
            for (int q0 = 0; q0 < nq; q0 += QUEUE_Q_BLOCK) {
                const int nb = std::min(QUEUE_Q_BLOCK, nq - q0);
                int ns = 0;
                const int n_lists = flat_min_sample_lists(sample_tiles, nb, kmax, &ns);
                VDB_TRY(ws.mins.reserve((size_t)nb * n_lists * sizeof(float)));
                VDB_TRY(launch_flat_min_sample(rb, metric, qb + (size_t)q0 * q_pitch, ws.q_sq.as<float>() + q0,
                                               ws.q_nrm.as<float>() + q0, nb, kmax, d_mask, sample_tiles, sample_stride,
                                               ws.mins.as<float>(), ws.thr_d.as<float>() + q0,
                                               will_queue ? ws.flag.as<int>() : nullptr, s, marks ? marks + q0 : nullptr));
            }
            flag_cleared = will_queue;
        } else {
            VDB_TRY(launch_flat_gemm_topk(r, metric, q, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), nq, kmax, d_mask,
                                          nullptr, nullptr, nullptr, sample_tiles, sample_stride,
                                          ws.part_d.as<float>(), ws.part_i.as<int>(), n_cand_sample, s));
            SelectOut so{nullptr, nullptr, nullptr, kmax, 0, ws.thr_d.as<float>(), ws.thr_k.as<int64_t>()};
            VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, n_cand_sample, kmax, 1 << 30, nullptr, 0, so, s, ws.mid));
        }
        thr_init = ws.thr_d.as<float>();
    }

    // With a bound in hand and a single pass (k <= KMAX_FUSED) the survivors are queued and selected
    // directly; the list kernel below then runs only if a queue overflowed (a device-side flag, no
    // host round trip).
    const int* run_if = nullptr;
    if (will_queue) {
        h->last_used_queue = true;
        h->last_flag = ws.flag.as<int>();
        if (!flag_cleared) {
            VDB_HIP(hipMemsetAsync(ws.flag.p, 0, sizeof(int), s));
            if (marks) VDB_HIP(hipMemsetAsync(marks, 0, (size_t)nq * sizeof(int), s));
        }
        if (prefilter)
            VDB_TRY(launch_prefilter_slack(metric, ws.q_nrm.as<float>(), h->max_nrm.as<float>(), nq, r.dim, h->slack.as<float>(),
                                           thr_from_copies ? ws.thr_d.as<float>() : nullptr, ws.flag.as<int>(), s, marks));
        const Rows& rq = prefilter ? rf : r;
        const char* qb = prefilter ? ws.q16.as<char>() : reinterpret_cast<const char*>(q);
        const size_t q_pitch = rq.bf16 ? (size_t)rq.ld16 * 2 : (size_t)r.ld * 4;
        for (int q0 = 0; q0 < nq; q0 += QUEUE_Q_BLOCK) {
            const int nb = std::min(QUEUE_Q_BLOCK, nq - q0);
            int n_slabs = 0, cap = 0;
            VDB_TRY(flat_queue_plan(rq, nb, kmax, sample_stride, &n_slabs, &cap, prefilter));
            VDB_TRY(ws.queue.reserve((size_t)nb * n_slabs * cap * 8));
            VDB_TRY(ws.queue_cnt.reserve((size_t)nb * n_slabs * sizeof(int)));
            PrefilterArgs pf{r.data, r.ld, reinterpret_cast<const float*>(q) + (size_t)q0 * r.ld, r.ld, h->slack.as<float>() + q0};
            VDB_TRY(launch_flat_queue(rq, metric, qb + (size_t)q0 * q_pitch, ws.q_sq.as<float>() + q0, ws.q_nrm.as<float>() + q0,
                                      nb, kmax, d_mask, thr_init + q0, n_slabs, cap, ws.queue.p, ws.queue_cnt.as<int>(),
                                      ws.flag.as<int>(), d_out_d ? d_out_d + (int64_t)q0 * k : nullptr,
                                      d_out_i ? d_out_i + (int64_t)q0 * k : nullptr,
                                      d_out_i32 ? d_out_i32 + (int64_t)q0 * k : nullptr, s, prefilter ? &pf : nullptr,
                                      marks ? marks + q0 : nullptr));
        }
        run_if = ws.flag.as<int>();
    }
    if (redo_some) {
        // Compact the marked queries (their padded rows, norm terms and fp32 bounds), search those with the list
        // kernel on a grid laid out for one query tile (so that a handful of queries still fills the GPU), select into
        // compact results and move them to their queries' places.  More marked queries than a tile holds: the whole
        // batch again, as before (the loop below, behind the second flag).  Everything is launched either way and
        // returns at once while its flag or count is zero.
        const float* qf = reinterpret_cast<const float*>(q);
        const int lim = std::min(nq, flat_gemm_query_tile());
        int n_cand_redo = 0;
        VDB_TRY(flat_gemm_num_cand(r.n, lim, k, &n_cand_redo));
        const size_t d_bytes = ((size_t)lim * k * sizeof(float) + 7) / 8 * 8;
        VDB_TRY(ws.redo_list.reserve(((size_t)nq + 2) * sizeof(int)));
        VDB_TRY(ws.redo_q.reserve((size_t)lim * r.ld * sizeof(float)));
        VDB_TRY(ws.redo_terms.reserve((size_t)lim * 3 * sizeof(float)));
        VDB_TRY(ws.redo_out.reserve(d_bytes + (size_t)lim * k * (sizeof(int64_t) + sizeof(int))));
        VDB_TRY(ws.part_d.reserve((size_t)lim * n_cand_redo * kmax * sizeof(float)));
        VDB_TRY(ws.part_i.reserve((size_t)lim * n_cand_redo * kmax * sizeof(int)));
        int* list = ws.redo_list.as<int>();
        int* count = list + nq;
        int* all_flag = count + 1;
        float* t_sq = ws.redo_terms.as<float>();
        float* t_nrm = t_sq + lim;
        float* t_thr = t_nrm + lim;
        float* c_d = ws.redo_out.as<float>();
        int64_t* c_i = reinterpret_cast<int64_t*>(ws.redo_out.as<char>() + d_bytes);
        int* c_i32 = reinterpret_cast<int*>(c_i + (size_t)lim * k);
        // (a bound found on the bf16 copies was widened for them; the list kernel takes none then)
        const float* thr_f32 = (thr_init && !thr_from_copies) ? thr_init : nullptr;
        VDB_TRY(launch_redo_gather(run_if, marks, nq, qf, r.ld, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), thr_f32, list, count,
                                   ws.redo_q.as<float>(), t_sq, t_nrm, t_thr, lim, all_flag, s));
        VDB_TRY(launch_flat_gemm_topk(r, metric, ws.redo_q.as<float>(), t_sq, t_nrm, lim, kmax, d_mask, nullptr, nullptr,
                                      thr_f32 ? t_thr : nullptr, 0, 1, ws.part_d.as<float>(), ws.part_i.as<int>(), n_cand_redo, s,
                                      run_if, count));
        SelectOut o{c_d, d_out_i ? c_i : nullptr, d_out_i32 ? c_i32 : nullptr, k, 0, nullptr, nullptr, run_if};
        VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), lim, n_cand_redo, kmax, 1 << 30, nullptr, 0, o, s, ws.mid));
        VDB_TRY(launch_redo_scatter(list, count, lim, k, c_d, c_i, c_i32, d_out_d, d_out_i, d_out_i32, s));
        run_if = all_flag;
    }

    for (int64_t done = 0; done < k_eff; done += KMAX_FUSED) {
        const int kk = (int)std::min<int64_t>(KMAX_FUSED, k_eff - done);
        const float* lb_d = (done > 0) ? ws.lb_d.as<float>() : nullptr;
        const int64_t* lb_k = (done > 0) ? ws.lb_k.as<int64_t>() : nullptr;
        VDB_TRY(launch_flat_gemm_topk(r, metric, q, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), nq, kk, d_mask,
                                      lb_d, lb_k, done == 0 ? thr_init : nullptr, 0, 1,
                                      ws.part_d.as<float>(), ws.part_i.as<int>(), n_cand, s, run_if));
        SelectOut o{d_out_d, d_out_i, d_out_i32, k, (int)done,
                    multi ? ws.lb_d.as<float>() : nullptr, multi ? ws.lb_k.as<int64_t>() : nullptr, run_if};
        VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, n_cand, kk, 1 << 30, nullptr, 0, o, s, ws.mid));
    }
    return VDB_OK;
}

static int flat_distances_impl(vdb_rows* h, int metric, const float* d_q, int nq, float* d_out, hipStream_t s,
                               bool back_to_front = false) {
    Rows& r = h->r;
    Workspace& ws = h->ws;
    VDB_REQUIRE(valid_metric(metric), "unknown metric code %d", metric);
    if (nq <= 0 || r.n == 0) return VDB_OK;
    if (r.bf16) {
        VDB_REQUIRE(is_dot_metric(metric), "bf16 storage serves euclidean, euclidean_squared, cosine and dot (metric %d asked)", metric);
        const void* q16 = nullptr;
        VDB_TRY(prepare_queries_bf16(ws, r, d_q, nq, &q16, s));
        return launch_flat_gemm_dist(r, metric, q16, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), nq, d_out, s);
    }
    const float* q = nullptr;
    VDB_TRY(prepare_queries(ws, d_q, nq, r.dim, r.ld, r.plan, &q, s));
    if (is_dot_metric(metric) && nq >= GEMM_MIN_NQ)
        return launch_flat_gemm_dist(r, metric, q, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), nq, d_out, s);
    StreamJob more{};
    more.timed = true;
    if (back_to_front && nq == 1) more.reverse_items = (r.n + 255) / 256;      // one query: one item per 256-row chunk
    VDB_TRY(stream_flat_sums(ws, r, metric, q, nq, d_out, s, false, &more));
    return launch_finish_dist(metric, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), r.sq, r.nrm, r.cos_eps, d_out, nq, r.n, s);
}

// copy a host bitmask of n bits to the workspace
static int upload_mask(Workspace& ws, const uint32_t* mask, int64_t n, const uint32_t** d_mask, hipStream_t s) {
    *d_mask = nullptr;
    if (!mask) return VDB_OK;
    const size_t words = (size_t)((n + 31) / 32);
    VDB_TRY(ws.mask.reserve(words * sizeof(uint32_t) + 16));
    VDB_HIP(hipMemcpyAsync(ws.mask.p, mask, words * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    *d_mask = ws.mask.as<uint32_t>();
    return VDB_OK;
}

// A filter bitmask that stays in HBM: evaluated and uploaded once, named by every search that uses it.
struct vdb_mask {
    DevBuf words;
    int64_t n_bits = 0;
};

// the mask of a host-pointer search: a resident one as it is, a host one uploaded to the workspace
static int resolve_mask(Workspace& ws, const uint32_t* host_mask, const vdb_mask* resident, int64_t n_rows,
                        const uint32_t** d_mask, hipStream_t s) {
    if (!resident) return upload_mask(ws, host_mask, n_rows, d_mask, s);
    if (resident->n_bits < n_rows) {
        set_error("the resident mask covers %lld rows, the index holds %lld", (long long)resident->n_bits, (long long)n_rows);
        return VDB_ERR_INVALID;
    }
    *d_mask = resident->words.as<uint32_t>();
    return VDB_OK;
}

// host-pointer wrapper: queries up, results down.  Small transfers (a query or a batch of them, k results
// each) go through page-locked buffers: one asynchronous copy up, one down (distances and rows lie in one
// device block), one wait.  Copies straight to pageable memory cost a staging round trip each.
static const size_t PINNED_IO_MAX = 4u << 20;

template <class F>
static int with_host_io(Workspace& ws, const float* queries, int nq, int dim, int k, float* out_dist, int64_t* out_row,
                        hipStream_t s, F&& run) {
    if (nq <= 0) return VDB_OK;
    const size_t q_bytes = (size_t)nq * dim * sizeof(float);
    const size_t d_bytes = (size_t)nq * k * sizeof(float), d_pad = (d_bytes + 7) / 8 * 8;
    const size_t i_bytes = (size_t)nq * k * sizeof(int64_t);
    VDB_TRY(ws.stage.reserve(q_bytes));
    VDB_TRY(ws.out_d.reserve(d_pad + i_bytes));          // distances, then rows
    float* dd = ws.out_d.as<float>();
    int64_t* di = reinterpret_cast<int64_t*>(ws.out_d.as<char>() + d_pad);
    const bool pinned = q_bytes <= PINNED_IO_MAX && d_pad + i_bytes <= PINNED_IO_MAX;
    if (pinned) {
        VDB_TRY(ws.pin_in.reserve(q_bytes));
        VDB_TRY(ws.pin_out.reserve(d_pad + i_bytes));
        std::memcpy(ws.pin_in.p, queries, q_bytes);
        VDB_HIP(hipMemcpyAsync(ws.stage.p, ws.pin_in.p, q_bytes, hipMemcpyHostToDevice, s));
    } else {
        VDB_HIP(hipMemcpyAsync(ws.stage.p, queries, q_bytes, hipMemcpyHostToDevice, s));
    }
    VDB_TRY(run(ws.stage.as<float>(), dd, di));
    if (pinned) {
        if (out_dist || out_row) VDB_HIP(hipMemcpyAsync(ws.pin_out.p, ws.out_d.p, d_pad + i_bytes, hipMemcpyDeviceToHost, s));
        VDB_HIP(hipStreamSynchronize(s));
        if (out_dist) std::memcpy(out_dist, ws.pin_out.p, d_bytes);
        if (out_row) std::memcpy(out_row, ws.pin_out.as<char>() + d_pad, i_bytes);
        return VDB_OK;
    }
    if (out_dist) VDB_HIP(hipMemcpyAsync(out_dist, dd, d_bytes, hipMemcpyDeviceToHost, s));
    if (out_row) VDB_HIP(hipMemcpyAsync(out_row, di, i_bytes, hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

// ============================================================== where inverted lists live
// After a build the lists are packed one behind the other in list order.  A list that outgrows its place is moved to
// the end of the position space with twice the room; the place it leaves is dead until the next build.  Appending a
// row therefore touches one list (amortised: one row), never the whole index.
struct ListMove { int64_t from, to, count; };
struct ListLayout {
    int n_lists = 0;
    std::vector<int64_t> beg, len, cap;
    int64_t tail = 0;           // positions handed out: rows, free room and dead places
    int64_t live = 0;           // rows in lists
    int64_t dead = 0;           // positions in abandoned places
    int64_t row_limit = 0;      // 1 + the largest row number stored: the length of a filter mask
    DevBuf dev;                 // int64 [2 * n_lists]: where every list begins, then where it ends
    std::vector<int64_t> staged;
    bool dev_stale = true;
    std::vector<int64_t> len_desc;      // list lengths at the last sort, longest first
    int64_t grown_since_sort = 0;

    void reset(int n) {
        n_lists = n;
        beg.assign(n, 0); len.assign(n, 0); cap.assign(n, 0);
        tail = live = dead = row_limit = 0;
        dev_stale = true;
        len_desc.clear();
        grown_since_sort = 0;
    }
    void set_packed(const int64_t* offsets) {
        for (int l = 0; l < n_lists; ++l) { beg[l] = offsets[l]; len[l] = cap[l] = offsets[l + 1] - offsets[l]; }
        tail = live = offsets[n_lists];
        dead = 0;
        dev_stale = true;
        len_desc.clear();
        grown_since_sort = 0;
    }
    const int64_t* d_beg() const { return dev.as<int64_t>(); }
    const int64_t* d_end() const { return dev.as<int64_t>() + n_lists; }
    int sync(hipStream_t s) {
        if (!dev_stale) return VDB_OK;
        staged.resize((size_t)2 * n_lists);
        for (int l = 0; l < n_lists; ++l) { staged[l] = beg[l]; staged[(size_t)n_lists + l] = beg[l] + len[l]; }
        VDB_TRY(dev.reserve(staged.size() * sizeof(int64_t)));
        VDB_HIP(hipMemcpyAsync(dev.p, staged.data(), staged.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
        VDB_HIP(hipStreamSynchronize(s));      // `staged` is pageable and is rewritten by the next call
        dev_stale = false;
        return VDB_OK;
    }
    // an upper bound of the rows in the n_probe longest lists (the most candidates one query can have): exact at
    // the last sort, plus everything appended since
    int64_t probe_bound(int n_probe) {
        if (len_desc.size() != (size_t)n_lists || grown_since_sort > std::max<int64_t>(4096, live / 16)) {
            len_desc.assign(len.begin(), len.end());
            std::sort(len_desc.begin(), len_desc.end(), std::greater<int64_t>());
            grown_since_sort = 0;
        }
        int64_t t = grown_since_sort;
        for (int i = 0; i < std::min(n_probe, n_lists); ++i) t += len_desc[i];
        return t;
    }
    // what plan_append changed, so that a caller whose allocations fail afterwards can take the plan back: the lists
    // then describe exactly the rows that are stored (a moved list's old place is never overwritten by the plan)
    struct Undo {
        struct Entry { int list; int64_t beg, len, cap; };
        std::vector<Entry> lists;
        int64_t tail = 0, live = 0, dead = 0, row_limit = 0, grown = 0;
        bool dev_stale = false;
    };
    void rollback(const Undo& u) {
        for (const auto& e : u.lists) { beg[e.list] = e.beg; len[e.list] = e.len; cap[e.list] = e.cap; }
        tail = u.tail; live = u.live; dead = u.dead; row_limit = u.row_limit; grown_since_sort = u.grown;
        dev_stale = true;       // the device copy may have been refreshed from the plan
    }
    // positions for `count` new rows (dst[i] for the row that goes to lists[i]) and the lists that have to move first
    int plan_append(const int32_t* lists, int64_t count, std::vector<int64_t>& dst, std::vector<ListMove>& moves, Undo& undo) {
        std::unordered_map<int, int64_t> need;
        for (int64_t i = 0; i < count; ++i) {
            VDB_REQUIRE(lists[i] >= 0 && lists[i] < n_lists, "list number %d out of range", lists[i]);
            ++need[lists[i]];
        }
        undo.tail = tail; undo.live = live; undo.dead = dead; undo.row_limit = row_limit; undo.grown = grown_since_sort;
        undo.dev_stale = dev_stale;
        undo.lists.clear();
        for (const auto& kv : need) undo.lists.push_back(Undo::Entry{kv.first, beg[kv.first], len[kv.first], cap[kv.first]});
        for (const auto& kv : need) {
            const int l = kv.first;
            if (len[l] + kv.second <= cap[l]) continue;
            const int64_t room = std::max<int64_t>(16, 2 * (len[l] + kv.second));
            if (len[l] > 0) moves.push_back(ListMove{beg[l], tail, len[l]});
            dead += cap[l];
            beg[l] = tail; cap[l] = room;
            tail += room;
        }
        dst.resize((size_t)count);
        for (int64_t i = 0; i < count; ++i) dst[(size_t)i] = beg[lists[i]] + len[lists[i]]++;
        live += count;
        grown_since_sort += count;
        dev_stale = true;
        return VDB_OK;
    }
};

// ============================================================== IVF handle
struct vdb_ivf : Handle {
    int dim = 0, metric = 0;
    vdb_rows* centroids = nullptr;      // n_lists rows
    vdb_rows* members = nullptr;        // the rows of every list, at the positions `lay` names
    ListLayout lay;
    DevBuf row_ids;                     // int   [positions]  row number of every stored position
    DevBuf app_rows, app_sq, app_nrm, app_dst, app_no;     // staging of appended rows
    int n_lists = 0;
    // bf16 copy of the member rows for the matrix-core scan of batched Euclidean searches (vdb_ivf_set_prefilter):
    // same positions as the fp32 rows, brought up to date lazily from the position ranges writes have touched
    int prefilter = VDB_PREFILTER_AUTO;
    DevBuf shadow16, max_nrm;
    DevBuf cent16, cent_max_nrm;                           // the same of the centroids (row-major), for ranking the lists
    bool cent16_stale = true;
    int64_t shadow_cap = 0;                                // positions the copy has room for
    bool shadow_stale = true;                              // everything must be converted (no copy yet, or a rebuild)
    std::vector<std::pair<int64_t, int64_t>> shadow_dirty; // [from, to) position ranges written since the last search
    void shadow_note(int64_t from, int64_t to) {
        if (shadow_stale || to <= from) return;
        if (shadow_dirty.size() >= 32) {                   // many small writes: one range over all of them
            for (const auto& r : shadow_dirty) { from = std::min(from, r.first); to = std::max(to, r.second); }
            shadow_dirty.clear();
        }
        shadow_dirty.emplace_back(from, to);
    }
    bool last_used_mfma = false;
    int64_t probe_bound(int n_probe) { return lay.probe_bound(n_probe); }
    int64_t last_scan_rows = -1;
    int last_nq = 0, last_nprobe = 0;
    hipStream_t last_stream = nullptr;  // stream of the last search (its probe lists are read back for the statistics)
    Workspace ws;
};

// ============================================================== PQ handle
struct vdb_pq : Handle {
    int dim = 0, m = 0, ksub = 0, dsub = 0;
    bool trained = false;
    SumPlan plan;                       // over one sub-vector
    DevBuf book;                        // float [m][ksub][dsub]
    DevBuf codes;                       // uint8 [cap][m]
    int64_t n = 0, cap = 0;
    hipStream_t last_stream = nullptr;  // stream of the last *_dev call
    bool last_used_bytes = false;       // the last search went through the byte-table pass
    int* last_flag = nullptr;           // its overflow flag (device)
    Workspace ws;
    Workspace io;                       // staging of the host-pointer entry points
};

// ---- static scratch of the no-allocation entry point ---------------------------------------
// Device memory the library holds from load to exit on the GPU it runs on.  In host code the name
// `scratch` is only a stand-in for the symbol: the device address is taken once, in vdb_init, with
// hipGetSymbolAddress, and only that pointer is ever handed to kernels and copies.
static __device__ __align__(256) unsigned char scratch[VDB_STATIC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;        // one call at a time enqueues work that uses the array

static int resolve_scratch(void** out) {
    void* p = nullptr;
    VDB_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    VDB_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address for the static scratch");
    *out = p;
    return VDB_OK;
}

static int scan_metric_of(int metric) {
    // the per-pair scalar functions behind get_metric_fn (distance/registry.py:95-177)
    switch (metric) {
        case VDB_EUCLIDEAN: return VDB_EUCLIDEAN_DIRECT;
        case VDB_EUCLIDEAN_SQ: return VDB_EUCLIDEAN_SQ_DIRECT;
        case VDB_COSINE: return VDB_COSINE_PAIR;
        default: return metric;
    }
}

extern "C" {

// ============================================================== library
int vdb_init(int device) {
    Context& c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (c.device >= 0) {
        // one process, one GPU: every handle, stream and buffer lives on the device of the first call
        if (c.device != device) {
            set_error("the library is already initialised on device %d; device %d asked (one process per GPU)", c.device, device);
            return VDB_ERR_STATE;
        }
        VDB_HIP(hipSetDevice(device));
        return VDB_OK;
    }
    int count = 0;
    VDB_HIP(hipGetDeviceCount(&count));
    VDB_REQUIRE(device >= 0 && device < count, "device %d out of range (%d visible)", device, count);
    VDB_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VDB_HIP(hipGetDeviceProperties(&prop, device));
    if (!c.zeros) {
        VDB_HIP(hipMalloc(reinterpret_cast<void**>(&c.zeros), 256));
        VDB_HIP(hipMemset(c.zeros, 0, 256));
    }
    c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // device address of the static scratch on THIS device (the device is current here); the library
    // never runs on a second device in one process, so one pointer is all there is to keep
    VDB_TRY(resolve_scratch(&c.scratch));
    if (!c.scratch_done) VDB_HIP(hipEventCreateWithFlags(&c.scratch_done, hipEventDisableTiming));
    c.device = device;
    return VDB_OK;
}

const char* vdb_last_error(void) { return g_err.c_str(); }

int vdb_device_info(char* name, int name_len, int64_t* mem_total, int64_t* mem_free) {
    VDB_TRY(ensure_init());
    hipDeviceProp_t prop;
    VDB_HIP(hipGetDeviceProperties(&prop, ctx().device));
    if (name && name_len > 0) { strncpy(name, prop.name, name_len - 1); name[name_len - 1] = 0; }
    size_t fr = 0, tot = 0;
    VDB_HIP(hipMemGetInfo(&fr, &tot));
    if (mem_total) *mem_total = (int64_t)tot;
    if (mem_free) *mem_free = (int64_t)fr;
    return VDB_OK;
}

int vdb_measure_ceilings(float* copy_gbs, float* mfma_f32_tflops, float* mfma_bf16_tflops, float* valu_direct_tops) {
    VDB_REQUIRE(copy_gbs && mfma_f32_tflops && mfma_bf16_tflops && valu_direct_tops, "null argument");
    ScopedHandle tmp;
    VDB_TRY(tmp.open());
    return measure_ceilings(tmp.stream, copy_gbs, mfma_f32_tflops, mfma_bf16_tflops, valu_direct_tops);
}

int vdb_profile_enable(int on) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_profile = on != 0;
    return VDB_OK;
}

int vdb_profile_read(float* kernel_ms, int* launches) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    float total = 0.0f;
    int n = 0;
    for (auto& e : g_prof_events) {
        float ms = 0.0f;
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
            total += ms;
            ++n;
        }
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    g_prof_events.clear();
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = n;
    return VDB_OK;
}

int vdb_synchronize(void) {
    VDB_TRY(ensure_init());
    VDB_HIP(hipSetDevice(ctx().device));
    VDB_HIP(hipDeviceSynchronize());        // every handle has its own stream
    return VDB_OK;
}

// ============================================================== rows
int vdb_rows_create(vdb_rows** out, int dim) { return vdb_rows_create_typed(out, dim, VDB_STORE_F32); }

int vdb_rows_create_typed(vdb_rows** out, int dim, int storage) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(out != nullptr, "null output handle");
    VDB_REQUIRE(dim >= 1 && dim <= 8192, "dimension must be in 1..8192, got %d", dim);
    VDB_REQUIRE(storage == VDB_STORE_F32 || storage == VDB_STORE_BF16, "unknown storage type %d", storage);
    vdb_rows* h = new vdb_rows();
    h->r.dim = dim;
    h->r.ld = round_up(dim, 4);
    h->r.bf16 = storage == VDB_STORE_BF16;
    h->r.ld16 = round_up(dim, 8);
    static const bool prefilter_default = getenv("VDB_PREFILTER") != nullptr && atoi(getenv("VDB_PREFILTER")) != 0;
    if (prefilter_default && !h->r.bf16) h->prefilter = VDB_PREFILTER_BF16;
    int st = make_sum_plan(dim, &h->r.plan);
    if (st == VDB_OK) st = h->open();
    if (st != VDB_OK) { delete h; return st; }
    *out = h;
    return VDB_OK;
}

int vdb_rows_destroy(vdb_rows* h) {
    if (!h) return VDB_OK;
    {
        Entry g(h);
        // work on this handle may still be queued on its own stream or, from a *_dev call, on the caller's
        (void)hipStreamSynchronize(h->stream);
        if (h->last_stream && h->last_stream != h->stream) (void)hipStreamSynchronize(h->last_stream);
        h->data.release(); h->data16.release(); h->sq.release(); h->nrm.release(); h->ws.release();
        h->shadow16.release(); h->max_nrm.release(); h->slack.release();
    }
    h->close();
    delete h;
    return VDB_OK;
}

int vdb_rows_append(vdb_rows* h, const float* rows, int64_t n) {
    VDB_REQUIRE(h && (rows || n == 0) && n >= 0, "bad arguments to vdb_rows_append");
    if (n == 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = h->stream;
    Rows& r = h->r;
    h->shadow_n = std::min(h->shadow_n, r.n);
    VDB_TRY(rows_reserve(h, r.n + n, s));
    if (r.ld == r.dim && !r.bf16) {
        float* dst = r.data + r.n * r.ld;
        VDB_HIP(hipMemcpyAsync(dst, rows, (size_t)n * r.dim * sizeof(float), hipMemcpyHostToDevice, s));
        VDB_TRY(row_norms(h->ws, dst, r.ld, r.dim, n, r.plan, r.sq + r.n, r.nrm + r.n, s));
        r.n += n;
    } else {
        // stage in chunks, pad on the device
        const int64_t chunk = std::max<int64_t>(1, (int64_t)(64 << 20) / (r.dim * (int64_t)sizeof(float)));
        for (int64_t o = 0; o < n; o += chunk) {
            const int64_t c = std::min(chunk, n - o);
            VDB_TRY(h->ws.stage.reserve((size_t)c * r.dim * sizeof(float)));
            VDB_HIP(hipMemcpyAsync(h->ws.stage.p, rows + o * r.dim, (size_t)c * r.dim * sizeof(float), hipMemcpyHostToDevice, s));
            VDB_TRY(rows_append_dev_impl(h, h->ws.stage.as<float>(), c, r.dim, s));
            VDB_HIP(hipStreamSynchronize(s));
        }
    }
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_rows_append_dev(vdb_rows* h, const float* d_rows, int64_t n, int64_t ld, void* stream) {
    VDB_REQUIRE(h && (d_rows || n == 0) && n >= 0 && ld >= h->r.dim, "bad arguments to vdb_rows_append_dev");
    Entry g(h);
    return rows_append_dev_impl(h, d_rows, n, ld, pick_stream(stream));
}

int vdb_rows_set(vdb_rows* h, int64_t row, const float* vec) {
    VDB_REQUIRE(h && vec && row >= 0 && row < h->r.n, "row %lld out of range", (long long)row);
    Entry g(h);
    hipStream_t s = h->stream;
    Rows& r = h->r;
    if (r.bf16) {
        VDB_TRY(h->ws.stage.reserve((size_t)r.dim * sizeof(float)));
        VDB_TRY(h->ws.round32.reserve((size_t)r.ld * sizeof(float)));
        VDB_HIP(hipMemcpyAsync(h->ws.stage.p, vec, (size_t)r.dim * sizeof(float), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(round_bf16_kernel, dim3(1), dim3(256), 0, s, h->ws.stage.as<float>(), (int64_t)r.dim, r.dim, (int64_t)1,
                           r.data16 + row * r.ld16, r.ld16, h->ws.round32.as<float>(), r.ld);
        VDB_HIP(hipGetLastError());
        VDB_TRY(row_norms(h->ws, h->ws.round32.as<float>(), r.ld, r.dim, 1, r.plan, r.sq + row, r.nrm + row, s));
        VDB_HIP(hipStreamSynchronize(s));
        return VDB_OK;
    }
    float* dst = r.data + row * r.ld;
    h->shadow_n = std::min(h->shadow_n, row);       // the bf16 copy is rebuilt from this row on
    VDB_HIP(hipMemcpyAsync(dst, vec, (size_t)r.dim * sizeof(float), hipMemcpyHostToDevice, s));   // padding stays zero
    VDB_TRY(row_norms(h->ws, dst, r.ld, r.dim, 1, r.plan, r.sq + row, r.nrm + row, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_rows_truncate(vdb_rows* h, int64_t n) {
    VDB_REQUIRE(h && n >= 0 && n <= h->r.n, "cannot truncate %lld rows to %lld", h ? (long long)h->r.n : 0LL, (long long)n);
    Entry g(h);
    h->r.n = n;
    h->shadow_n = std::min(h->shadow_n, n);
    return VDB_OK;
}

int vdb_rows_set_prefilter(vdb_rows* h, int mode) {
    VDB_REQUIRE(h && mode >= VDB_PREFILTER_NONE && mode <= VDB_PREFILTER_AUTO, "unknown prefilter mode %d", mode);
    Entry g(h);
    VDB_REQUIRE(mode == VDB_PREFILTER_NONE || !h->r.bf16, "a bf16 filter pass needs fp32 storage");
    h->prefilter = mode;
    if (mode == VDB_PREFILTER_NONE) {
        (void)hipStreamSynchronize(h->stream);
        if (h->last_stream && h->last_stream != h->stream) (void)hipStreamSynchronize(h->last_stream);
        h->shadow16.release(); h->shadow_n = 0; h->shadow_cap = 0;
    }
    return VDB_OK;
}

int vdb_rows_size(const vdb_rows* h, int64_t* n) {
    VDB_REQUIRE(h && n, "null argument");
    *n = h->r.n;
    return VDB_OK;
}

int vdb_rows_set_cosine_eps(vdb_rows* h, float eps) {
    VDB_REQUIRE(h && eps >= 0.0f && eps == eps, "cosine eps must be a non-negative number");
    Entry g(h);
    h->r.cos_eps = eps;
    return VDB_OK;
}

int vdb_rows_norms(vdb_rows* h, float* sqnorm_out, float* norm_out) {
    VDB_REQUIRE(h, "null handle");
    Entry g(h);
    hipStream_t s = h->stream;
    if (h->r.n == 0) return VDB_OK;
    if (sqnorm_out) VDB_HIP(hipMemcpyAsync(sqnorm_out, h->r.sq, (size_t)h->r.n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (norm_out) VDB_HIP(hipMemcpyAsync(norm_out, h->r.nrm, (size_t)h->r.n * sizeof(float), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

// ============================================================== flat
int vdb_flat_search_dev(vdb_rows* h, int metric, const float* d_queries, int nq, int k, const uint32_t* d_mask,
                        float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0), "null argument");
    Entry g(h);
    return flat_search_impl(h, metric, d_queries, nq, k, d_mask, d_out_dist, d_out_row, nullptr, pick_stream(stream));
}

static int flat_search_host(vdb_rows* h, int metric, const float* queries, int nq, int k, const uint32_t* mask,
                            const vdb_mask* resident, float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(h && (queries || nq == 0), "null argument");
    VDB_REQUIRE(k >= 1, "k must be >= 1, got %d", k);
    Entry g(h);
    hipStream_t s = h->stream;
    const uint32_t* d_mask = nullptr;
    VDB_TRY(resolve_mask(h->ws, mask, resident, h->r.n, &d_mask, s));
    return with_host_io(h->ws, queries, nq, h->r.dim, k, out_dist, out_row, s,
                        [&](const float* dq, float* dd, int64_t* di) {
                            return flat_search_impl(h, metric, dq, nq, k, d_mask, dd, di, nullptr, s);
                        });
}

int vdb_flat_search(vdb_rows* h, int metric, const float* queries, int nq, int k, const uint32_t* mask,
                    float* out_dist, int64_t* out_row) {
    return flat_search_host(h, metric, queries, nq, k, mask, nullptr, out_dist, out_row);
}

int vdb_flat_search_masked(vdb_rows* h, int metric, const float* queries, int nq, int k, const vdb_mask* mask,
                           float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(mask, "null mask");
    return flat_search_host(h, metric, queries, nq, k, nullptr, mask, out_dist, out_row);
}

// ---- resident filter masks
int vdb_mask_create(vdb_mask** out, const uint32_t* words, int64_t n_bits) {
    VDB_REQUIRE(out && (words || n_bits == 0) && n_bits >= 0, "bad argument");
    VDB_TRY(ensure_init());
    (void)hipSetDevice(ctx().device);
    auto* m = new (std::nothrow) vdb_mask();
    if (!m) { set_error("out of host memory"); return VDB_ERR_NOMEM; }
    const size_t bytes = (size_t)((n_bits + 31) / 32) * sizeof(uint32_t);
    // 16 bytes of slack: the scans read whole words past the last row
    int rc = m->words.reserve(bytes + 16);
    if (rc == VDB_OK && hipMemset(m->words.p, 0, bytes + 16) != hipSuccess) rc = VDB_ERR_HIP;
    if (rc == VDB_OK && bytes && hipMemcpy(m->words.p, words, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = VDB_ERR_HIP;
    if (rc != VDB_OK) {
        if (rc == VDB_ERR_HIP) set_error("uploading a mask of %lld bits failed", (long long)n_bits);
        m->words.release();
        delete m;
        return rc;
    }
    m->n_bits = n_bits;
    *out = m;
    return VDB_OK;
}

int vdb_mask_destroy(vdb_mask* m) {
    if (!m) return VDB_OK;
    (void)hipSetDevice(ctx().device);
    (void)hipDeviceSynchronize();          // a search enqueued through a *_dev entry point may still read it
    m->words.release();
    delete m;
    return VDB_OK;
}

int vdb_mask_info(const vdb_mask* m, int64_t* n_bits, const uint32_t** d_words) {
    VDB_REQUIRE(m, "null mask");
    if (n_bits) *n_bits = m->n_bits;
    if (d_words) *d_words = m->words.as<uint32_t>();
    return VDB_OK;
}

int vdb_flat_search_static_dev(vdb_rows* h, int metric, const float* d_queries, int nq, int k, const uint32_t* d_mask,
                               float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0), "null argument");
    Entry g(h);
    // every workspace of this call is carved out of the static array: no hipMalloc, no hipFree.  The
    // array is shared by all calls on this GPU; the lock orders the host side, the caller orders streams.
    std::lock_guard<std::mutex> sg(g_scratch_mu);
    VDB_REQUIRE(ctx().scratch != nullptr, "the static scratch was not resolved at initialisation");
    Arena arena;
    arena.base = reinterpret_cast<char*>(ctx().scratch);
    arena.size = VDB_STATIC_SCRATCH_BYTES;
    Workspace ws;
    ws.bind(&arena);
    hipStream_t s = pick_stream(stream);
    // Calls from different streams or handles share the array: each waits on the device for the one before it (an
    // event made once at initialisation; nothing is allocated here).  A stream that is being captured into a graph
    // is left alone: the order of the graph's own nodes is the caller's.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool ordered = cap == hipStreamCaptureStatusNone && ctx().scratch_done != nullptr;
    if (ordered && ctx().scratch_used) VDB_HIP(hipStreamWaitEvent(s, ctx().scratch_done, 0));
    const int st = flat_search_impl(h, metric, d_queries, nq, k, d_mask, d_out_dist, d_out_row, nullptr, s, &ws);
    if (ordered) {
        VDB_HIP(hipEventRecord(ctx().scratch_done, s));
        ctx().scratch_used = true;
    }
    // the overflow flag of this call lies in the shared array, where the next call of any handle writes: it is not
    // kept (vdb_flat_search_stats reports no overflow for such a call; the results are complete either way)
    h->last_flag = nullptr;
    return st;
}

int vdb_flat_search_stats(vdb_rows* h, int* used_queue, int* overflowed) {
    VDB_REQUIRE(h && used_queue && overflowed, "null argument");
    Entry g(h);
    *used_queue = h->last_used_queue ? (h->last_used_prefilter ? 2 : 1) : 0;
    *overflowed = 0;
    if (h->last_used_queue && h->last_flag) {
        hipStream_t s = h->last_stream ? h->last_stream : h->stream;
        VDB_HIP(hipMemcpyAsync(overflowed, h->last_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        VDB_HIP(hipStreamSynchronize(s));
    }
    return VDB_OK;
}

int vdb_flat_distances_dev(vdb_rows* h, int metric, const float* d_queries, int nq, float* d_out, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0), "null argument");
    Entry g(h);
    return flat_distances_impl(h, metric, d_queries, nq, d_out, pick_stream(stream));
}

int vdb_flat_distances(vdb_rows* h, int metric, const float* queries, int nq, float* out) {
    VDB_REQUIRE(h && (queries || nq == 0) && (out || nq == 0), "null argument");
    if (nq <= 0 || h->r.n == 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = h->stream;
    Workspace& ws = h->ws;
    VDB_TRY(ws.stage.reserve((size_t)nq * h->r.dim * sizeof(float)));
    VDB_TRY(ws.dist.reserve((size_t)nq * h->r.n * sizeof(float)));
    VDB_HIP(hipMemcpyAsync(ws.stage.p, queries, (size_t)nq * h->r.dim * sizeof(float), hipMemcpyHostToDevice, s));
    VDB_TRY(flat_distances_impl(h, metric, ws.stage.as<float>(), nq, ws.dist.as<float>(), s));
    VDB_HIP(hipMemcpyAsync(out, ws.dist.p, (size_t)nq * h->r.n * sizeof(float), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_flat_range_search(vdb_rows* h, int metric, const float* query, float radius, const uint32_t* mask,
                          int64_t cap, float* out_dist, int64_t* out_row, int64_t* count) {
    VDB_REQUIRE(h && query && count, "null argument");
    VDB_REQUIRE(valid_metric(metric), "unknown metric code %d", metric);
    Entry g(h);
    hipStream_t s = h->stream;
    Rows& r = h->r;
    Workspace& ws = h->ws;
    *count = 0;
    VDB_REQUIRE(!r.bf16, "range search needs fp32 storage");
    if (r.n == 0 || cap <= 0) return VDB_OK;
    const uint32_t* d_mask = nullptr;
    VDB_TRY(upload_mask(ws, mask, r.n, &d_mask, s));
    VDB_TRY(ws.stage.reserve((size_t)r.dim * sizeof(float)));
    VDB_HIP(hipMemcpyAsync(ws.stage.p, query, (size_t)r.dim * sizeof(float), hipMemcpyHostToDevice, s));
    const float* q = nullptr;
    VDB_TRY(prepare_queries(ws, ws.stage.as<float>(), 1, r.dim, r.ld, r.plan, &q, s));
    // the rows are streamed once into a buffer of sums; everything after reads that buffer
    VDB_TRY(ws.cand.reserve((size_t)r.n * sizeof(float)));
    VDB_TRY(stream_flat_sums(ws, r, metric, q, 1, ws.cand.as<float>(), s));
    RangeJob j{ws.cand.as<float>(), r.n, d_mask, metric, ws.q_sq.as<float>(), ws.q_nrm.as<float>(), r.sq, r.nrm, radius};
    const int64_t nb = range_blocks(r.n);
    VDB_TRY(ws.list_cnt.reserve((size_t)nb * sizeof(int)));
    VDB_TRY(ws.cand_off.reserve((size_t)(nb + 1) * sizeof(int64_t)));
    int64_t* d_total = ws.cand_off.as<int64_t>() + nb;
    VDB_TRY(launch_range_count(j, ws.list_cnt.as<int>(), ws.cand_off.as<int64_t>(), d_total, s));
    int64_t total = 0;
    VDB_HIP(hipMemcpyAsync(&total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    const int64_t m = std::min(total, cap);
    if (m <= 0) return VDB_OK;
    VDB_TRY(ws.queue.reserve((size_t)range_sort_capacity(m) * sizeof(uint64_t)));
    VDB_TRY(ws.out_d.reserve((size_t)m * sizeof(float)));
    VDB_TRY(ws.out_i.reserve((size_t)m * sizeof(int64_t)));
    VDB_TRY(launch_range_collect(j, ws.cand_off.as<int64_t>(), cap, m, ws.queue.as<uint64_t>(), ws.out_d.as<float>(),
                                 ws.out_i.as<int64_t>(), s));
    if (out_dist) VDB_HIP(hipMemcpyAsync(out_dist, ws.out_d.p, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, s));
    if (out_row) VDB_HIP(hipMemcpyAsync(out_row, ws.out_i.p, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    *count = m;
    return VDB_OK;
}

// ============================================================== assignment / k-means
int vdb_assign(const float* x, int64_t n, const float* centroids, int n_centroids, int dim, int metric,
               int32_t* out_idx, float* out_dist) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(x && centroids && n >= 0 && n_centroids >= 1 && dim >= 1, "bad arguments to vdb_assign");
    VDB_REQUIRE(valid_metric(metric), "unknown metric code %d", metric);
    if (n == 0) return VDB_OK;
    vdb_rows* c = nullptr;
    VDB_TRY(vdb_rows_create(&c, dim));
    int st = vdb_rows_append(c, centroids, n_centroids);
    if (st == VDB_OK) {
        Entry g(c);
        hipStream_t s = c->stream;
        Workspace& ws = c->ws;
        const int64_t chunk = std::max<int64_t>(1024, std::min<int64_t>(n, (int64_t)(256 << 20) / (dim * (int64_t)sizeof(float))));
        DevBuf i32;
        for (int64_t o = 0; o < n && st == VDB_OK; o += chunk) {
            const int nq = (int)std::min(chunk, n - o);
            auto body = [&]() -> int {
                VDB_TRY(ws.stage.reserve((size_t)nq * dim * sizeof(float)));
                VDB_TRY(ws.out_d.reserve((size_t)nq * sizeof(float)));
                VDB_TRY(i32.reserve((size_t)nq * sizeof(int)));
                VDB_HIP(hipMemcpyAsync(ws.stage.p, x + o * dim, (size_t)nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
                VDB_TRY(flat_search_impl(c, metric, ws.stage.as<float>(), nq, 1, nullptr, ws.out_d.as<float>(), nullptr,
                                         i32.as<int>(), s));
                if (out_idx) VDB_HIP(hipMemcpyAsync(out_idx + o, i32.p, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, s));
                if (out_dist) VDB_HIP(hipMemcpyAsync(out_dist + o, ws.out_d.p, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost, s));
                VDB_HIP(hipStreamSynchronize(s));
                return VDB_OK;
            };
            st = body();
        }
        i32.release();
    }
    std::string keep = g_err;
    vdb_rows_destroy(c);
    if (st != VDB_OK) g_err = keep;
    return st;
}

int vdb_kmeans_accumulate(const float* x, int64_t n, int dim, const int32_t* assign, int n_clusters,
                          double* out_sums, int64_t* out_counts) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(x && assign && out_sums && out_counts && n >= 0 && dim >= 1 && n_clusters >= 1,
                "bad arguments to vdb_kmeans_accumulate");
    ScopedHandle tmp;
    VDB_TRY(tmp.open());
    hipStream_t s = tmp.stream;
    DevBuf sums, counts, xs, as;
    auto body = [&]() -> int {
        VDB_TRY(sums.reserve((size_t)n_clusters * dim * sizeof(double)));
        VDB_TRY(counts.reserve((size_t)n_clusters * sizeof(unsigned long long)));
        VDB_HIP(hipMemsetAsync(sums.p, 0, (size_t)n_clusters * dim * sizeof(double), s));
        VDB_HIP(hipMemsetAsync(counts.p, 0, (size_t)n_clusters * sizeof(unsigned long long), s));
        const int64_t chunk = std::max<int64_t>(1024, (int64_t)(256 << 20) / (dim * (int64_t)sizeof(float)));
        for (int64_t o = 0; o < n; o += chunk) {
            const int64_t c = std::min(chunk, n - o);
            VDB_TRY(xs.reserve((size_t)c * dim * sizeof(float)));
            VDB_TRY(as.reserve((size_t)c * sizeof(int)));
            VDB_HIP(hipMemcpyAsync(xs.p, x + o * dim, (size_t)c * dim * sizeof(float), hipMemcpyHostToDevice, s));
            VDB_HIP(hipMemcpyAsync(as.p, assign + o, (size_t)c * sizeof(int), hipMemcpyHostToDevice, s));
            VDB_TRY(launch_kmeans_accumulate(xs.as<float>(), c, dim, dim, as.as<int>(), n_clusters, sums.as<double>(),
                                             counts.as<unsigned long long>(), s));
            VDB_HIP(hipStreamSynchronize(s));
        }
        VDB_HIP(hipMemcpyAsync(out_sums, sums.p, (size_t)n_clusters * dim * sizeof(double), hipMemcpyDeviceToHost, s));
        VDB_HIP(hipMemcpyAsync(out_counts, counts.p, (size_t)n_clusters * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        VDB_HIP(hipStreamSynchronize(s));
        return VDB_OK;
    };
    const int st = body();
    sums.release(); counts.release(); xs.release(); as.release();
    return st;
}

// ============================================================== k-means on resident rows
}  // extern "C"

struct vdb_kmeans : Handle {
    vdb_rows* train = nullptr;          // borrowed: the training rows (fp32 storage)
    vdb_rows* sub = nullptr;            // own dense copy of the trained columns when they are a slice
    vdb_rows* cent = nullptr;           // current centroids as a row store (the matrix-core assignment searches it)
    int col0 = 0, dsub = 0, k = 0;
    bool f64 = false;                   // the centroids are float64 (PQ form after its first step)
    bool have_next = false;
    int last_form = 0;
    DevBuf cur32, next32, cur64, next64, assign, dist, nearest, hist, totals, cluster_off, members, counts, bsums,
           chosen, qrow, csq64, scal, rows_in, dst_in, dense;
    DevBuf x16, x_max_nrm, q16f;        // k-means++ rounds over a bf16 copy of the rows (launch_km_pp_bf16)
    const Rows& x() const { return (sub ? sub : train)->r; }
    Workspace& xws() { return (sub ? sub : train)->ws; }
};

static int kmeans_refresh_centroid_store(vdb_kmeans* h, hipStream_t s) {
    h->cent->r.n = 0;
    return rows_append_dev_impl(h->cent, h->cur32.as<float>(), h->k, h->dsub, s);
}

// nearest centroid of every training row under the fp32 forms: metric = VDB_EUCLIDEAN (KMeans._assign,
// ivf.py:228-231) or VDB_EUCLIDEAN_SQ (the first PQ step, pq.py:156-157)
static int kmeans_assign_f32(vdb_kmeans* h, int metric, hipStream_t s) {
    const Rows& x = h->x();
    VDB_TRY(kmeans_refresh_centroid_store(h, s));
    VDB_TRY(h->assign.reserve((size_t)x.n * sizeof(int)));
    VDB_TRY(h->dist.reserve((size_t)x.n * sizeof(float)));
    const int64_t chunk = 262144;
    const bool dense = x.ld == h->dsub;
    if (!dense) VDB_TRY(h->dense.reserve((size_t)std::min<int64_t>(chunk, x.n) * h->dsub * sizeof(float)));
    for (int64_t o = 0; o < x.n; o += chunk) {
        const int nq = (int)std::min<int64_t>(chunk, x.n - o);
        const float* q = x.data + o * x.ld;
        if (!dense) {
            VDB_TRY(launch_pad_rows(q, x.ld, h->dsub, nq, h->dense.as<float>(), h->dsub, s));
            q = h->dense.as<float>();
        }
        VDB_TRY(flat_search_impl(h->cent, metric, q, nq, 1, nullptr, h->dist.as<float>() + o, nullptr, h->assign.as<int>() + o, s));
    }
    return VDB_OK;
}

extern "C" {

int vdb_kmeans_create(vdb_kmeans** out, vdb_rows* train, int col0, int dsub, int n_clusters) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(out && train, "null argument");
    VDB_REQUIRE(!train->r.bf16, "training rows must be stored in fp32");
    VDB_REQUIRE(col0 >= 0 && dsub >= 1 && col0 + dsub <= train->r.dim, "columns [%d, %d) are outside the %d-dimensional rows",
                col0, col0 + dsub, train->r.dim);
    VDB_REQUIRE(n_clusters >= 1 && n_clusters <= 32768, "n_clusters must be in 1..32768, got %d", n_clusters);
    VDB_REQUIRE(train->r.n >= n_clusters, "%lld training rows for %d clusters", (long long)train->r.n, n_clusters);
    vdb_kmeans* h = new vdb_kmeans();
    h->train = train; h->col0 = col0; h->dsub = dsub; h->k = n_clusters;
    int st = h->open();
    if (st == VDB_OK) st = vdb_rows_create(&h->cent, dsub);
    if (st == VDB_OK && (col0 != 0 || dsub != train->r.dim)) {
        st = vdb_rows_create(&h->sub, dsub);
        if (st == VDB_OK) {
            Entry g(train);
            st = rows_append_dev_impl(h->sub, train->r.data + col0, train->r.n, train->r.ld, h->stream);
            if (st == VDB_OK && hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream synchronisation failed"); st = VDB_ERR_HIP; }
        }
    }
    if (st == VDB_OK) st = h->cur32.reserve((size_t)n_clusters * dsub * sizeof(float));
    if (st == VDB_OK) st = h->next32.reserve((size_t)n_clusters * dsub * sizeof(float));
    if (st == VDB_OK) st = h->cur64.reserve((size_t)n_clusters * dsub * sizeof(double));
    if (st == VDB_OK) st = h->next64.reserve((size_t)n_clusters * dsub * sizeof(double));
    if (st != VDB_OK) { std::string keep = g_err; vdb_kmeans_destroy(h); g_err = keep; return st; }
    *out = h;
    return VDB_OK;
}

int vdb_kmeans_destroy(vdb_kmeans* h) {
    if (!h) return VDB_OK;
    {
        Entry g(h);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        DevBuf* all[] = {&h->cur32, &h->next32, &h->cur64, &h->next64, &h->assign, &h->dist, &h->nearest, &h->hist, &h->totals,
                         &h->cluster_off, &h->members, &h->counts, &h->bsums, &h->chosen, &h->qrow, &h->csq64, &h->scal,
                         &h->rows_in, &h->dst_in, &h->dense};
        for (DevBuf* b : all) b->release();
    }
    vdb_rows_destroy(h->sub);
    vdb_rows_destroy(h->cent);
    h->close();
    delete h;
    return VDB_OK;
}

int vdb_kmeans_init_rows(vdb_kmeans* h, const int64_t* rows, int n_rows) {
    VDB_REQUIRE(h && rows && n_rows == h->k, "vdb_kmeans_init_rows needs one training row per cluster (%d given)", n_rows);
    for (int i = 0; i < n_rows; ++i) VDB_REQUIRE(rows[i] >= 0 && rows[i] < h->x().n, "row %lld out of range", (long long)rows[i]);
    Entry g(h);
    hipStream_t s = h->stream;
    VDB_TRY(h->rows_in.reserve((size_t)n_rows * sizeof(int64_t)));
    VDB_HIP(hipMemcpyAsync(h->rows_in.p, rows, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice, s));
    VDB_TRY(launch_km_gather(h->x().data, h->x().ld, h->dsub, h->rows_in.as<long long>(), nullptr, n_rows, h->cur32.as<float>(),
                             h->cur64.as<double>(), s));
    VDB_HIP(hipStreamSynchronize(s));
    h->f64 = false; h->have_next = false;
    return VDB_OK;
}

int vdb_kmeans_init_pp(vdb_kmeans* h, int64_t first_row, const double* uniforms, int64_t* out_rows) {
    VDB_REQUIRE(h && (uniforms || h->k == 1), "null argument");
    VDB_REQUIRE(first_row >= 0 && first_row < h->x().n, "row %lld out of range", (long long)first_row);
    Entry g(h);
    Entry gx(h->sub ? h->sub : h->train);        // the scan below uses the row store's own workspace
    hipStream_t s = h->stream;
    const Rows& x = h->x();
    vdb_rows* xs = h->sub ? h->sub : h->train;
    const int k = h->k;
    VDB_TRY(h->chosen.reserve(((size_t)k + 1) * sizeof(int64_t)));        // the chosen rows, then one error word
    VDB_TRY(h->dist.reserve((size_t)x.n * sizeof(float)));
    VDB_TRY(h->nearest.reserve((size_t)x.n * sizeof(float)));
    VDB_TRY(h->bsums.reserve((size_t)km_pp_blocks(x.n) * sizeof(double)));
    VDB_TRY(h->qrow.reserve((size_t)h->dsub * sizeof(float) + 64));
    long long* chosen = h->chosen.as<long long>();
    VDB_HIP(hipMemcpyAsync(chosen, &first_row, sizeof(int64_t), hipMemcpyHostToDevice, s));
    VDB_HIP(hipMemsetAsync(chosen + k, 0, sizeof(int64_t), s));
    // rows that can serve as queries as they lie: the chosen row's stored norms are the query's, the streamed dot
    // products are finished by the update kernel (four launches per round instead of six)
    static const bool no_short = getenv("VDB_KMPP_LONG") != nullptr;
    const bool short_round = !no_short && !x.bf16 && x.ld == x.dim && x.dim % 8 == 0 && x.dim == h->dsub;
    if (short_round) {
        VDB_TRY(xs->ws.q_sq.reserve(sizeof(float)));
        VDB_TRY(xs->ws.q_nrm.reserve(sizeof(float)));
    }
    // From the second round on: one pass over a bf16 copy of the rows that rules out the rows a new centroid cannot
    // come nearer to than their minimum, and the distance pass's own arithmetic for the others (km_pp_bf16_kernel):
    // half the bytes per round, two launches.  One-leaf rows (up to 128 columns), enough rounds to pay for the copy.
    // (VDB_KMPP_NO_BF16 is read at every call: tests compare the two ways.)
    const int ld16 = ivf_mfma_pitch(x.dim);
    bool bf16_rounds = short_round && !getenv("VDB_KMPP_NO_BF16") && x.dim <= 128 && x.plan.n_leaves == 1 && k >= 16 && x.n >= 4096;
    if (bf16_rounds) {
        const int st = h->x16.reserve((size_t)round_up64(x.n, 32) * ld16 * 2);
        if (st == VDB_ERR_NOMEM) { (void)hipGetLastError(); bf16_rounds = false; }
        else VDB_TRY(st);
    }
    if (bf16_rounds) {
        VDB_TRY(h->x_max_nrm.reserve(sizeof(float)));
        VDB_TRY(h->q16f.reserve((size_t)ld16 * sizeof(float)));
        VDB_TRY(launch_ivf_round_rows_tiled(x.data, x.ld, x.dim, 0, x.n, h->x16.as<uint16_t>(), ld16, s));
        VDB_HIP(hipMemsetAsync(h->x_max_nrm.p, 0, sizeof(float), s));
        VDB_TRY(launch_max_norm(x.nrm, 0, x.n, h->x_max_nrm.as<float>(), s));
    }
    for (int j = 1; j < k; ++j) {
        if (short_round) {
            KmNextQuery next;
            next.x = x.data; next.ldx = x.ld; next.dsub = h->dsub; next.x_sq = x.sq; next.x_nrm = x.nrm;
            next.q = h->qrow.as<float>(); next.q_sq = xs->ws.q_sq.as<float>(); next.q_nrm = xs->ws.q_nrm.as<float>();
            if (bf16_rounds) { next.q16f = h->q16f.as<float>(); next.q16f_n = ld16; }
            // (the first query is gathered here, the later ones by the selection of the round before)
            if (j == 1)
                VDB_TRY(launch_km_gather_query(x.data, x.ld, h->dsub, chosen, x.sq, x.nrm, h->qrow.as<float>(),
                                               xs->ws.q_sq.as<float>(), xs->ws.q_nrm.as<float>(), s, next.q16f, next.q16f_n));
            if (bf16_rounds && j > 1) {
                KmBf16Pass bp{h->x16.as<uint16_t>(), ld16 / 8, x.data, x.ld, x.dim, x.sq, h->x_max_nrm.as<float>(),
                              h->qrow.as<float>(), h->q16f.as<float>(), xs->ws.q_sq.as<float>(), xs->ws.q_nrm.as<float>(),
                              x.n, h->nearest.as<float>(), h->bsums.as<double>()};
                VDB_TRY(launch_km_pp_bf16(bp, s));
                VDB_TRY(launch_km_pp_select(h->nearest.as<float>(), x.n, h->bsums.as<double>(), uniforms[j - 1], chosen + j,
                                            chosen + k, next, s));
                continue;
            }
            StreamJob more{};
            more.timed = true;
            if ((j & 1) != 0) more.reverse_items = (x.n + 255) / 256;      // (every other pass back to front, as below)
            VDB_TRY(stream_flat_sums(xs->ws, x, VDB_EUCLIDEAN, h->qrow.as<float>(), 1, h->dist.as<float>(), s, false, &more));
            VDB_TRY(launch_km_pp_step(h->dist.as<float>(), x.n, j == 1, h->nearest.as<float>(), h->bsums.as<double>(),
                                      uniforms[j - 1], chosen + j, chosen + k, s, xs->ws.q_sq.as<float>(), x.sq, &next));
            continue;
        }
        // distances of every row to the centroid chosen last: pairwise_euclidean against ONE new centroid; the
        // minimum over all chosen ones is kept as a running minimum (min is exact, so the result is the same)
        VDB_TRY(launch_km_gather(x.data, x.ld, h->dsub, chosen + (j - 1), nullptr, 1, h->qrow.as<float>(), nullptr, s));
        // (every other pass back to front: it starts on the rows the pass before left in the memory-side cache)
        VDB_TRY(flat_distances_impl(xs, VDB_EUCLIDEAN, h->qrow.as<float>(), 1, h->dist.as<float>(), s, (j & 1) != 0));
        VDB_TRY(launch_km_pp_step(h->dist.as<float>(), x.n, j == 1, h->nearest.as<float>(), h->bsums.as<double>(),
                                  uniforms[j - 1], chosen + j, chosen + k, s));
    }
    int64_t bad_weights = 0;
    VDB_HIP(hipMemcpyAsync(&bad_weights, chosen + k, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    VDB_REQUIRE(bad_weights == 0, "probabilities contain NaN: every training row coincides with a centroid chosen so far "
                                  "(k-means++ over fewer distinct rows than clusters), or a distance is not a number");
    VDB_TRY(launch_km_gather(x.data, x.ld, h->dsub, chosen, nullptr, k, h->cur32.as<float>(), h->cur64.as<double>(), s));
    if (out_rows) VDB_HIP(hipMemcpyAsync(out_rows, chosen, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    h->f64 = false; h->have_next = false;
    return VDB_OK;
}

int vdb_kmeans_step(vdb_kmeans* h, int form, int64_t* out_counts, double* out_shift) {
    VDB_REQUIRE(h, "null handle");
    VDB_REQUIRE(form == VDB_KMEANS_IVF || form == VDB_KMEANS_PQ, "unknown k-means form %d", form);
    VDB_REQUIRE(form == VDB_KMEANS_PQ || !h->f64, "the float64 centroids of the PQ form cannot go back to the KMeans form");
    Entry g(h);
    hipStream_t s = h->stream;
    const Rows& x = h->x();
    const int k = h->k;
    // ---- assignment
    if (!h->f64) {
        VDB_TRY(kmeans_assign_f32(h, form == VDB_KMEANS_IVF ? VDB_EUCLIDEAN : VDB_EUCLIDEAN_SQ, s));
    } else {
        VDB_TRY(h->assign.reserve((size_t)x.n * sizeof(int)));
        VDB_TRY(h->csq64.reserve((size_t)k * sizeof(double)));
        VDB_TRY(launch_km_assign_f64(x.data, x.ld, h->dsub, x.n, x.sq, h->cur64.as<double>(), h->csq64.as<double>(), k,
                                     h->assign.as<int>(), s));
    }
    // ---- rows sorted by cluster, sample order kept inside a cluster; sequential sums
    int64_t chunk; int nb;
    km_sort_chunk(x.n, &chunk, &nb);
    VDB_TRY(h->hist.reserve((size_t)nb * k * sizeof(int)));
    VDB_TRY(h->totals.reserve((size_t)k * sizeof(int)));
    VDB_TRY(h->cluster_off.reserve((size_t)(k + 1) * sizeof(int)));
    VDB_TRY(h->members.reserve((size_t)x.n * sizeof(int)));
    VDB_TRY(h->counts.reserve((size_t)k * sizeof(int64_t)));
    VDB_TRY(h->scal.reserve(64));
    VDB_TRY(launch_km_sort(h->assign.as<int>(), x.n, k, h->hist.as<int>(), h->totals.as<int>(), h->cluster_off.as<int>(),
                           h->members.as<int>(), s));
    const int sum_form = form == VDB_KMEANS_IVF ? 0 : (h->f64 ? 2 : 1);
    VDB_TRY(launch_km_sums(sum_form, x.data, x.ld, h->dsub, h->members.as<int>(), h->cluster_off.as<int>(), k,
                           h->next32.as<float>(), h->next64.as<double>(), h->counts.as<long long>(), s));
    VDB_TRY(launch_km_shift(h->next32.as<float>(), h->cur32.as<float>(), (int64_t)k * h->dsub, h->scal.as<double>(), s));
    if (out_counts) VDB_HIP(hipMemcpyAsync(out_counts, h->counts.p, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (out_shift) VDB_HIP(hipMemcpyAsync(out_shift, h->scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    h->have_next = true;
    h->last_form = form;
    return VDB_OK;
}

int vdb_kmeans_reseed(vdb_kmeans* h, const int32_t* clusters, const int64_t* rows, int count, double* out_shift) {
    VDB_REQUIRE(h && (count == 0 || (clusters && rows)) && count >= 0, "bad arguments to vdb_kmeans_reseed");
    VDB_REQUIRE(h->have_next, "vdb_kmeans_reseed follows vdb_kmeans_step");
    for (int i = 0; i < count; ++i)
        VDB_REQUIRE(clusters[i] >= 0 && clusters[i] < h->k && rows[i] >= 0 && rows[i] < h->x().n, "cluster or row out of range");
    Entry g(h);
    hipStream_t s = h->stream;
    if (count > 0) {
        VDB_TRY(h->rows_in.reserve((size_t)count * sizeof(int64_t)));
        VDB_TRY(h->dst_in.reserve((size_t)count * sizeof(int)));
        VDB_HIP(hipMemcpyAsync(h->rows_in.p, rows, (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice, s));
        VDB_HIP(hipMemcpyAsync(h->dst_in.p, clusters, (size_t)count * sizeof(int), hipMemcpyHostToDevice, s));
        VDB_TRY(launch_km_gather(h->x().data, h->x().ld, h->dsub, h->rows_in.as<long long>(), h->dst_in.as<int>(), count,
                                 h->next32.as<float>(), h->next64.as<double>(), s));
    }
    VDB_TRY(launch_km_shift(h->next32.as<float>(), h->cur32.as<float>(), (int64_t)h->k * h->dsub, h->scal.as<double>(), s));
    if (out_shift) VDB_HIP(hipMemcpyAsync(out_shift, h->scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_kmeans_commit(vdb_kmeans* h) {
    VDB_REQUIRE(h && h->have_next, "vdb_kmeans_commit follows vdb_kmeans_step");
    Entry g(h);
    std::swap(h->cur32, h->next32);
    std::swap(h->cur64, h->next64);
    h->f64 = h->last_form == VDB_KMEANS_PQ;
    h->have_next = false;
    return VDB_OK;
}

int vdb_kmeans_centroids(vdb_kmeans* h, float* out32, double* out64) {
    VDB_REQUIRE(h && (out32 || out64), "null argument");
    Entry g(h);
    hipStream_t s = h->stream;
    const size_t cnt = (size_t)h->k * h->dsub;
    if (out32) VDB_HIP(hipMemcpyAsync(out32, h->cur32.p, cnt * sizeof(float), hipMemcpyDeviceToHost, s));
    if (out64) {
        VDB_REQUIRE(h->f64, "the centroids are fp32 (float64 centroids exist after a PQ-form step)");
        VDB_HIP(hipMemcpyAsync(out64, h->cur64.p, cnt * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_kmeans_set_centroids(vdb_kmeans* h, const float* centroids) {
    VDB_REQUIRE(h && centroids, "null argument");
    Entry g(h);
    hipStream_t s = h->stream;
    VDB_HIP(hipMemcpyAsync(h->cur32.p, centroids, (size_t)h->k * h->dsub * sizeof(float), hipMemcpyHostToDevice, s));
    VDB_HIP(hipStreamSynchronize(s));
    h->f64 = false; h->have_next = false;
    return VDB_OK;
}

int vdb_kmeans_assignments(vdb_kmeans* h, int32_t* out_assign, float* out_dist) {
    VDB_REQUIRE(h && (out_assign || out_dist), "null argument");
    Entry g(h);
    hipStream_t s = h->stream;
    const int64_t n = h->x().n;
    if (!h->have_next) {
        // no step pending: assign against the current centroids in the KMeans form (KMeans.predict, ivf.py:233-237)
        VDB_REQUIRE(!h->f64, "assignments of float64 centroids are those of the last step");
        VDB_TRY(kmeans_assign_f32(h, VDB_EUCLIDEAN, s));
    }
    if (out_assign) VDB_HIP(hipMemcpyAsync(out_assign, h->assign.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    if (out_dist) {
        VDB_REQUIRE(h->dist.cap >= (size_t)n * sizeof(float) && !(h->have_next && h->f64), "no fp32 distances for this step");
        VDB_HIP(hipMemcpyAsync(out_dist, h->dist.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

// ============================================================== IVF
int vdb_ivf_create(vdb_ivf** out, int dim, int metric) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(out != nullptr, "null output handle");
    VDB_REQUIRE(valid_metric(metric), "unknown metric code %d", metric);
    vdb_ivf* h = new vdb_ivf();
    h->dim = dim;
    h->metric = metric;
    int st = vdb_rows_create(&h->centroids, dim);
    if (st == VDB_OK) st = vdb_rows_create(&h->members, dim);
    if (st == VDB_OK) st = h->open();
    if (st != VDB_OK) { vdb_rows_destroy(h->centroids); vdb_rows_destroy(h->members); delete h; return st; }
    *out = h;
    return VDB_OK;
}

int vdb_ivf_destroy(vdb_ivf* h) {
    if (!h) return VDB_OK;
    {
        Entry g(h);
        (void)hipStreamSynchronize(h->stream);
        if (h->last_stream && h->last_stream != h->stream) (void)hipStreamSynchronize(h->last_stream);
        h->lay.dev.release(); h->row_ids.release(); h->ws.release();
        h->app_rows.release(); h->app_sq.release(); h->app_nrm.release(); h->app_dst.release(); h->app_no.release();
        h->shadow16.release(); h->max_nrm.release(); h->cent16.release(); h->cent_max_nrm.release();
    }
    vdb_rows_destroy(h->centroids);
    vdb_rows_destroy(h->members);
    h->close();
    delete h;
    return VDB_OK;
}

int vdb_ivf_set_centroids(vdb_ivf* h, const float* centroids, int n_lists) {
    VDB_REQUIRE(h && centroids && n_lists >= 1, "bad arguments to vdb_ivf_set_centroids");
    Entry g(h);
    VDB_TRY(vdb_rows_truncate(h->centroids, 0));
    VDB_TRY(vdb_rows_append(h->centroids, centroids, n_lists));
    h->n_lists = n_lists;
    h->lay.reset(n_lists);
    h->cent16_stale = true;
    h->shadow_stale = true;
    h->shadow_dirty.clear();
    VDB_TRY(vdb_rows_truncate(h->members, 0));
    return VDB_OK;
}

// rows[gather[i]] becomes the row at position i and answers to the row number numbers[i]
static int ivf_build_impl(vdb_ivf* h, const float* rows, int64_t n, const int64_t* gather, int64_t n_src, const int64_t* numbers,
                          const int64_t* list_offsets) {
    VDB_REQUIRE(h && h->n_lists >= 1, "centroids must be set before the lists are built");
    VDB_REQUIRE(n >= 0 && (n == 0 || (rows && numbers)) && list_offsets, "bad arguments to the list build");
    VDB_REQUIRE(list_offsets[0] == 0 && list_offsets[h->n_lists] == n, "list offsets do not cover %lld rows", (long long)n);
    VDB_REQUIRE(n < (int64_t)0x7fffffff, "at most 2^31-1 rows per device");
    for (int l = 0; l < h->n_lists; ++l)
        VDB_REQUIRE(list_offsets[l] <= list_offsets[l + 1], "list offsets must be non-decreasing");
    std::vector<int> perm((size_t)n), nums((size_t)n);
    int64_t row_limit = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t g = gather ? gather[i] : i;
        VDB_REQUIRE(g >= 0 && g < n_src, "row %lld out of range", (long long)g);
        VDB_REQUIRE(numbers[i] >= 0 && numbers[i] < (int64_t)0x7fffffff, "row number %lld out of range", (long long)numbers[i]);
        perm[(size_t)i] = (int)g;
        nums[(size_t)i] = (int)numbers[i];
        row_limit = std::max(row_limit, numbers[i] + 1);
    }
    Entry g(h);
    hipStream_t s = h->stream;
    vdb_rows* m = h->members;
    m->r.n = 0;
    h->shadow_stale = true;
    h->shadow_dirty.clear();
    h->lay.set_packed(list_offsets);
    h->lay.row_limit = row_limit;
    VDB_TRY(h->lay.sync(s));
    if (n == 0) return VDB_OK;
    VDB_TRY(h->row_ids.reserve((size_t)n * sizeof(int)));
    // a list's 256-row chunks start anywhere, so the buffer keeps 256 rows of room behind the last position
    VDB_TRY(rows_reserve(m, n + 256, s));
    // rows up in the caller's order, gathered into list order on the device
    DevBuf raw, d_perm;
    int st = raw.reserve((size_t)n_src * h->dim * sizeof(float));
    if (st == VDB_OK) st = d_perm.reserve((size_t)n * sizeof(int));
    if (st == VDB_OK) {
        auto body = [&]() -> int {
            VDB_HIP(hipMemcpyAsync(h->row_ids.p, nums.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
            VDB_HIP(hipMemcpyAsync(d_perm.p, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
            VDB_HIP(hipMemcpyAsync(raw.p, rows, (size_t)n_src * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(n * m->r.ld)), dim3(256), 0, s, raw.as<float>(), (int64_t)h->dim,
                               h->dim, n, d_perm.as<int>(), m->r.data, m->r.ld);
            VDB_HIP(hipGetLastError());
            VDB_TRY(row_norms(m->ws, m->r.data, m->r.ld, m->r.dim, n, m->r.plan, m->r.sq, m->r.nrm, s));
            VDB_HIP(hipStreamSynchronize(s));
            return VDB_OK;
        };
        st = body();
    }
    raw.release(); d_perm.release();
    if (st == VDB_OK) {
        m->r.n = n;
    } else {
        h->lay.reset(h->n_lists);       // no rows went in: the lists are empty, not half described
        m->r.n = 0;
    }
    return st;
}

int vdb_ivf_build_lists(vdb_ivf* h, const float* rows, int64_t n, const int64_t* list_rows, const int64_t* list_offsets) {
    VDB_REQUIRE(n == 0 || list_rows, "bad arguments to vdb_ivf_build_lists");
    return ivf_build_impl(h, rows, n, list_rows, n, list_rows, list_offsets);
}

int vdb_ivf_build_numbered(vdb_ivf* h, const float* rows, int64_t n, const int64_t* row_numbers, const int64_t* list_offsets) {
    return ivf_build_impl(h, rows, n, nullptr, n, row_numbers, list_offsets);
}

// one thread per float of the padded destination rows
__global__ void ivf_scatter_rows_kernel(const float* __restrict__ src, int ld, int64_t count, const int64_t* __restrict__ dst,
                                        const int* __restrict__ numbers, const float* __restrict__ sq, const float* __restrict__ nrm,
                                        float* __restrict__ data, float* __restrict__ osq, float* __restrict__ onrm,
                                        int* __restrict__ row_ids) {
    const int64_t total = count * ld;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / ld;
        const int c = (int)(t - i * ld);
        const int64_t d = dst[i];
        data[d * ld + c] = src[t];
        if (c == 0) { osq[d] = sq[i]; onrm[d] = nrm[i]; row_ids[d] = numbers[i]; }
    }
}

__global__ void ivf_move_rows_kernel(float* __restrict__ data, float* __restrict__ sq, float* __restrict__ nrm, int* __restrict__ row_ids,
                                     int ld, int64_t from, int64_t to, int64_t count) {
    const int64_t total = count * ld;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        data[to * ld + t] = data[from * ld + t];
        if (t < count) { sq[to + t] = sq[from + t]; nrm[to + t] = nrm[from + t]; row_ids[to + t] = row_ids[from + t]; }
    }
}

int vdb_ivf_append(vdb_ivf* h, const float* rows, const int64_t* row_numbers, const int32_t* lists, int64_t count) {
    VDB_REQUIRE(h && h->n_lists >= 1, "centroids must be set before rows are appended");
    VDB_REQUIRE(count >= 0 && (count == 0 || (rows && row_numbers && lists)), "bad arguments to vdb_ivf_append");
    if (count == 0) return VDB_OK;
    std::vector<int> nums((size_t)count);
    int64_t row_limit = h->lay.row_limit;
    for (int64_t i = 0; i < count; ++i) {
        VDB_REQUIRE(row_numbers[i] >= 0 && row_numbers[i] < (int64_t)0x7fffffff, "row number %lld out of range", (long long)row_numbers[i]);
        nums[(size_t)i] = (int)row_numbers[i];
        row_limit = std::max(row_limit, row_numbers[i] + 1);
    }
    Entry g(h);
    hipStream_t s = h->stream;
    vdb_rows* m = h->members;
    Rows& r = m->r;
    ListLayout& lay = h->lay;
    std::vector<int64_t> dst;
    std::vector<ListMove> moves;
    const int64_t old_tail = lay.tail;
    ListLayout::Undo undo;
    VDB_TRY(lay.plan_append(lists, count, dst, moves, undo));
    // Everything that can fail comes before the first row is written; if it does, the plan is taken back and the
    // index is what it was (buffers that have grown meanwhile keep their contents).
    Workspace& ws = h->ws;
    r.n = old_tail;         // room for every position handed out so far: the rows behind r.n are kept when buffers grow
    const int reserved = [&]() -> int {
        VDB_REQUIRE(lay.tail < (int64_t)0x7fffffff, "at most 2^31-1 positions per device");
        VDB_TRY(rows_reserve(m, lay.tail + 256, s));     // 256 rows of room behind the last position, as after a build
        VDB_TRY(h->row_ids.reserve_keep((size_t)r.cap * sizeof(int), (size_t)old_tail * sizeof(int), s));
        VDB_TRY(ws.stage.reserve((size_t)count * h->dim * sizeof(float)));
        VDB_TRY(h->app_rows.reserve((size_t)count * r.ld * sizeof(float)));
        VDB_TRY(h->app_sq.reserve((size_t)count * sizeof(float)));
        VDB_TRY(h->app_nrm.reserve((size_t)count * sizeof(float)));
        VDB_TRY(h->app_dst.reserve((size_t)count * sizeof(int64_t)));
        VDB_TRY(h->app_no.reserve((size_t)count * sizeof(int)));
        return VDB_OK;
    }();
    if (reserved != VDB_OK) { lay.rollback(undo); return reserved; }
    lay.row_limit = row_limit;
    r.n = lay.tail;
    for (const ListMove& mv : moves) h->shadow_note(mv.to, mv.to + mv.count);
    if (count <= 16) {
        for (int64_t i = 0; i < count; ++i) h->shadow_note(dst[(size_t)i], dst[(size_t)i] + 1);
    } else {
        const auto mm = std::minmax_element(dst.begin(), dst.end());
        h->shadow_note(*mm.first, *mm.second + 1);
    }
    for (const ListMove& mv : moves) {
        hipLaunchKernelGGL(ivf_move_rows_kernel, dim3(grid_for(mv.count * r.ld)), dim3(256), 0, s, r.data, r.sq, r.nrm,
                           h->row_ids.as<int>(), r.ld, mv.from, mv.to, mv.count);
    }
    VDB_HIP(hipGetLastError());
    // the new rows: padded copy, norms in the order every other row's norms were summed in, then into their places
    VDB_HIP(hipMemcpyAsync(ws.stage.p, rows, (size_t)count * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
    VDB_HIP(hipMemcpyAsync(h->app_dst.p, dst.data(), (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice, s));
    VDB_HIP(hipMemcpyAsync(h->app_no.p, nums.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(count * r.ld)), dim3(256), 0, s, ws.stage.as<float>(), (int64_t)h->dim, h->dim,
                       count, (const int*)nullptr, h->app_rows.as<float>(), r.ld);
    VDB_HIP(hipGetLastError());
    VDB_TRY(row_norms(m->ws, h->app_rows.as<float>(), r.ld, r.dim, count, r.plan, h->app_sq.as<float>(), h->app_nrm.as<float>(), s));
    hipLaunchKernelGGL(ivf_scatter_rows_kernel, dim3(grid_for(count * r.ld)), dim3(256), 0, s, h->app_rows.as<float>(), r.ld, count,
                       h->app_dst.as<int64_t>(), h->app_no.as<int>(), h->app_sq.as<float>(), h->app_nrm.as<float>(), r.data, r.sq,
                       r.nrm, h->row_ids.as<int>());
    VDB_HIP(hipGetLastError());
    VDB_HIP(hipStreamSynchronize(s));          // the host vectors above are pageable
    return VDB_OK;
}

int vdb_ivf_remove_at(vdb_ivf* h, int32_t list, int64_t position) {
    VDB_REQUIRE(h && list >= 0 && list < h->n_lists, "bad list number");
    Entry g(h);
    ListLayout& lay = h->lay;
    VDB_REQUIRE(position >= 0 && position < lay.len[list], "position %lld is not in list %d (%lld rows)", (long long)position, list,
                (long long)lay.len[list]);
    Rows& r = h->members->r;
    const int64_t last = lay.len[list] - 1;
    // the list's last row takes the place of the removed one (index/ivf.py:708-745)
    if (position < last) {
        hipLaunchKernelGGL(ivf_move_rows_kernel, dim3(1), dim3(256), 0, h->stream, r.data, r.sq, r.nrm, h->row_ids.as<int>(), r.ld,
                           lay.beg[list] + last, lay.beg[list] + position, (int64_t)1);
        VDB_HIP(hipGetLastError());
        h->shadow_note(lay.beg[list] + position, lay.beg[list] + position + 1);
    }
    --lay.len[list];
    --lay.live;
    lay.dev_stale = true;
    return VDB_OK;
}

int vdb_ivf_layout(vdb_ivf* h, int64_t* live_rows, int64_t* positions, int64_t* dead_positions) {
    VDB_REQUIRE(h, "null handle");
    Entry g(h);
    if (live_rows) *live_rows = h->lay.live;
    if (positions) *positions = h->lay.tail;
    if (dead_positions) *dead_positions = h->lay.dead;
    return VDB_OK;
}

int vdb_ivf_build(vdb_ivf* h, const float* rows, int64_t n, const int32_t* list_of_row) {
    VDB_REQUIRE(h && h->n_lists >= 1, "centroids must be set before the lists are built");
    VDB_REQUIRE(n >= 0 && (n == 0 || (rows && list_of_row)), "bad arguments to vdb_ivf_build");
    std::vector<int64_t> off((size_t)h->n_lists + 1, 0), order((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        VDB_REQUIRE(list_of_row[i] >= 0 && list_of_row[i] < h->n_lists, "list number %d out of range", list_of_row[i]);
        off[(size_t)list_of_row[i] + 1]++;
    }
    for (int l = 0; l < h->n_lists; ++l) off[l + 1] += off[l];
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    for (int64_t i = 0; i < n; ++i) order[(size_t)cur[list_of_row[i]]++] = i;      // stable: insertion order inside a list
    return vdb_ivf_build_lists(h, rows, n, order.data(), off.data());
}

static int ivf_probe_impl(vdb_ivf* h, const float* q_pad_unused, const float* d_q, int nq, int n_probe, int* d_probes,
                          float* d_probe_dist, hipStream_t s) {
    (void)q_pad_unused;
    Workspace& ws = h->ws;
    const int cm = scan_metric_of(h->metric);
    // a few queries over at most 1024 centroids: one full sort of the distance row in LDS is quicker than
    // selection plus list merge (0.197 -> 0.179 ms per query at 1024 lists; slower from 4096 lists on)
    const bool few_small = nq <= 8 && h->n_lists <= 1024;
    if (n_probe <= KMAX_FUSED && !few_small) {
        VDB_TRY(flat_search_impl(h->centroids, cm, d_q, nq, n_probe, nullptr, d_probe_dist, nullptr, d_probes, s));
    } else {
        VDB_REQUIRE(h->n_lists <= 16384, "n_probe > %d needs n_lists <= 16384", KMAX_FUSED);
        VDB_TRY(ws.dist.reserve((size_t)nq * h->n_lists * sizeof(float)));
        VDB_TRY(flat_distances_impl(h->centroids, cm, d_q, nq, ws.dist.as<float>(), s));
        VDB_TRY(launch_row_sort_topk(ws.dist.as<float>(), nq, h->n_lists, n_probe, d_probe_dist, d_probes, s));
    }
    return VDB_OK;
}

// One query: four launches.  Centroid sums (the streaming scan over the centroid rows) -> one block that takes the
// query's norms, ranks the centroids and lays out the scan (launch_ivf_one_plan) -> list scan -> selection whose
// last block merges and maps slots to rows.  Same values and the same order as the batch path, which spends
// fifteen launches on a single query.
static bool ivf_one_query_ok(const vdb_ivf* h, int nq, int k, int n_probe) {
    static const bool off = getenv("VDB_IVF_NO_ONE") != nullptr;
    const Rows& m = h->members->r;
    const Rows& c = h->centroids->r;
    return !off && nq == 1 && k <= KMAX_FUSED && n_probe <= 64 && m.ld == m.dim && m.dim % 8 == 0 && !m.bf16 && !c.bf16 &&
           h->n_lists <= 16384;
}

static int ivf_search_one(vdb_ivf* h, const float* d_q, int k, int n_probe, const uint32_t* d_mask,
                          float* d_out_d, int64_t* d_out_i, int64_t* d_out_order, hipStream_t s) {
    vdb_rows* m = h->members;
    vdb_rows* c = h->centroids;
    Workspace& ws = h->ws;
    const int sm = scan_metric_of(h->metric);
    const int64_t k_eff = std::min<int64_t>(k, h->lay.live);
    const int64_t cand_stride = round_up64(std::max<int64_t>(h->probe_bound(n_probe), 1), 64);
    VDB_REQUIRE(cand_stride < (int64_t)0x7fffffff, "n_probe=%d lists can hold more than 2^31 rows", n_probe);
    h->last_nq = 1; h->last_nprobe = n_probe; h->last_scan_rows = -1; h->last_stream = s;

    // 1: the query against every centroid
    VDB_TRY(ws.dist.reserve((size_t)h->n_lists * sizeof(float)));
    VDB_TRY(stream_flat_sums(c->ws, c->r, sm, d_q, 1, ws.dist.as<float>(), s));
    // 2: norms, ranking, layout of the scan
    StreamPlanBufs b{};
    const int64_t items_cap = cand_stride / 256 + n_probe;
    VDB_TRY(reserve_stream_plan(ws, 1, n_probe, items_cap, &b));
    VDB_TRY(ws.probes.reserve((size_t)n_probe * sizeof(int)));
    VDB_TRY(ws.probe_d.reserve((size_t)n_probe * sizeof(float)));
    VDB_TRY(ws.q_sq.reserve(sizeof(float)));
    VDB_TRY(ws.q_nrm.reserve(sizeof(float)));
    VDB_TRY(ws.cand.reserve((size_t)cand_stride * sizeof(float)));
    VDB_TRY(ws.cand_off.reserve((size_t)(n_probe + 1) * sizeof(int64_t)));
    VDB_TRY(ws.tile_off.reserve((size_t)(n_probe + 1) * sizeof(int)));
    VDB_TRY(launch_ivf_one_plan(ws.dist.as<float>(), sm, c->r.sq, c->r.nrm, c->r.cos_eps, d_q, m->r.plan, h->n_lists, n_probe,
                                h->lay.d_beg(), h->lay.d_end(), ws.q_sq.as<float>(), ws.q_nrm.as<float>(), ws.probes.as<int>(),
                                ws.probe_d.as<float>(), ws.cand_off.as<int64_t>(), ws.tile_off.as<int>(), b, s));
    // 3: the probed lists
    StreamJob j{};
    j.data = m->r.data; j.ld = m->r.ld; j.dim = m->r.dim; j.plan = m->r.plan;
    j.op = op_of_metric(sm); j.post = post_of_metric(sm);
    if (j.post == POST_SQRT) j.post = POST_NONE;          // the selection takes the root (CandJob::sqrt_here)
    j.q = d_q; j.ldq = m->r.ld; j.group = 1; j.n_seg = 1;
    j.seg_row_off = nullptr; j.seg_q_off = nullptr; j.seg_item_off = b.seg_item_off;
    j.sq_qid = b.sq_qid; j.sq_out = b.sq_out; j.items = b.items; j.items_cap = b.items_cap;
    j.out = ws.cand.as<float>(); j.timed = true;
    VDB_TRY(launch_stream_sums(j, s));
    // 4: the k best, merged and mapped to rows by the block that finishes last
    const int64_t avg_tiles = std::max<int64_t>(1, ((int64_t)n_probe * (h->lay.live / std::max(1, h->n_lists)) + 255) / 256);
    // one step of the selection walk per block where the candidates allow it (the walk is a chain of dependent
    // loads); the blocks' lists are merged by the last of them, so not more than 64
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(64, avg_tiles / 8));
    const int n_cand = splits;                                 // one list per block in this mode
    const int kk = (int)std::max<int64_t>(k_eff, 1);
    VDB_TRY(ws.part_d.reserve((size_t)n_cand * kk * sizeof(float)));
    VDB_TRY(ws.part_i.reserve((size_t)n_cand * kk * sizeof(int)));
    VDB_TRY(ws.ticket.reserve(sizeof(int)));
    if (ws.ticket_zeroed != ws.ticket.p) {
        VDB_HIP(hipMemsetAsync(ws.ticket.p, 0, sizeof(int), s));
        ws.ticket_zeroed = ws.ticket.p;
    }
    CandJob cj{};
    cj.cand = ws.cand.as<float>(); cj.cand_stride = cand_stride; cj.nq = 1; cj.n = 0;
    cj.probes = ws.probes.as<int>(); cj.n_probe = n_probe; cj.n_lists = h->n_lists;
    cj.list_beg = h->lay.d_beg(); cj.list_end = h->lay.d_end();
    cj.cand_off = ws.cand_off.as<int64_t>(); cj.tile_off = ws.tile_off.as<int>();
    cj.row_ids = h->row_ids.as<int>(); cj.mask = d_mask;
    cj.sqrt_here = true;
    cj.metric = sm; cj.q_sq = ws.q_sq.as<float>(); cj.q_nrm = ws.q_nrm.as<float>();
    cj.row_sq = m->r.sq; cj.row_nrm = m->r.nrm;
    cj.k = kk; cj.splits = splits; cj.part_d = ws.part_d.as<float>(); cj.part_i = ws.part_i.as<int>();
    cj.ticket = ws.ticket.as<int>(); cj.k_out = k; cj.fin_d = d_out_d; cj.fin_rows = d_out_i; cj.fin_order = d_out_order;
    return launch_topk_cand(cj, s);
}

// bf16 copy of the member rows up to date: everything after a build, otherwise the ranges writes have touched.
// VDB_ERR_NOMEM: no room for the copy (the caller then scans the fp32 rows alone).
static int ivf_shadow_sync(vdb_ivf* h, hipStream_t s) {
    Rows& r = h->members->r;
    const int ld16 = ivf_mfma_pitch(r.dim);
    if (!h->max_nrm.p) {
        VDB_TRY(h->max_nrm.reserve(sizeof(float)));
        h->shadow_stale = true;
    }
    if (h->shadow_cap < r.cap || !h->shadow16.p) {
        const size_t want = (size_t)r.cap * ld16 * 2;
        if (h->prefilter == VDB_PREFILTER_AUTO) {
            // the copy is taken only while it leaves a tenth of the device (at least 2 GiB) free for everything else
            size_t free_b = 0, total_b = 0;
            VDB_HIP(hipMemGetInfo(&free_b, &total_b));
            if (free_b + h->shadow16.cap < want + std::max<size_t>(total_b / 10, (size_t)2 << 30)) return VDB_ERR_NOMEM;
        }
        // positions past the copy's old end are converted below (they are new, hence noted or stale)
        // (whole tiles of 32 positions: the copy interleaves the rows of a tile)
        VDB_TRY(h->shadow16.reserve_keep(want, h->shadow_stale ? 0 : (size_t)round_up64(std::min(h->shadow_cap, r.n), 32) * ld16 * 2, s));
        h->shadow_cap = r.cap;
    }
    uint16_t* dst = h->shadow16.as<uint16_t>();
    if (h->shadow_stale) {
        VDB_TRY(launch_ivf_round_rows_tiled(r.data, r.ld, r.dim, 0, r.n, dst, ld16, s));
        VDB_HIP(hipMemsetAsync(h->max_nrm.p, 0, sizeof(float), s));
        VDB_TRY(launch_max_norm(r.nrm, 0, r.n, h->max_nrm.as<float>(), s));
        h->shadow_stale = false;
    } else {
        for (const auto& d : h->shadow_dirty) {
            const int64_t from = std::max<int64_t>(d.first, 0), to = std::min(d.second, r.n);
            if (to <= from) continue;
            VDB_TRY(launch_ivf_round_rows_tiled(r.data + from * r.ld, r.ld, r.dim, from, to - from, dst, ld16, s));
            // a running maximum: a row that was overwritten may leave it larger than needed, never smaller
            VDB_TRY(launch_max_norm(r.nrm, from, to, h->max_nrm.as<float>(), s));
        }
    }
    h->shadow_dirty.clear();
    return VDB_OK;
}

// bf16 copy of the centroids (row-major) and their largest norm, made again after vdb_ivf_set_centroids
static int ivf_cent16_sync(vdb_ivf* h, hipStream_t s) {
    if (!h->cent16_stale && h->cent16.p) return VDB_OK;
    const Rows& c = h->centroids->r;
    const int ld16 = ivf_mfma_pitch(c.dim);
    VDB_TRY(h->cent16.reserve((size_t)h->n_lists * ld16 * 2));
    VDB_TRY(h->cent_max_nrm.reserve(sizeof(float)));
    VDB_TRY(launch_ivf_round_rows(c.data, c.ld, c.dim, h->n_lists, h->cent16.as<uint16_t>(), ld16, s));
    VDB_HIP(hipMemsetAsync(h->cent_max_nrm.p, 0, sizeof(float), s));
    VDB_TRY(launch_max_norm(c.nrm, 0, h->n_lists, h->cent_max_nrm.as<float>(), s));
    h->cent16_stale = false;
    return VDB_OK;
}

static constexpr int IVF_MFMA_GROUP = 32;           // queries per item: the columns of one 32 x 32 accumulator
static constexpr int IVF_MFMA_QUEUE = 8192;         // most (slot, distance) entries per query
static constexpr int IVF_MFMA_SEGMENT = 2048;       // kept pairs per wave of the scan pass

static bool ivf_mfma_ok(const vdb_ivf* h, int sm, int nq, int64_t k_eff) {
    static const bool off = getenv("VDB_IVF_NO_MFMA") != nullptr;
    const Rows& m = h->members->r;
    return !off && h->prefilter != VDB_PREFILTER_NONE && (sm == VDB_EUCLIDEAN_DIRECT || sm == VDB_EUCLIDEAN_SQ_DIRECT) && nq >= 2 &&
           k_eff <= KMAX_FUSED && !m.bf16 && m.dim <= 2048;
}

static int ivf_search_impl(vdb_ivf* h, const float* d_q, int nq, int k, int n_probe, const uint32_t* d_mask,
                           float* d_out_d, int64_t* d_out_i, int64_t* d_out_order, hipStream_t s) {
    VDB_REQUIRE(h->n_lists >= 1, "index is not trained");
    VDB_REQUIRE(nq >= 0 && k >= 1 && n_probe >= 1, "need nq >= 0, k >= 1, n_probe >= 1");
    if (nq == 0) return VDB_OK;
    n_probe = std::min(n_probe, h->n_lists);
    vdb_rows* m = h->members;
    Workspace& ws = h->ws;
    if (h->lay.live > 0 && ivf_one_query_ok(h, nq, k, n_probe)) {
        VDB_TRY(h->lay.sync(s));
        h->last_used_mfma = false;
        return ivf_search_one(h, d_q, k, n_probe, d_mask, d_out_d, d_out_i, d_out_order, s);     // writes all k results
    }
    VDB_TRY(fill_results(d_out_d, d_out_i, (int64_t)nq * k, s));
    if (d_out_order) VDB_TRY(fill_results(nullptr, d_out_order, (int64_t)nq * k, s));
    if (h->lay.live == 0) return VDB_OK;
    VDB_TRY(h->lay.sync(s));

    const int64_t k_eff = std::min<int64_t>(k, h->lay.live);
    const int kmax = (int)std::min<int64_t>(KMAX_FUSED, k_eff);
    const bool multi = k_eff > KMAX_FUSED;
    const int sm = scan_metric_of(h->metric);
    // a query's candidate slots: at most the n_probe longest lists
    const int64_t cand_stride = round_up64(std::max<int64_t>(h->probe_bound(n_probe), 1), 64);
    VDB_REQUIRE(cand_stride < (int64_t)0x7fffffff, "n_probe=%d lists can hold more than 2^31 rows", n_probe);
    const int q_block = (int)std::max<int64_t>(1, std::min<int64_t>(nq, CAND_BUDGET / (cand_stride * (int64_t)sizeof(float))));
    // queries per row chunk: 8, or up to 128 with the row-resident kernel (VDB_STREAM_GROUP=8 or 16 for A/B timing)
    int group = nq == 1 ? 1 : stream_batch_group(m->r.dim, m->r.ld, m->r.plan, op_of_metric(sm), 8, -1);
    // Euclidean batches: the scan on the matrix cores over bf16 copies of the rows, the survivors recomputed in the
    // reference's arithmetic; the streaming scan below then runs only behind the overflow flag
    bool mfma = ivf_mfma_ok(h, sm, nq, k_eff);
    if (mfma) {
        const int st = ivf_shadow_sync(h, s);
        if (st == VDB_ERR_NOMEM) { (void)hipGetLastError(); mfma = false; }
        else VDB_TRY(st);
    }
    h->last_used_mfma = mfma;
    const int scan_group = group;                   // what the streaming kernel is configured for
    if (mfma && group > IVF_MFMA_GROUP) group = IVF_MFMA_GROUP;      // items of at most 32 queries serve both kernels

    VDB_TRY(ws.probes.reserve((size_t)nq * n_probe * sizeof(int)));
    VDB_TRY(ws.probe_d.reserve((size_t)nq * n_probe * sizeof(float)));
    const float* q = nullptr;
    const int ld16 = ivf_mfma_pitch(m->r.dim);
    // the lists to probe: ranked on the matrix cores too where the scan is (same lists, same order), else by the
    // exact search over the centroids
    static const bool no_mfma_probe = getenv("VDB_IVF_NO_MFMA_PROBE") != nullptr;
    const Rows& cr = h->centroids->r;
    const bool mfma_probe = mfma && !no_mfma_probe && n_probe <= 64 && !cr.bf16 && cr.ld == m->r.ld &&
                            (int64_t)nq * h->n_lists <= ((int64_t)1 << 28);
    if (mfma_probe) {
        VDB_TRY(prepare_queries(ws, d_q, nq, m->r.dim, m->r.ld, m->r.plan, &q, s));
        VDB_TRY(ws.q16.reserve((size_t)nq * ld16 * 2));
        VDB_TRY(launch_ivf_round_rows(q, m->r.ld, m->r.dim, nq, ws.q16.as<uint16_t>(), ld16, s));
        VDB_TRY(ivf_cent16_sync(h, s));
        VDB_TRY(ws.dist.reserve((size_t)nq * h->n_lists * sizeof(float)));
        IvfProbeJob pj{};
        pj.cent16 = h->cent16.as<uint16_t>(); pj.cent_sq = cr.sq; pj.max_cent_nrm = h->cent_max_nrm.as<float>();
        pj.cent = cr.data; pj.ld = cr.ld; pj.dim = cr.dim; pj.sum_plan = cr.plan; pj.n_lists = h->n_lists;
        pj.q = q; pj.ldq = m->r.ld; pj.q16 = ws.q16.as<uint16_t>(); pj.ld16 = ld16;
        pj.q_sq = ws.q_sq.as<float>(); pj.q_nrm = ws.q_nrm.as<float>(); pj.nq = nq;
        pj.n_probe = n_probe; pj.take_sqrt = sm == VDB_EUCLIDEAN_DIRECT;
        pj.keys = ws.dist.as<float>(); pj.probes = ws.probes.as<int>(); pj.probe_d = ws.probe_d.as<float>();
        VDB_TRY(launch_ivf_mfma_probe(pj, s));
    } else {
        ProfileMute only_the_scan(mfma);             // the matrix-core scan is the timed kernel of such a batch
        VDB_TRY(ivf_probe_impl(h, nullptr, d_q, nq, n_probe, ws.probes.as<int>(), ws.probe_d.as<float>(), s));
    }
    h->last_nq = nq; h->last_nprobe = n_probe; h->last_scan_rows = -1; h->last_stream = s;

    if (!mfma_probe) VDB_TRY(prepare_queries(ws, d_q, nq, m->r.dim, m->r.ld, m->r.plan, &q, s));
    StreamPlanBufs b{};
    // every (query, probe) pair adds at most ceil(list rows / 256) items
    VDB_TRY(reserve_stream_plan(ws, h->n_lists, (int64_t)q_block * n_probe, (int64_t)q_block * (cand_stride / 256 + n_probe), &b));
    VDB_TRY(ws.cand.reserve((size_t)q_block * cand_stride * sizeof(float)));
    VDB_TRY(ws.cand_off.reserve((size_t)q_block * (n_probe + 1) * sizeof(int64_t)));
    VDB_TRY(ws.tile_off.reserve((size_t)q_block * (n_probe + 1) * sizeof(int)));
    VDB_TRY(ws.list_cnt.reserve((size_t)h->n_lists * sizeof(int)));
    VDB_TRY(ws.cursor.reserve((size_t)h->n_lists * sizeof(int)));
    VDB_TRY(ws.slots.reserve((size_t)nq * k * sizeof(int)));
    if (multi) {
        VDB_TRY(ws.lb_d.reserve((size_t)q_block * sizeof(float)));
        VDB_TRY(ws.lb_k.reserve((size_t)q_block * sizeof(int64_t)));
    }
    const int64_t avg_tiles = std::max<int64_t>(1, ((int64_t)n_probe * (h->lay.live / std::max(1, h->n_lists)) + 255) / 256);
    const int* run_if = nullptr;
    // a queue as long as a query's candidates cannot overflow; longer ones are cut at IVF_MFMA_QUEUE entries
    const int queue_cap = (int)std::min<int64_t>(cand_stride, IVF_MFMA_QUEUE);
    if (mfma) {
        VDB_TRY(ws.q16.reserve((size_t)nq * ld16 * 2));
        VDB_TRY(ws.mins.reserve((size_t)q_block * ivf_mfma_subsets() * sizeof(float)));
        VDB_TRY(ws.thr_d.reserve((size_t)q_block * 2 * sizeof(float)));                // thresholds, slacks
        VDB_TRY(ws.queue.reserve((size_t)q_block * queue_cap * 3 * sizeof(int)));      // slots, keys, positions
        VDB_TRY(ws.seg.reserve((size_t)ivf_mfma_scan_waves() * IVF_MFMA_SEGMENT * sizeof(int4)));
        VDB_TRY(ws.seg_len.reserve((size_t)ivf_mfma_scan_waves() * sizeof(int)));
        VDB_TRY(ws.queue_cnt.reserve((size_t)q_block * ivf_mfma_len_stride() * sizeof(int)));
        VDB_TRY(ws.flag.reserve(sizeof(int)));
        if (!mfma_probe) VDB_TRY(launch_ivf_round_rows(q, m->r.ld, m->r.dim, nq, ws.q16.as<uint16_t>(), ld16, s));
        run_if = ws.flag.as<int>();
    }

    for (int q0 = 0; q0 < nq; q0 += q_block) {
        const int nb = std::min(q_block, nq - q0);
        const int* probes = ws.probes.as<int>() + (int64_t)q0 * n_probe;
        VDB_TRY(launch_ivf_plan(probes, nb, n_probe, h->lay.d_beg(), h->lay.d_end(), h->n_lists, group, cand_stride,
                                ws.list_cnt.as<int>(), ws.cursor.as<int>(), ws.cand_off.as<int64_t>(), ws.tile_off.as<int>(), b, s));
        if (mfma) {
            IvfMfmaJob mj{};
            mj.rows16 = h->shadow16.as<uint16_t>(); mj.ld16 = ld16; mj.row_sq = m->r.sq;
            mj.rows = m->r.data; mj.ld = m->r.ld; mj.dim = m->r.dim; mj.sum_plan = m->r.plan;
            mj.q = q + (int64_t)q0 * m->r.ld; mj.ldq = m->r.ld; mj.q16 = ws.q16.as<uint16_t>() + (int64_t)q0 * ld16;
            mj.q_sq = ws.q_sq.as<float>() + q0; mj.q_nrm = ws.q_nrm.as<float>() + q0; mj.nq = nb;
            mj.plan = b; mj.cand_stride = cand_stride;
            mj.probes = probes; mj.n_probe = n_probe; mj.n_lists = h->n_lists;
            mj.cand_off = ws.cand_off.as<int64_t>(); mj.list_beg = h->lay.d_beg();
            mj.row_ids = h->row_ids.as<int>(); mj.mask = d_mask; mj.max_row_nrm = h->max_nrm.as<float>();
            mj.mins = ws.mins.as<float>(); mj.thr = ws.thr_d.as<float>(); mj.queue_slot = ws.queue.as<int>();
            mj.queue_d = ws.queue.as<float>() + (size_t)q_block * queue_cap;
            mj.queue_pos = ws.queue.as<int>() + (size_t)q_block * queue_cap * 2;
            mj.slack = ws.thr_d.as<float>() + q_block;
            mj.seg = reinterpret_cast<int4*>(ws.seg.p); mj.seg_len = ws.seg_len.as<int>(); mj.seg_cap = IVF_MFMA_SEGMENT;
            mj.slots = ws.slots.as<int>() + (int64_t)q0 * k;
            mj.queue_len = ws.queue_cnt.as<int>(); mj.queue_cap = queue_cap; mj.overflow = ws.flag.as<int>();
            mj.k = kmax; mj.k_out = k; mj.take_sqrt = sm == VDB_EUCLIDEAN_DIRECT;
            mj.out_d = d_out_d + (int64_t)q0 * k; mj.out_rows = d_out_i + (int64_t)q0 * k;
            mj.out_order = d_out_order ? d_out_order + (int64_t)q0 * k : nullptr;
            mj.timed = true;
            VDB_TRY(launch_ivf_mfma_search(mj, s));
        }
        StreamJob j{};
        j.data = m->r.data; j.ld = m->r.ld; j.dim = m->r.dim; j.plan = m->r.plan;
        j.op = op_of_metric(sm); j.post = post_of_metric(sm);
        if (j.post == POST_SQRT) j.post = POST_NONE;      // the selection below takes the root (CandJob::sqrt_here)
        j.q = q + (int64_t)q0 * m->r.ld; j.ldq = m->r.ld; j.group = scan_group; j.n_seg = h->n_lists;
        j.seg_row_off = h->lay.d_beg(); j.seg_q_off = b.seg_q_off; j.seg_item_off = b.seg_item_off;
        j.sq_qid = b.sq_qid; j.sq_out = b.sq_out; j.items = b.items; j.items_cap = b.items_cap;
        j.out = ws.cand.as<float>(); j.timed = !mfma; j.run_if = run_if;
        VDB_TRY(launch_stream_sums(j, s));

        const int splits = cand_splits(avg_tiles, nb);
        const int n_cand = splits * topk_cand_lists_per_split();
        VDB_TRY(ws.part_d.reserve((size_t)nb * n_cand * kmax * sizeof(float)));
        VDB_TRY(ws.part_i.reserve((size_t)nb * n_cand * kmax * sizeof(int)));
        for (int64_t done = 0; done < k_eff; done += KMAX_FUSED) {
            const int kk = (int)std::min<int64_t>(KMAX_FUSED, k_eff - done);
            CandJob c{};
            c.cand = ws.cand.as<float>(); c.cand_stride = cand_stride; c.nq = nb; c.n = 0;
            const bool row_terms = !(sm == VDB_EUCLIDEAN_DIRECT || sm == VDB_EUCLIDEAN_SQ_DIRECT || sm == VDB_MANHATTAN ||
                                     sm == VDB_CHEBYSHEV || sm == VDB_HAMMING);        // cosine forms divide by the row's norm
            if (d_mask || row_terms) {
                c.probes = probes; c.n_probe = n_probe; c.n_lists = h->n_lists; c.list_beg = h->lay.d_beg(); c.list_end = h->lay.d_end();
                c.cand_off = ws.cand_off.as<int64_t>(); c.tile_off = ws.tile_off.as<int>();
                c.row_ids = h->row_ids.as<int>(); c.mask = d_mask;
            } else {
                // no filter, no row term: a query's slots are one contiguous run (probe after probe), and nothing about
                // the lists behind them is needed before the k results are mapped to rows
                c.n = cand_stride; c.n_of_query = ws.cand_off.as<int64_t>() + n_probe; c.n_stride = n_probe + 1;
            }
            c.sqrt_here = true;
            c.metric = sm; c.q_sq = ws.q_sq.as<float>() + q0; c.q_nrm = ws.q_nrm.as<float>() + q0;
            c.row_sq = m->r.sq; c.row_nrm = m->r.nrm;
            c.lb_d = done > 0 ? ws.lb_d.as<float>() : nullptr; c.lb_k = done > 0 ? ws.lb_k.as<int64_t>() : nullptr;
            c.k = kk; c.splits = splits; c.part_d = ws.part_d.as<float>(); c.part_i = ws.part_i.as<int>();
            c.run_if = run_if;
            VDB_TRY(launch_topk_cand(c, s));
            SelectOut o{d_out_d + (int64_t)q0 * k, nullptr, ws.slots.as<int>() + (int64_t)q0 * k, k, (int)done,
                        multi ? ws.lb_d.as<float>() : nullptr, multi ? ws.lb_k.as<int64_t>() : nullptr, run_if};
            VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nb, n_cand, kk, 1 << 30, nullptr, 0, o, s, ws.mid));
            VDB_TRY(launch_ivf_rows(ws.slots.as<int>() + (int64_t)q0 * k, nb, kk, k, (int)done, probes, n_probe, h->n_lists,
                                    ws.cand_off.as<int64_t>(), h->lay.d_beg(), h->row_ids.as<int>(),
                                    d_out_i + (int64_t)q0 * k, d_out_order ? d_out_order + (int64_t)q0 * k : nullptr, s, run_if));
        }
    }
    return VDB_OK;
}

int vdb_ivf_search_dev(vdb_ivf* h, const float* d_queries, int nq, int k, int n_probe, const uint32_t* d_mask,
                       float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0), "null argument");
    Entry g(h);
    return ivf_search_impl(h, d_queries, nq, k, n_probe, d_mask, d_out_dist, d_out_row, nullptr, pick_stream(stream));
}

int vdb_ivf_search_ordered_dev(vdb_ivf* h, const float* d_queries, int nq, int k, int n_probe, const uint32_t* d_mask,
                               float* d_out_dist, int64_t* d_out_row, int64_t* d_out_order, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0) && d_out_order, "null argument");
    Entry g(h);
    return ivf_search_impl(h, d_queries, nq, k, n_probe, d_mask, d_out_dist, d_out_row, d_out_order, pick_stream(stream));
}

static int ivf_search_host(vdb_ivf* h, const float* queries, int nq, int k, int n_probe, const uint32_t* mask,
                           const vdb_mask* resident, float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(h && (queries || nq == 0), "null argument");
    VDB_REQUIRE(k >= 1, "k must be >= 1, got %d", k);
    Entry g(h);
    hipStream_t s = h->stream;
    const uint32_t* d_mask = nullptr;
    VDB_TRY(resolve_mask(h->ws, mask, resident, h->lay.row_limit, &d_mask, s));
    return with_host_io(h->members->ws, queries, nq, h->dim, k, out_dist, out_row, s,
                        [&](const float* dq, float* dd, int64_t* di) {
                            return ivf_search_impl(h, dq, nq, k, n_probe, d_mask, dd, di, nullptr, s);
                        });
}

int vdb_ivf_search(vdb_ivf* h, const float* queries, int nq, int k, int n_probe, const uint32_t* mask,
                   float* out_dist, int64_t* out_row) {
    return ivf_search_host(h, queries, nq, k, n_probe, mask, nullptr, out_dist, out_row);
}

int vdb_ivf_search_masked(vdb_ivf* h, const float* queries, int nq, int k, int n_probe, const vdb_mask* mask,
                          float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(mask, "null mask");
    return ivf_search_host(h, queries, nq, k, n_probe, nullptr, mask, out_dist, out_row);
}

int vdb_ivf_probe(vdb_ivf* h, const float* queries, int nq, int n_probe, int32_t* out_lists, float* out_dist) {
    VDB_REQUIRE(h && (queries || nq == 0) && h->n_lists >= 1 && n_probe >= 1, "bad arguments to vdb_ivf_probe");
    if (nq <= 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = h->stream;
    Workspace& ws = h->ws;
    n_probe = std::min(n_probe, h->n_lists);
    VDB_TRY(ws.stage.reserve((size_t)nq * h->dim * sizeof(float)));
    VDB_TRY(ws.probes.reserve((size_t)nq * n_probe * sizeof(int)));
    VDB_TRY(ws.probe_d.reserve((size_t)nq * n_probe * sizeof(float)));
    VDB_HIP(hipMemcpyAsync(ws.stage.p, queries, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
    VDB_TRY(ivf_probe_impl(h, nullptr, ws.stage.as<float>(), nq, n_probe, ws.probes.as<int>(), ws.probe_d.as<float>(), s));
    if (out_lists) VDB_HIP(hipMemcpyAsync(out_lists, ws.probes.p, (size_t)nq * n_probe * sizeof(int), hipMemcpyDeviceToHost, s));
    if (out_dist) VDB_HIP(hipMemcpyAsync(out_dist, ws.probe_d.p, (size_t)nq * n_probe * sizeof(float), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_ivf_set_prefilter(vdb_ivf* h, int mode) {
    VDB_REQUIRE(h && mode >= VDB_PREFILTER_NONE && mode <= VDB_PREFILTER_AUTO, "unknown prefilter mode %d", mode);
    Entry g(h);
    h->prefilter = mode;
    if (mode == VDB_PREFILTER_NONE) {
        (void)hipStreamSynchronize(h->stream);
        if (h->last_stream && h->last_stream != h->stream) (void)hipStreamSynchronize(h->last_stream);
        h->shadow16.release();
        h->shadow_cap = 0;
        h->shadow_stale = true;
        h->shadow_dirty.clear();
    }
    return VDB_OK;
}

int vdb_ivf_search_stats(vdb_ivf* h, int* used_matrix_scan, int* overflowed) {
    VDB_REQUIRE(h && used_matrix_scan && overflowed, "null argument");
    Entry g(h);
    *used_matrix_scan = h->last_used_mfma ? 1 : 0;
    *overflowed = 0;
    if (h->last_used_mfma && h->ws.flag.p) {
        hipStream_t s = h->last_stream ? h->last_stream : h->stream;
        VDB_HIP(hipMemcpyAsync(overflowed, h->ws.flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
        VDB_HIP(hipStreamSynchronize(s));
    }
    return VDB_OK;
}

int vdb_ivf_last_scan_rows(vdb_ivf* h, int64_t* rows_scanned) {
    VDB_REQUIRE(h && rows_scanned, "null argument");
    Entry g(h);
    hipStream_t s = h->stream;
    *rows_scanned = 0;
    const size_t cnt = (size_t)h->last_nq * h->last_nprobe;
    if (cnt == 0 || h->ws.probes.cap < cnt * sizeof(int)) return VDB_OK;
    std::vector<int> probes(cnt);
    // the probe lists were written on the stream of that search
    if (h->last_stream && h->last_stream != s) VDB_HIP(hipStreamSynchronize(h->last_stream));
    VDB_HIP(hipMemcpyAsync(probes.data(), h->ws.probes.p, cnt * sizeof(int), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    int64_t total = 0;
    for (int l : probes)
        if (l >= 0 && l < h->n_lists) total += h->lay.len[l];
    *rows_scanned = total;
    return VDB_OK;
}

// ============================================================== PQ
int vdb_pq_encode(const float* x, int64_t n, const float* codebook, int m, int ksub, int dsub, uint8_t* out_codes) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(x && codebook && out_codes && n >= 0 && m >= 1 && ksub >= 1 && ksub <= 256 && dsub >= 1,
                "bad arguments to vdb_pq_encode");
    if (n == 0) return VDB_OK;
    ScopedHandle tmp;
    VDB_TRY(tmp.open());
    hipStream_t s = tmp.stream;
    SumPlan plan;
    VDB_TRY(make_sum_plan(dsub, &plan));
    const int dim = m * dsub;
    DevBuf book, xs, codes;
    auto body = [&]() -> int {
        VDB_TRY(book.reserve((size_t)m * ksub * dsub * sizeof(float)));
        VDB_HIP(hipMemcpyAsync(book.p, codebook, (size_t)m * ksub * dsub * sizeof(float), hipMemcpyHostToDevice, s));
        const int64_t chunk = std::max<int64_t>(1024, (int64_t)(256 << 20) / (dim * (int64_t)sizeof(float)));
        for (int64_t o = 0; o < n; o += chunk) {
            const int64_t c = std::min(chunk, n - o);
            VDB_TRY(xs.reserve((size_t)c * dim * sizeof(float)));
            VDB_TRY(codes.reserve((size_t)c * m));
            VDB_HIP(hipMemcpyAsync(xs.p, x + o * dim, (size_t)c * dim * sizeof(float), hipMemcpyHostToDevice, s));
            VDB_TRY(launch_pq_encode(xs.as<float>(), c, dim, book.as<float>(), m, ksub, dsub, plan, codes.as<uint8_t>(), s));
            VDB_HIP(hipMemcpyAsync(out_codes + o * m, codes.p, (size_t)c * m, hipMemcpyDeviceToHost, s));
            VDB_HIP(hipStreamSynchronize(s));
        }
        return VDB_OK;
    };
    const int st = body();
    book.release(); xs.release(); codes.release();
    return st;
}

int vdb_pq_create(vdb_pq** out, int dim, int m, int ksub) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(out != nullptr, "null output handle");
    VDB_REQUIRE(dim >= 1 && m >= 1 && dim % m == 0, "dimension %d must be divisible by the number of sub-vectors %d", dim, m);
    VDB_REQUIRE(ksub >= 1 && ksub <= 256, "codewords per sub-space must be in 1..256, got %d", ksub);
    vdb_pq* h = new vdb_pq();
    h->dim = dim; h->m = m; h->ksub = ksub; h->dsub = dim / m;
    int st = make_sum_plan(h->dsub, &h->plan);
    if (st == VDB_OK) st = h->open();
    if (st != VDB_OK) { delete h; return st; }
    *out = h;
    return VDB_OK;
}

int vdb_pq_destroy(vdb_pq* h) {
    if (!h) return VDB_OK;
    {
        Entry g(h);
        (void)hipStreamSynchronize(h->stream);
        if (h->last_stream && h->last_stream != h->stream) (void)hipStreamSynchronize(h->last_stream);
        h->book.release(); h->codes.release(); h->ws.release(); h->io.release();
    }
    h->close();
    delete h;
    return VDB_OK;
}

int vdb_pq_set_codebook(vdb_pq* h, const float* codebook) {
    VDB_REQUIRE(h && codebook, "null argument");
    Entry g(h);
    hipStream_t s = h->stream;
    const size_t bytes = (size_t)h->m * h->ksub * h->dsub * sizeof(float);
    VDB_TRY(h->book.reserve(bytes));
    VDB_HIP(hipMemcpyAsync(h->book.p, codebook, bytes, hipMemcpyHostToDevice, s));
    VDB_HIP(hipStreamSynchronize(s));
    h->trained = true;
    return VDB_OK;
}

static int pq_reserve(vdb_pq* h, int64_t n_total, hipStream_t s) {
    if (n_total <= h->cap) return VDB_OK;
    int64_t cap = std::max<int64_t>(n_total, h->cap + h->cap / 2);
    cap = std::max<int64_t>(cap, 4096);
    VDB_TRY(h->codes.reserve_keep((size_t)cap * h->m + 16, (size_t)h->n * h->m, s));
    h->cap = cap;
    return VDB_OK;
}

int vdb_pq_append_codes(vdb_pq* h, const uint8_t* codes, int64_t n) {
    VDB_REQUIRE(h && (codes || n == 0) && n >= 0, "bad arguments to vdb_pq_append_codes");
    if (n == 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = h->stream;
    VDB_TRY(pq_reserve(h, h->n + n, s));
    VDB_HIP(hipMemcpyAsync(h->codes.as<uint8_t>() + h->n * h->m, codes, (size_t)n * h->m, hipMemcpyHostToDevice, s));
    VDB_HIP(hipStreamSynchronize(s));
    h->n += n;
    return VDB_OK;
}

int vdb_pq_append_codes_dev(vdb_pq* h, const uint8_t* d_codes, int64_t n, void* stream) {
    VDB_REQUIRE(h && (d_codes || n == 0) && n >= 0, "bad arguments to vdb_pq_append_codes_dev");
    if (n == 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = pick_stream(stream);
    h->last_stream = s;
    VDB_TRY(pq_reserve(h, h->n + n, s));
    VDB_HIP(hipMemcpyAsync(h->codes.as<uint8_t>() + h->n * h->m, d_codes, (size_t)n * h->m, hipMemcpyDeviceToDevice, s));
    h->n += n;
    return VDB_OK;
}

int vdb_pq_append_rows(vdb_pq* h, const float* rows, int64_t n, uint8_t* out_codes_or_null) {
    VDB_REQUIRE(h && (rows || n == 0) && n >= 0, "bad arguments to vdb_pq_append_rows");
    VDB_REQUIRE(h->trained, "codebook not set");
    if (n == 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = h->stream;
    VDB_TRY(pq_reserve(h, h->n + n, s));
    const int64_t chunk = std::max<int64_t>(1024, (int64_t)(256 << 20) / (h->dim * (int64_t)sizeof(float)));
    for (int64_t o = 0; o < n; o += chunk) {
        const int64_t c = std::min(chunk, n - o);
        VDB_TRY(h->ws.stage.reserve((size_t)c * h->dim * sizeof(float)));
        VDB_HIP(hipMemcpyAsync(h->ws.stage.p, rows + o * h->dim, (size_t)c * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
        uint8_t* dst = h->codes.as<uint8_t>() + (h->n + o) * h->m;
        VDB_TRY(launch_pq_encode(h->ws.stage.as<float>(), c, h->dim, h->book.as<float>(), h->m, h->ksub, h->dsub, h->plan, dst, s));
        if (out_codes_or_null)
            VDB_HIP(hipMemcpyAsync(out_codes_or_null + o * h->m, dst, (size_t)c * h->m, hipMemcpyDeviceToHost, s));
        VDB_HIP(hipStreamSynchronize(s));
    }
    h->n += n;
    return VDB_OK;
}

int vdb_pq_truncate(vdb_pq* h, int64_t n) {
    VDB_REQUIRE(h && n >= 0 && n <= h->n, "cannot truncate to %lld rows", (long long)n);
    Entry g(h);
    h->n = n;
    return VDB_OK;
}

int vdb_pq_set_code(vdb_pq* h, int64_t row, const uint8_t* code) {
    VDB_REQUIRE(h && code && row >= 0 && row < h->n, "row %lld out of range", (long long)row);
    Entry g(h);
    hipStream_t s = h->stream;
    VDB_HIP(hipMemcpyAsync(h->codes.as<uint8_t>() + row * h->m, code, (size_t)h->m, hipMemcpyHostToDevice, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_pq_size(const vdb_pq* h, int64_t* n) {
    VDB_REQUIRE(h && n, "null argument");
    *n = h->n;
    return VDB_OK;
}

int vdb_pq_lut(vdb_pq* h, const float* query, float* out_lut) {
    VDB_REQUIRE(h && query && out_lut, "null argument");
    VDB_REQUIRE(h->trained, "codebook not set");
    Entry g(h);
    hipStream_t s = h->stream;
    Workspace& ws = h->ws;
    const size_t lut_bytes = (size_t)h->m * h->ksub * sizeof(float);
    VDB_TRY(ws.stage.reserve((size_t)h->dim * sizeof(float)));
    VDB_TRY(ws.lut.reserve(lut_bytes));
    VDB_HIP(hipMemcpyAsync(ws.stage.p, query, (size_t)h->dim * sizeof(float), hipMemcpyHostToDevice, s));
    VDB_TRY(launch_pq_lut(ws.stage.as<float>(), h->dim, 1, h->book.as<float>(), h->m, h->ksub, h->dsub, h->plan, ws.lut.as<float>(), s));
    VDB_HIP(hipMemcpyAsync(out_lut, ws.lut.p, lut_bytes, hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

// bound pass of the PQ scan: blocks, tiles of 256 rows per block at most, smallest code set that gets one
static constexpr int PQ_BOUND_SPLITS = 192;
static constexpr int64_t PQ_BOUND_TILES_PER_BLOCK = 8;
static constexpr int64_t PQ_BOUND_MIN_TILES = 256;      // 65 536 codes (measured faster from 100 k codes on, one query or eight)

static int pq_search_impl(vdb_pq* h, const float* d_q, int nq, int k, const uint32_t* d_mask,
                          float* d_out_d, int64_t* d_out_i, hipStream_t s) {
    VDB_REQUIRE(h->trained, "index is not trained");
    VDB_REQUIRE(nq >= 0 && k >= 1, "need nq >= 0 and k >= 1");
    if (nq == 0) return VDB_OK;
    h->last_stream = s;
    Workspace& ws = h->ws;
    const int64_t k_eff = std::min<int64_t>(k, h->n);
    if (k_eff < k) VDB_TRY(fill_results(d_out_d, d_out_i, (int64_t)nq * k, s));
    if (h->n == 0) return VDB_OK;
    VDB_TRY(ws.lut.reserve((size_t)nq * h->m * h->ksub * sizeof(float)));
    VDB_TRY(ws.queue_cnt.reserve((size_t)nq * sizeof(int)));
    int* qlen = ws.queue_cnt.as<int>();          // entries in every query's packed candidate area; the table kernel clears it
    VDB_TRY(ws.round32.reserve((size_t)nq * h->m * sizeof(float)));          // the smallest entry of every table row
    // (the counters and the two flags of the byte-table pass: cleared by the table kernel, the first of the search)
    constexpr int PQ8_Q_BLOCK = 64, PQ8_LEN_STRIDE = 32, PQ8_COUNTERS = PQ8_Q_BLOCK * PQ8_LEN_STRIDE + 2;
    VDB_TRY(ws.cursor.reserve((size_t)PQ8_COUNTERS * sizeof(int)));
    VDB_TRY(launch_pq_lut(d_q, h->dim, nq, h->book.as<float>(), h->m, h->ksub, h->dsub, h->plan, ws.lut.as<float>(), s, qlen,
                          ws.round32.as<float>(), ws.cursor.as<int>(), PQ8_COUNTERS));
    const int64_t ntiles = (h->n + 255) / 256;
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + 15) / 16, (8 * ctx().num_cus + nq - 1) / nq));
    const int n_cand = splits * pq_scan_lists_per_split();
    const int kmax = (int)std::min<int64_t>(KMAX_FUSED, k_eff);
    VDB_TRY(ws.part_d.reserve((size_t)nq * n_cand * kmax * sizeof(float)));
    VDB_TRY(ws.part_i.reserve((size_t)nq * n_cand * kmax * sizeof(int)));
    const bool multi = k_eff > KMAX_FUSED;
    if (multi) {
        VDB_TRY(ws.lb_d.reserve((size_t)nq * sizeof(float)));
        VDB_TRY(ws.lb_k.reserve((size_t)nq * sizeof(int64_t)));
    }
    // Bound pass (one to three queries over a large code set, k results in one pass): a few blocks scan the first
    // rows of the set and every wave reports its best row; the k-th smallest of those (different rows, so k rows
    // are at most that far) is at least the query's k-th best over the whole set, and the full scan drops every row
    // whose running sum passes it (pq_scan_kernel).  Any subset gives a valid bound; a looser one only makes the
    // scan slower, never different.  Behind the bound nearly all wave lists of the full scan stay empty, so the
    // waves pack what they hold into the front of the query's candidate area and one block per query picks the k
    // best.  (Four queries per pass: a lane stays while any of its four sums is in reach, so a bound removes
    // little and its pass costs what it saves; they keep one list per wave and the list merge.)
    const float* bound_d = nullptr;
    static const bool no_bound = getenv("VDB_PQ_NO_BOUND") != nullptr;
    // VDB_PQ_BOUND_MIN_TILES: the smallest code set (in tiles of 256 rows) that gets a bound pass (tests lower it)
    const char* min_tiles_env = getenv("VDB_PQ_BOUND_MIN_TILES");         // read per call: tests change it between searches
    const int64_t bound_min_tiles = min_tiles_env ? atoll(min_tiles_env) : PQ_BOUND_MIN_TILES;
    // Two queries and more: eight queries per pass on one-byte tables behind the same bound, the fp32 sums of the rows
    // that pass recomputed (pq_scan8.hip); the fp32 scan below then runs only behind the overflow flag.
    static const bool no_bytes = getenv("VDB_PQ_NO_BYTE_TABLES") != nullptr;
    const bool bytes = !multi && !no_bound && !no_bytes && !getenv("VDB_PQ_NO_BYTE_TABLES_NOW") && nq >= 2 && pq_scan8_supports(h->m, h->ksub) && ntiles >= bound_min_tiles;
    const bool bounded = !multi && !no_bound && (nq < 4 || bytes) && ntiles >= bound_min_tiles;
    if (bounded) {
        // (the byte-table pass needs the same sample for any batch size: one query per block in its bound pass)
        const int bsplits = (int)std::min<int64_t>(PQ_BOUND_SPLITS, bytes ? (ntiles + 15) / 16 : splits);
        const int64_t n_s = std::min<int64_t>(h->n, PQ_BOUND_TILES_PER_BLOCK * bsplits * 256);
        const int area = bsplits * pq_scan_lists_per_split();
        if (area >= 4 * kmax) {
            VDB_TRY(ws.part_d.reserve((size_t)nq * area * sizeof(float)));
            VDB_TRY(ws.part_i.reserve((size_t)nq * area * sizeof(int)));
            VDB_TRY(launch_pq_scan(h->codes.as<uint8_t>(), n_s, h->m, h->ksub, ws.lut.as<float>(), nq, 1, d_mask, nullptr, nullptr,
                                   bsplits, ws.part_d.as<float>(), ws.part_i.as<int>(), nullptr, s, nullptr, 0, false,
                                   bytes ? nullptr : qlen, nullptr, bytes ? 1 : 4));
            SelectOut o{d_out_d, d_out_i, nullptr, k, 0, nullptr, nullptr};
            // (eight queries and more: every wave's row at its own place instead of an atomic per wave and query)
            if (bytes) VDB_TRY(launch_select_area(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, area, kmax, o, s));
            else VDB_TRY(launch_select_packed(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, area, kmax, o, s, qlen));
            bound_d = d_out_d + (kmax - 1);
        }
    }
    const int* run_if = nullptr;
    int64_t bound_stride = k;
    h->last_used_bytes = false;
    if (bytes && bound_d) {
        constexpr int Q_BLOCK = PQ8_Q_BLOCK, AREA = 32768, SEG_CAP = 512, LEN_STRIDE = PQ8_LEN_STRIDE;
        const int qb = std::min(nq, Q_BLOCK);
        const int bsplits = pq_scan8_splits(h->n, qb);
        const int n_seg = pq_scan8_groups(qb) * bsplits * 4;
        VDB_TRY(ws.q16.reserve(pq_scan8_table_bytes(qb, h->m)));
        VDB_TRY(ws.thr_d.reserve((size_t)nq * sizeof(float)));
        VDB_TRY(ws.mins.reserve((size_t)nq * sizeof(float)));
        VDB_TRY(ws.seg.reserve((size_t)n_seg * SEG_CAP * sizeof(int2)));
        VDB_TRY(ws.seg_len.reserve((size_t)n_seg * sizeof(int)));
        VDB_TRY(ws.queue.reserve((size_t)qb * AREA * (sizeof(float) + sizeof(int))));
        int* counters = ws.cursor.as<int>();
        int* flag = counters + (size_t)Q_BLOCK * LEN_STRIDE;     // some query's rows did not fit: the fp32 scan answers
        int* redo = flag + 1;                                    // some query's candidates are left to the general selection
        for (int q0 = 0; q0 < nq; q0 += Q_BLOCK) {
            const int nb = std::min(Q_BLOCK, nq - q0);
            PqScan8Job j{};
            j.codes = h->codes.as<uint8_t>(); j.n = h->n; j.m = h->m;
            j.lut = ws.lut.as<float>() + (size_t)q0 * h->m * h->ksub; j.nq = nb;
            j.lut_mins = ws.round32.as<float>() + (size_t)q0 * h->m;
            j.bound_d = bound_d + (int64_t)q0 * k; j.bound_stride = k; j.mask = d_mask;
            j.splits = pq_scan8_splits(h->n, nb);
            j.t8 = ws.q16.p; j.thr = ws.thr_d.as<float>() + q0; j.bound_keep = ws.mins.as<float>() + q0;
            j.seg = ws.seg.p; j.seg_len = ws.seg_len.as<int>(); j.seg_cap = SEG_CAP;
            j.area_d = ws.queue.as<float>(); j.area_i = ws.queue.as<int>() + (size_t)qb * AREA; j.area = AREA;
            j.queue_len = counters; j.len_stride = LEN_STRIDE; j.overflow = flag; j.timed = true;
            VDB_TRY(launch_pq_scan8(j, s));
            SelectOut o{d_out_d + (int64_t)q0 * k, d_out_i + (int64_t)q0 * k, nullptr, k, 0, nullptr, nullptr};
            VDB_TRY(launch_select_packed(j.area_d, j.area_i, nb, AREA, kmax, o, s, counters, LEN_STRIDE, 16, redo));
        }
        run_if = flag;
        bound_d = ws.mins.as<float>(); bound_stride = 1;     // the outputs that held the bounds have been written
        h->last_used_bytes = true; h->last_flag = flag;
    }
    for (int64_t done = 0; done < k_eff; done += KMAX_FUSED) {
        const int kk = (int)std::min<int64_t>(KMAX_FUSED, k_eff - done);
        VDB_TRY(launch_pq_scan(h->codes.as<uint8_t>(), h->n, h->m, h->ksub, ws.lut.as<float>(), nq, kk, d_mask,
                               done > 0 ? ws.lb_d.as<float>() : nullptr, done > 0 ? ws.lb_k.as<int64_t>() : nullptr, splits,
                               ws.part_d.as<float>(), ws.part_i.as<int>(), nullptr, s, bound_d, bound_stride, run_if == nullptr,
                               bound_d ? qlen : nullptr, run_if));
        SelectOut o{d_out_d, d_out_i, nullptr, k, (int)done,
                    multi ? ws.lb_d.as<float>() : nullptr, multi ? ws.lb_k.as<int64_t>() : nullptr, run_if};
        if (bound_d) VDB_TRY(launch_select_packed(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, n_cand * kk, kk, o, s, qlen));
        else VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nq, n_cand, kk, 1 << 30, nullptr, 0, o, s, ws.mid));
    }
    return VDB_OK;
}

int vdb_pq_search_dev(vdb_pq* h, const float* d_queries, int nq, int k, const uint32_t* d_mask,
                      float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0), "null argument");
    Entry g(h);
    return pq_search_impl(h, d_queries, nq, k, d_mask, d_out_dist, d_out_row, pick_stream(stream));
}

static int pq_search_host(vdb_pq* h, const float* queries, int nq, int k, const uint32_t* mask, const vdb_mask* resident,
                          float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(h && (queries || nq == 0), "null argument");
    VDB_REQUIRE(k >= 1, "k must be >= 1, got %d", k);
    Entry g(h);
    hipStream_t s = h->stream;
    const uint32_t* d_mask = nullptr;
    VDB_TRY(resolve_mask(h->ws, mask, resident, h->n, &d_mask, s));
    return with_host_io(h->io, queries, nq, h->dim, k, out_dist, out_row, s,
                        [&](const float* dq, float* dd, int64_t* di) {
                            return pq_search_impl(h, dq, nq, k, d_mask, dd, di, s);
                        });
}

int vdb_pq_search_stats(vdb_pq* h, int* used_byte_tables, int* overflowed) {
    VDB_REQUIRE(h && used_byte_tables && overflowed, "null argument");
    Entry g(h);
    *used_byte_tables = h->last_used_bytes ? 1 : 0;
    *overflowed = 0;
    if (h->last_used_bytes && h->last_flag) {
        hipStream_t s = h->last_stream ? h->last_stream : h->stream;
        VDB_HIP(hipStreamSynchronize(s));
        int f = 0;
        VDB_HIP(hipMemcpy(&f, h->last_flag, sizeof(int), hipMemcpyDeviceToHost));
        *overflowed = f != 0;
    }
    return VDB_OK;
}

int vdb_pq_search(vdb_pq* h, const float* queries, int nq, int k, const uint32_t* mask, float* out_dist, int64_t* out_row) {
    return pq_search_host(h, queries, nq, k, mask, nullptr, out_dist, out_row);
}

int vdb_pq_search_masked(vdb_pq* h, const float* queries, int nq, int k, const vdb_mask* mask, float* out_dist,
                         int64_t* out_row) {
    VDB_REQUIRE(mask, "null mask");
    return pq_search_host(h, queries, nq, k, nullptr, mask, out_dist, out_row);
}

// ============================================================== IVF-PQ (index/hybrid.py)
struct vdb_ivfpq : Handle {
    int dim = 0, m = 0, ksub = 0, dsub = 0;
    SumPlan plan;                       // over one sub-vector
    vdb_rows* centroids = nullptr;      // coarse centroids
    DevBuf book;                        // float [m][ksub][dsub]
    DevBuf codes;                       // uint8 [positions][m], every list at the positions `lay` names
    ListLayout lay;
    DevBuf row_ids;                     // int   [positions]
    DevBuf app_codes, app_dst, app_no;  // staging of appended codes
    int n_lists = 0;
    bool ready = false;
    hipStream_t last_stream = nullptr;  // stream of the last *_dev call
    Workspace ws, io;
    int64_t probe_bound(int n_probe) { return lay.probe_bound(n_probe); }
};

int vdb_ivfpq_create(vdb_ivfpq** out, int dim, int m, int ksub) {
    VDB_TRY(ensure_init());
    VDB_REQUIRE(out != nullptr, "null output handle");
    VDB_REQUIRE(dim >= 1 && m >= 1 && dim % m == 0, "dimension %d must be divisible by the number of sub-spaces %d", dim, m);
    VDB_REQUIRE(ksub >= 1 && ksub <= 256, "codewords per sub-space must be in 1..256, got %d", ksub);
    vdb_ivfpq* h = new vdb_ivfpq();
    h->dim = dim; h->m = m; h->ksub = ksub; h->dsub = dim / m;
    int st = make_sum_plan(h->dsub, &h->plan);
    if (st == VDB_OK) st = vdb_rows_create(&h->centroids, dim);
    if (st == VDB_OK) st = h->open();
    if (st != VDB_OK) { vdb_rows_destroy(h->centroids); delete h; return st; }
    *out = h;
    return VDB_OK;
}

int vdb_ivfpq_destroy(vdb_ivfpq* h) {
    if (!h) return VDB_OK;
    {
        Entry g(h);
        (void)hipStreamSynchronize(h->stream);
        if (h->last_stream && h->last_stream != h->stream) (void)hipStreamSynchronize(h->last_stream);
        h->book.release(); h->codes.release(); h->lay.dev.release(); h->row_ids.release(); h->ws.release(); h->io.release();
        h->app_codes.release(); h->app_dst.release(); h->app_no.release();
    }
    vdb_rows_destroy(h->centroids);
    h->close();
    delete h;
    return VDB_OK;
}

int vdb_ivfpq_set_quantizers(vdb_ivfpq* h, const float* centroids, int n_lists, const float* codebook) {
    VDB_REQUIRE(h && centroids && codebook && n_lists >= 1, "bad arguments to vdb_ivfpq_set_quantizers");
    Entry g(h);
    VDB_TRY(vdb_rows_truncate(h->centroids, 0));
    VDB_TRY(vdb_rows_append(h->centroids, centroids, n_lists));
    hipStream_t s = h->stream;
    const size_t bytes = (size_t)h->m * h->ksub * h->dsub * sizeof(float);
    VDB_TRY(h->book.reserve(bytes));
    VDB_HIP(hipMemcpyAsync(h->book.p, codebook, bytes, hipMemcpyHostToDevice, s));
    VDB_HIP(hipStreamSynchronize(s));
    h->n_lists = n_lists;
    h->lay.reset(n_lists);
    h->ready = true;
    return VDB_OK;
}

int vdb_ivfpq_encode(vdb_ivfpq* h, const float* rows, int64_t n, int32_t* out_list, uint8_t* out_codes) {
    VDB_REQUIRE(h && h->ready && (rows || n == 0) && n >= 0 && out_list && out_codes, "bad arguments to vdb_ivfpq_encode");
    if (n == 0) return VDB_OK;
    Entry g(h);
    hipStream_t s = h->stream;
    Workspace& ws = h->ws;
    vdb_rows* c = h->centroids;
    DevBuf i32, res, codes;
    auto body = [&]() -> int {
        const int64_t chunk = std::max<int64_t>(1024, (int64_t)(128 << 20) / (h->dim * (int64_t)sizeof(float)));
        for (int64_t o = 0; o < n; o += chunk) {
            const int nb = (int)std::min(chunk, n - o);
            VDB_TRY(ws.stage.reserve((size_t)nb * h->dim * sizeof(float)));
            VDB_TRY(ws.out_d.reserve((size_t)nb * sizeof(float)));
            VDB_TRY(i32.reserve((size_t)nb * sizeof(int)));
            VDB_TRY(res.reserve((size_t)nb * h->dim * sizeof(float)));
            VDB_TRY(codes.reserve((size_t)nb * h->m));
            VDB_HIP(hipMemcpyAsync(ws.stage.p, rows + o * h->dim, (size_t)nb * h->dim * sizeof(float), hipMemcpyHostToDevice, s));
            // nearest coarse centroid in the identity-form squared distance (hybrid.py:205-210)
            VDB_TRY(flat_search_impl(c, VDB_EUCLIDEAN_SQ, ws.stage.as<float>(), nb, 1, nullptr, ws.out_d.as<float>(), nullptr,
                                     i32.as<int>(), s));
            VDB_TRY(launch_residual(ws.stage.as<float>(), h->dim, c->r.data, c->r.ld, i32.as<int>(), h->dim, nb, res.as<float>(), s));
            VDB_TRY(launch_pq_encode(res.as<float>(), nb, h->dim, h->book.as<float>(), h->m, h->ksub, h->dsub, h->plan,
                                     codes.as<uint8_t>(), s));
            VDB_HIP(hipMemcpyAsync(out_list + o, i32.p, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, s));
            VDB_HIP(hipMemcpyAsync(out_codes + o * h->m, codes.p, (size_t)nb * h->m, hipMemcpyDeviceToHost, s));
            VDB_HIP(hipStreamSynchronize(s));
        }
        return VDB_OK;
    };
    const int st = body();
    i32.release(); res.release(); codes.release();
    return st;
}

// codes[gather[i]] becomes the code at position i and answers to the row number numbers[i]
static int ivfpq_build_impl(vdb_ivfpq* h, const uint8_t* codes, int64_t n, const int64_t* gather, const int64_t* numbers,
                            const int64_t* list_offsets) {
    VDB_REQUIRE(h && h->ready, "quantizers must be set before the lists are built");
    VDB_REQUIRE(n >= 0 && (n == 0 || (codes && numbers)) && list_offsets, "bad arguments to the list build");
    VDB_REQUIRE(list_offsets[0] == 0 && list_offsets[h->n_lists] == n, "list offsets do not cover %lld rows", (long long)n);
    VDB_REQUIRE(n < (int64_t)0x7fffffff, "at most 2^31-1 rows per device");
    std::vector<int> nums((size_t)n);
    std::vector<uint8_t> ordered((size_t)n * h->m);
    for (int l = 0; l < h->n_lists; ++l)
        VDB_REQUIRE(list_offsets[l] <= list_offsets[l + 1], "list offsets must be non-decreasing");
    int64_t row_limit = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t g = gather ? gather[i] : i;
        VDB_REQUIRE(g >= 0 && g < n, "row %lld out of range", (long long)g);
        VDB_REQUIRE(numbers[i] >= 0 && numbers[i] < (int64_t)0x7fffffff, "row number %lld out of range", (long long)numbers[i]);
        nums[(size_t)i] = (int)numbers[i];
        row_limit = std::max(row_limit, numbers[i] + 1);
        memcpy(&ordered[(size_t)i * h->m], codes + g * h->m, h->m);
    }
    Entry g(h);
    hipStream_t s = h->stream;
    h->lay.set_packed(list_offsets);
    h->lay.row_limit = row_limit;
    VDB_TRY(h->lay.sync(s));
    if (n > 0) {
        VDB_TRY(h->row_ids.reserve((size_t)n * sizeof(int)));
        VDB_TRY(h->codes.reserve((size_t)n * h->m));
        VDB_HIP(hipMemcpyAsync(h->row_ids.p, nums.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
        VDB_HIP(hipMemcpyAsync(h->codes.p, ordered.data(), (size_t)n * h->m, hipMemcpyHostToDevice, s));
    }
    VDB_HIP(hipStreamSynchronize(s));
    return VDB_OK;
}

int vdb_ivfpq_build(vdb_ivfpq* h, const uint8_t* codes, int64_t n, const int64_t* list_rows, const int64_t* list_offsets) {
    VDB_REQUIRE(n == 0 || list_rows, "bad arguments to vdb_ivfpq_build");
    return ivfpq_build_impl(h, codes, n, list_rows, list_rows, list_offsets);
}

int vdb_ivfpq_build_numbered(vdb_ivfpq* h, const uint8_t* codes, int64_t n, const int64_t* row_numbers, const int64_t* list_offsets) {
    return ivfpq_build_impl(h, codes, n, nullptr, row_numbers, list_offsets);
}

__global__ void ivfpq_scatter_codes_kernel(const uint8_t* __restrict__ src, int m, int64_t count, const int64_t* __restrict__ dst,
                                           const int* __restrict__ numbers, uint8_t* __restrict__ codes, int* __restrict__ row_ids) {
    const int64_t total = count * m;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / m;
        const int c = (int)(t - i * m);
        const int64_t d = dst[i];
        codes[d * m + c] = src[t];
        if (c == 0) row_ids[d] = numbers[i];
    }
}

__global__ void ivfpq_move_codes_kernel(uint8_t* __restrict__ codes, int* __restrict__ row_ids, int m, int64_t from, int64_t to,
                                        int64_t count) {
    const int64_t total = count * m;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        codes[to * m + t] = codes[from * m + t];
        if (t < count) row_ids[to + t] = row_ids[from + t];
    }
}

int vdb_ivfpq_append(vdb_ivfpq* h, const uint8_t* codes, const int64_t* row_numbers, const int32_t* lists, int64_t count) {
    VDB_REQUIRE(h && h->ready, "quantizers must be set before rows are appended");
    VDB_REQUIRE(count >= 0 && (count == 0 || (codes && row_numbers && lists)), "bad arguments to vdb_ivfpq_append");
    if (count == 0) return VDB_OK;
    std::vector<int> nums((size_t)count);
    int64_t row_limit = h->lay.row_limit;
    for (int64_t i = 0; i < count; ++i) {
        VDB_REQUIRE(row_numbers[i] >= 0 && row_numbers[i] < (int64_t)0x7fffffff, "row number %lld out of range", (long long)row_numbers[i]);
        nums[(size_t)i] = (int)row_numbers[i];
        row_limit = std::max(row_limit, row_numbers[i] + 1);
    }
    Entry g(h);
    hipStream_t s = h->stream;
    ListLayout& lay = h->lay;
    std::vector<int64_t> dst;
    std::vector<ListMove> moves;
    const int64_t old_tail = lay.tail;
    ListLayout::Undo undo;
    VDB_TRY(lay.plan_append(lists, count, dst, moves, undo));
    // every allocation before the first code is written; a failure takes the plan back
    const int reserved = [&]() -> int {
        VDB_REQUIRE(lay.tail < (int64_t)0x7fffffff, "at most 2^31-1 positions per device");
        VDB_TRY(h->codes.reserve_keep((size_t)lay.tail * h->m, (size_t)old_tail * h->m, s));
        VDB_TRY(h->row_ids.reserve_keep((size_t)lay.tail * sizeof(int), (size_t)old_tail * sizeof(int), s));
        VDB_TRY(h->app_codes.reserve((size_t)count * h->m));
        VDB_TRY(h->app_dst.reserve((size_t)count * sizeof(int64_t)));
        VDB_TRY(h->app_no.reserve((size_t)count * sizeof(int)));
        return VDB_OK;
    }();
    if (reserved != VDB_OK) { lay.rollback(undo); return reserved; }
    lay.row_limit = row_limit;
    for (const ListMove& mv : moves)
        hipLaunchKernelGGL(ivfpq_move_codes_kernel, dim3(grid_for(mv.count * h->m)), dim3(256), 0, s, h->codes.as<uint8_t>(),
                           h->row_ids.as<int>(), h->m, mv.from, mv.to, mv.count);
    VDB_HIP(hipGetLastError());
    VDB_HIP(hipMemcpyAsync(h->app_codes.p, codes, (size_t)count * h->m, hipMemcpyHostToDevice, s));
    VDB_HIP(hipMemcpyAsync(h->app_dst.p, dst.data(), (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice, s));
    VDB_HIP(hipMemcpyAsync(h->app_no.p, nums.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ivfpq_scatter_codes_kernel, dim3(grid_for(count * h->m)), dim3(256), 0, s, h->app_codes.as<uint8_t>(), h->m,
                       count, h->app_dst.as<int64_t>(), h->app_no.as<int>(), h->codes.as<uint8_t>(), h->row_ids.as<int>());
    VDB_HIP(hipGetLastError());
    VDB_HIP(hipStreamSynchronize(s));          // the host vectors above are pageable
    return VDB_OK;
}

int vdb_ivfpq_remove_at(vdb_ivfpq* h, int32_t list, int64_t position) {
    VDB_REQUIRE(h && list >= 0 && list < h->n_lists, "bad list number");
    Entry g(h);
    ListLayout& lay = h->lay;
    VDB_REQUIRE(position >= 0 && position < lay.len[list], "position %lld is not in list %d (%lld rows)", (long long)position, list,
                (long long)lay.len[list]);
    // the codes behind the removed one close the gap in their order (index/hybrid.py:396-400 filters the list)
    const int64_t behind = lay.len[list] - 1 - position;
    if (behind > 0) {
        hipStream_t s = h->stream;
        const int64_t at = lay.beg[list] + position;
        VDB_TRY(h->app_codes.reserve((size_t)behind * h->m));
        VDB_TRY(h->app_no.reserve((size_t)behind * sizeof(int)));
        uint8_t* codes = h->codes.as<uint8_t>();
        int* ids = h->row_ids.as<int>();
        VDB_HIP(hipMemcpyAsync(h->app_codes.p, codes + (at + 1) * h->m, (size_t)behind * h->m, hipMemcpyDeviceToDevice, s));
        VDB_HIP(hipMemcpyAsync(h->app_no.p, ids + at + 1, (size_t)behind * sizeof(int), hipMemcpyDeviceToDevice, s));
        VDB_HIP(hipMemcpyAsync(codes + at * h->m, h->app_codes.p, (size_t)behind * h->m, hipMemcpyDeviceToDevice, s));
        VDB_HIP(hipMemcpyAsync(ids + at, h->app_no.p, (size_t)behind * sizeof(int), hipMemcpyDeviceToDevice, s));
    }
    --lay.len[list];
    --lay.live;
    lay.dev_stale = true;
    return VDB_OK;
}

int vdb_ivfpq_layout(vdb_ivfpq* h, int64_t* live_rows, int64_t* positions, int64_t* dead_positions) {
    VDB_REQUIRE(h, "null handle");
    Entry g(h);
    if (live_rows) *live_rows = h->lay.live;
    if (positions) *positions = h->lay.tail;
    if (dead_positions) *dead_positions = h->lay.dead;
    return VDB_OK;
}

static int ivfpq_search_impl(vdb_ivfpq* h, const float* d_q, int nq, int k, int n_probe, const uint32_t* d_mask,
                             float* d_out_d, int64_t* d_out_i, hipStream_t s) {
    VDB_REQUIRE(h->ready, "index is not trained");
    VDB_REQUIRE(nq >= 0 && k >= 1 && n_probe >= 1, "need nq >= 0, k >= 1, n_probe >= 1");
    if (nq == 0) return VDB_OK;
    h->last_stream = s;
    n_probe = std::min(n_probe, h->n_lists);
    Workspace& ws = h->ws;
    VDB_TRY(fill_results(d_out_d, d_out_i, (int64_t)nq * k, s));
    if (h->lay.live == 0) return VDB_OK;
    VDB_TRY(h->lay.sync(s));
    const int64_t k_eff = std::min<int64_t>(k, h->lay.live);
    const int kmax = (int)std::min<int64_t>(KMAX_FUSED, k_eff);
    const bool multi = k_eff > KMAX_FUSED;
    const int64_t cand_stride = round_up64(std::max<int64_t>(h->probe_bound(n_probe), 1), 64);
    VDB_REQUIRE(cand_stride < (int64_t)0x7fffffff, "n_probe=%d lists can hold more than 2^31 rows", n_probe);
    const int q_block = (int)std::max<int64_t>(1, std::min<int64_t>(nq, CAND_BUDGET / (cand_stride * (int64_t)sizeof(float))));

    // coarse ranking: identity-form squared distances to the centroids (hybrid.py:296-300)
    VDB_TRY(ws.probes.reserve((size_t)nq * n_probe * sizeof(int)));
    VDB_TRY(ws.probe_d.reserve((size_t)nq * n_probe * sizeof(float)));
    const bool few_small = nq <= 8 && h->n_lists <= 1024;      // as in ivf_probe_impl
    static const bool no_rank_rows = getenv("VDB_NO_RANK_ROWS") != nullptr;
    if (!no_rank_rows && n_probe <= 64 && (int64_t)nq * h->n_lists <= ((int64_t)1 << 28)) {
        // the whole distance matrix, then one block per query that ranks its row (the search's fused top-32 lists
        // cost 0.32 ms per 1024 x 4096, this 0.04; for one query over 4096 lists 52 us against 18)
        VDB_TRY(ws.dist.reserve((size_t)nq * h->n_lists * sizeof(float)));
        VDB_TRY(flat_distances_impl(h->centroids, VDB_EUCLIDEAN_SQ, d_q, nq, ws.dist.as<float>(), s));
        VDB_TRY(launch_rank_rows(ws.dist.as<float>(), nq, h->n_lists, n_probe, ws.probe_d.as<float>(), ws.probes.as<int>(), s));
    } else if (n_probe <= KMAX_FUSED && !few_small) {
        VDB_TRY(flat_search_impl(h->centroids, VDB_EUCLIDEAN_SQ, d_q, nq, n_probe, nullptr, ws.probe_d.as<float>(), nullptr,
                                 ws.probes.as<int>(), s));
    } else {
        VDB_REQUIRE(h->n_lists <= 16384, "n_probe > %d needs n_lists <= 16384", KMAX_FUSED);
        VDB_TRY(ws.dist.reserve((size_t)nq * h->n_lists * sizeof(float)));
        VDB_TRY(flat_distances_impl(h->centroids, VDB_EUCLIDEAN_SQ, d_q, nq, ws.dist.as<float>(), s));
        VDB_TRY(launch_row_sort_topk(ws.dist.as<float>(), nq, h->n_lists, n_probe, ws.probe_d.as<float>(), ws.probes.as<int>(), s));
    }

    VDB_TRY(ws.cand.reserve((size_t)q_block * cand_stride * sizeof(float)));
    VDB_TRY(ws.cand_off.reserve((size_t)q_block * (n_probe + 1) * sizeof(int64_t)));
    VDB_TRY(ws.tile_off.reserve((size_t)q_block * (n_probe + 1) * sizeof(int)));
    VDB_TRY(ws.slots.reserve((size_t)nq * k * sizeof(int)));
    if (multi) {
        VDB_TRY(ws.lb_d.reserve((size_t)q_block * sizeof(float)));
        VDB_TRY(ws.lb_k.reserve((size_t)q_block * sizeof(int64_t)));
    }
    const int64_t avg_tiles = std::max<int64_t>(1, ((int64_t)n_probe * (h->lay.live / std::max(1, h->n_lists)) + 255) / 256);
    for (int q0 = 0; q0 < nq; q0 += q_block) {
        const int nb = std::min(q_block, nq - q0);
        const int* probes = ws.probes.as<int>() + (int64_t)q0 * n_probe;
        static const bool query_order = getenv("VDB_IVFPQ_QUERY_ORDER") != nullptr;
        IvfPqUnits units{};
        const bool by_list = !query_order && ivfpq_scan_takes_units(h->m, h->ksub, h->dsub, h->dim, nb, n_probe);
        if (by_list) {
            // a batch for the two-table scan: its (query, list) units in list order, so that the queries that probe a
            // list follow each other on one CU and the list's codes are read once for them
            StreamPlanBufs b{};
            VDB_TRY(reserve_stream_plan(ws, h->n_lists, (int64_t)nb * n_probe, 1, &b));
            VDB_TRY(ws.list_cnt.reserve((size_t)h->n_lists * sizeof(int)));
            VDB_TRY(ws.cursor.reserve((size_t)h->n_lists * sizeof(int)));
            VDB_TRY(ws.slot_list.reserve((size_t)nb * n_probe * sizeof(int)));
            VDB_TRY(ws.unit_first.reserve((size_t)(ivfpq_scan_blocks() + 1) * sizeof(int)));
            VDB_TRY(launch_ivf_units_by_list(probes, nb, n_probe, h->lay.d_beg(), h->lay.d_end(), h->n_lists, cand_stride,
                                             ws.list_cnt.as<int>(), ws.cursor.as<int>(), ws.cand_off.as<int64_t>(),
                                             ws.tile_off.as<int>(), b, ws.slot_list.as<int>(), ivfpq_scan_blocks(),
                                             ws.unit_first.as<int>(), s));
            units = IvfPqUnits{ws.slot_list.as<int>(), b.sq_qid, b.sq_out, ws.unit_first.as<int>()};
        } else {
            VDB_TRY(launch_ivf_slot_offsets(probes, nb, n_probe, h->lay.d_beg(), h->lay.d_end(), h->n_lists,
                                            ws.cand_off.as<int64_t>(), ws.tile_off.as<int>(), s));
        }
        VDB_TRY(launch_ivfpq_scan(d_q + (int64_t)q0 * h->dim, h->dim, h->dim, nb, h->centroids->r.data, h->centroids->r.ld,
                                  h->book.as<float>(), h->m, h->ksub, h->dsub, h->plan, h->codes.as<uint8_t>(),
                                  h->lay.d_beg(), h->lay.d_end(), h->n_lists, probes, n_probe, ws.cand_off.as<int64_t>(),
                                  cand_stride, ws.cand.as<float>(), s, by_list ? &units : nullptr));
        static const bool no_cand_rank = getenv("VDB_NO_CAND_RANK") != nullptr;
        if (!no_cand_rank && !multi && !d_mask) {
            // no filter: one block per query reads its run of distances once and ranks the few in reach; a few queries
            // share their runs out over several blocks each, whose lists are merged
            const int parts = nb >= 32 ? 1 : std::max(2, std::min(64, 1024 / nb));     // (blocks of four waves)
            if (parts > 1) {
                VDB_TRY(ws.part_d.reserve((size_t)nb * parts * kmax * sizeof(float)));
                VDB_TRY(ws.part_i.reserve((size_t)nb * parts * kmax * sizeof(int)));
            }
            VDB_TRY(launch_cand_rank(ws.cand.as<float>(), cand_stride, ws.cand_off.as<int64_t>(), n_probe, nb, kmax, k,
                                     d_out_d + (int64_t)q0 * k, ws.slots.as<int>() + (int64_t)q0 * k, s, parts,
                                     ws.part_d.as<float>(), ws.part_i.as<int>(), true));
            if (parts > 1) {
                SelectOut o{d_out_d + (int64_t)q0 * k, nullptr, ws.slots.as<int>() + (int64_t)q0 * k, k, 0, nullptr, nullptr};
                VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nb, parts, kmax, 1 << 30, nullptr, 0, o, s, ws.mid));
            }
            VDB_TRY(launch_ivf_rows(ws.slots.as<int>() + (int64_t)q0 * k, nb, kmax, k, 0, probes, n_probe, h->n_lists,
                                    ws.cand_off.as<int64_t>(), h->lay.d_beg(), h->row_ids.as<int>(),
                                    d_out_i + (int64_t)q0 * k, nullptr, s));
            continue;
        }
        const int splits = cand_splits(avg_tiles, nb);
        const int n_cand = splits * topk_cand_lists_per_split();
        VDB_TRY(ws.part_d.reserve((size_t)nb * n_cand * kmax * sizeof(float)));
        VDB_TRY(ws.part_i.reserve((size_t)nb * n_cand * kmax * sizeof(int)));
        for (int64_t done = 0; done < k_eff; done += KMAX_FUSED) {
            const int kk = (int)std::min<int64_t>(KMAX_FUSED, k_eff - done);
            CandJob c{};
            c.cand = ws.cand.as<float>(); c.cand_stride = cand_stride; c.nq = nb; c.n = 0;
            c.probes = probes; c.n_probe = n_probe; c.n_lists = h->n_lists; c.list_beg = h->lay.d_beg(); c.list_end = h->lay.d_end();
            c.cand_off = ws.cand_off.as<int64_t>(); c.tile_off = ws.tile_off.as<int>();
            c.row_ids = h->row_ids.as<int>(); c.mask = d_mask;
            c.metric = VDB_EUCLIDEAN_DIRECT;          // the slots hold the squares of the distances
            c.sqrt_here = true;
            c.lb_d = done > 0 ? ws.lb_d.as<float>() : nullptr; c.lb_k = done > 0 ? ws.lb_k.as<int64_t>() : nullptr;
            c.k = kk; c.splits = splits; c.part_d = ws.part_d.as<float>(); c.part_i = ws.part_i.as<int>();
            VDB_TRY(launch_topk_cand(c, s));
            SelectOut o{d_out_d + (int64_t)q0 * k, nullptr, ws.slots.as<int>() + (int64_t)q0 * k, k, (int)done,
                        multi ? ws.lb_d.as<float>() : nullptr, multi ? ws.lb_k.as<int64_t>() : nullptr};
            VDB_TRY(launch_select(ws.part_d.as<float>(), ws.part_i.as<int>(), nb, n_cand, kk, 1 << 30, nullptr, 0, o, s, ws.mid));
            VDB_TRY(launch_ivf_rows(ws.slots.as<int>() + (int64_t)q0 * k, nb, kk, k, (int)done, probes, n_probe, h->n_lists,
                                    ws.cand_off.as<int64_t>(), h->lay.d_beg(), h->row_ids.as<int>(),
                                    d_out_i + (int64_t)q0 * k, nullptr, s));
        }
    }
    return VDB_OK;
}

int vdb_ivfpq_search_dev(vdb_ivfpq* h, const float* d_queries, int nq, int k, int n_probe, const uint32_t* d_mask,
                         float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_REQUIRE(h && (d_queries || nq == 0), "null argument");
    Entry g(h);
    return ivfpq_search_impl(h, d_queries, nq, k, n_probe, d_mask, d_out_dist, d_out_row, pick_stream(stream));
}

static int ivfpq_search_host(vdb_ivfpq* h, const float* queries, int nq, int k, int n_probe, const uint32_t* mask,
                             const vdb_mask* resident, float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(h && (queries || nq == 0), "null argument");
    VDB_REQUIRE(k >= 1, "k must be >= 1, got %d", k);
    Entry g(h);
    hipStream_t s = h->stream;
    const uint32_t* d_mask = nullptr;
    VDB_TRY(resolve_mask(h->ws, mask, resident, h->lay.row_limit, &d_mask, s));
    return with_host_io(h->io, queries, nq, h->dim, k, out_dist, out_row, s,
                        [&](const float* dq, float* dd, int64_t* di) {
                            return ivfpq_search_impl(h, dq, nq, k, n_probe, d_mask, dd, di, s);
                        });
}

int vdb_ivfpq_search(vdb_ivfpq* h, const float* queries, int nq, int k, int n_probe, const uint32_t* mask,
                     float* out_dist, int64_t* out_row) {
    return ivfpq_search_host(h, queries, nq, k, n_probe, mask, nullptr, out_dist, out_row);
}

int vdb_ivfpq_search_masked(vdb_ivfpq* h, const float* queries, int nq, int k, int n_probe, const vdb_mask* mask,
                            float* out_dist, int64_t* out_row) {
    VDB_REQUIRE(mask, "null mask");
    return ivfpq_search_host(h, queries, nq, k, n_probe, nullptr, mask, out_dist, out_row);
}

// ============================================================== cross-shard merge
int vdb_topk_merge_dev(const float* d_dist, const int64_t* d_row, int n_lists, int nq, int k,
                       float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_TRY(ensure_init());
    (void)hipSetDevice(ctx().device);
    VDB_REQUIRE(d_dist && d_row && d_out_dist && d_out_row && n_lists >= 1 && nq >= 0 && k >= 1,
                "bad arguments to vdb_topk_merge_dev");
    return launch_select_i64(d_dist, d_row, nullptr, (int64_t)nq * k, (int64_t)nq * k, nullptr, n_lists, nq, k,
                             d_out_dist, d_out_row, pick_stream(stream));
}

int vdb_topk_merge_strided_dev(const float* d_dist, const int64_t* d_row, int64_t list_stride_dist, int64_t list_stride_row,
                               const int64_t* d_row_base, int n_lists, int nq, int k,
                               float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_TRY(ensure_init());
    (void)hipSetDevice(ctx().device);
    VDB_REQUIRE(d_dist && d_row && d_out_dist && d_out_row && n_lists >= 1 && nq >= 0 && k >= 1,
                "bad arguments to vdb_topk_merge_strided_dev");
    VDB_REQUIRE(list_stride_dist >= (int64_t)nq * k && list_stride_row >= (int64_t)nq * k,
                "list strides (%lld, %lld) are shorter than one list of %lld entries", (long long)list_stride_dist,
                (long long)list_stride_row, (long long)nq * k);
    return launch_select_i64(d_dist, d_row, nullptr, list_stride_dist, list_stride_row, d_row_base, n_lists, nq, k,
                             d_out_dist, d_out_row, pick_stream(stream));
}

int vdb_topk_merge_keyed_dev(const float* d_dist, const int64_t* d_key, const int64_t* d_row, int n_lists, int nq, int k,
                             float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_TRY(ensure_init());
    (void)hipSetDevice(ctx().device);
    VDB_REQUIRE(d_dist && d_key && d_row && d_out_dist && d_out_row && n_lists >= 1 && nq >= 0 && k >= 1,
                "bad arguments to vdb_topk_merge_keyed_dev");
    return launch_select_i64(d_dist, d_key, d_row, (int64_t)nq * k, (int64_t)nq * k, nullptr, n_lists, nq, k,
                             d_out_dist, d_out_row, pick_stream(stream));
}

// the keyed merge over lists that lie `list_stride_*` elements apart (blocks of one packed all-gather)
int vdb_topk_merge_keyed_strided_dev(const float* d_dist, const int64_t* d_key, const int64_t* d_row, int64_t list_stride_dist,
                                     int64_t list_stride_key_row, int n_lists, int nq, int k,
                                     float* d_out_dist, int64_t* d_out_row, void* stream) {
    VDB_TRY(ensure_init());
    (void)hipSetDevice(ctx().device);
    VDB_REQUIRE(d_dist && d_key && d_row && d_out_dist && d_out_row && n_lists >= 1 && nq >= 0 && k >= 1,
                "bad arguments to vdb_topk_merge_keyed_strided_dev");
    VDB_REQUIRE(list_stride_dist >= (int64_t)nq * k && list_stride_key_row >= (int64_t)nq * k,
                "list strides (%lld, %lld) are shorter than one list of %lld entries", (long long)list_stride_dist,
                (long long)list_stride_key_row, (long long)nq * k);
    return launch_select_i64(d_dist, d_key, d_row, list_stride_dist, list_stride_key_row, nullptr, n_lists, nq, k,
                             d_out_dist, d_out_row, pick_stream(stream));
}

}  // extern "C"
