// Streaming distance scan: one thread per database row, the row tile staged
// through LDS so that global reads stay coalesced.  Bandwidth-bound work:
//   - Manhattan / Chebyshev (distance/metrics.py:486-490), which have no GEMM form;
//   - single queries and small batches of any metric (FlatIndex.search,
//     index/flat.py:331; Collection.search, core/collection.py:750);
//   - the direct-form per-pair distances of the IVF search
//     (index/ivf.py:611-613 over centroids, :619-628 over the probed lists);
//   - row norms at insertion time (metrics.py:456,470).
//
// Direct-form sums follow NumPy's pairwise order exactly (SumPlan), one
// thread walking its own row left to right, so Euclidean (direct), Manhattan
// and the squared norms are bit-identical to the reference.  Dot-product
// metrics only need rounding-level agreement and use fma.
//
// Every thread keeps its k best (distance, position) in an LDS list; at the
// end the block reduces its 256 lists to one list of k.
#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

static constexpr int NT = LIST_NT;          // threads = rows per tile
static constexpr int CK = 32;           // floats of every row staged per step
static constexpr int LW = CK + 4;       // padded LDS row width
static constexpr int MAXSTACK = 8;

enum { METRIC_SQNORM = 100 };

struct ScanParams {
    const float* data; int ld; int dim;
    const float* row_sq; const float* row_nrm;
    SumPlan plan;
    int metric;
    const float* q; int ldq; int nq;
    const float* q_sq; const float* q_nrm;
    int64_t n;
    const int64_t* list_off; const int* probes; int n_probe;
    const int* row_ids;
    const uint32_t* mask;
    const float* lb_d; const int64_t* lb_k;
    int k; int splits; int n_cand;
    float* part_d; int* part_i;
    float* dist_out; float* nrm_out;
    float radius; int use_radius;
};

template <int OP> __device__ __forceinline__ float term(float x, float q) {
    if (OP == OP_SQDIFF) { const float d = q - x; return d * d; }
    if (OP == OP_ABSDIFF || OP == OP_MAXABS) return fabsf(x - q);
    if (OP == OP_SQ) return x * x;
    return x * q;
}
template <int OP> __device__ __forceinline__ float comb(float a, float b) {
    return OP == OP_MAXABS ? fmaxf(a, b) : a + b;
}
template <int OP> __device__ __forceinline__ float accum(float r, float x, float q) {
    if (OP == OP_DOT) return __fmaf_rn(x, q, r);
    return comb<OP>(r, term<OP>(x, q));
}

template <int OP>
__global__ __launch_bounds__(NT) void scan_kernel(ScanParams p) {
    extern __shared__ __align__(16) float smem[];
    float* Xs = smem;                          // [NT][LW] staged row chunk
    float* Qs = Xs + NT * LW;                  // [ld] query (padded with zeros)
    float* St = Qs + p.ld;                     // [MAXSTACK][NT] partial-sum stack
    float* Ld = St + MAXSTACK * NT;            // [k][NT]
    int* Li = reinterpret_cast<int*>(Ld + p.k * NT);
    int* Red = Li + p.k * NT;                  // 12 ints for the block reduction

    const int tid = threadIdx.x;
    const int unit = blockIdx.x;
    const int split = blockIdx.y;
    const bool lists = p.list_off != nullptr;
    const int qi = lists ? unit / p.n_probe : unit;

    // rows of this unit
    int64_t seg0 = 0, seg1 = p.n;
    if (lists) {
        const int l = p.probes[unit];
        seg0 = p.list_off[l];
        seg1 = p.list_off[l + 1];
    }
    const int64_t ntiles = (seg1 - seg0 + NT - 1) / NT;

    // query -> LDS, scalar terms -> registers
    if (OP != OP_SQ) {
        for (int i = tid; i < p.ld; i += NT) Qs[i] = (i < p.dim) ? p.q[(int64_t)qi * p.ldq + i] : 0.0f;
    }
    float q_sq = 0.0f, q_nrm = 0.0f;
    if (p.metric == VDB_EUCLIDEAN || p.metric == VDB_EUCLIDEAN_SQ) q_sq = p.q_sq[qi];
    if (p.metric == VDB_COSINE || p.metric == VDB_COSINE_PAIR) q_nrm = p.q_nrm[qi];
    float lbd = -INFINITY; int64_t lbk = -1;
    if (p.lb_d) { lbd = p.lb_d[qi]; lbk = p.lb_k[qi]; }
    const int64_t key_hi = lists ? ((int64_t)(unit - qi * p.n_probe) << 32) : 0;   // probe rank orders equal distances

    const int k = p.k;
    list_init(Ld, Li, k, tid);
    float own_kth = INFINITY;

    const int ksteps = (p.ld + CK - 1) / CK;
    const int srow = tid >> 3;
    const int skc = (tid & 7) * 4;

    for (int64_t t = split; t < ntiles; t += p.splits) {
        const int64_t base = seg0 + t * NT;
        const int64_t pos = base + tid;                 // stored position of this thread's row
        const bool live = pos < seg1;

        float r[8];
        float res = 0.0f;
        int leaf = 0, in_leaf = 0, sp = 0;
        int L = p.plan.leaf_len[0];

        for (int ks = 0; ks < ksteps; ++ks) {
            __syncthreads();
            const int kk = ks * CK + skc;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int rr = ps * 32 + srow;
                const int64_t gr = min(base + rr, seg1 - 1);
                float4 v = make_float4(0, 0, 0, 0);
                if (kk < p.ld) v = *reinterpret_cast<const float4*>(p.data + gr * p.ld + kk);
                *reinterpret_cast<float4*>(Xs + rr * LW + skc) = v;
            }
            __syncthreads();

            const float* xr = Xs + tid * LW;
            const float* qr = Qs + ks * CK;
#pragma unroll
            for (int g = 0; g < CK / 8; ++g) {
                const int kpos = ks * CK + g * 8;
                if (kpos >= p.dim) break;
                const float4 x0 = *reinterpret_cast<const float4*>(xr + g * 8);
                const float4 x1 = *reinterpret_cast<const float4*>(xr + g * 8 + 4);
                float4 q0 = make_float4(0, 0, 0, 0), q1 = q0;
                if (OP != OP_SQ) {
                    q0 = *reinterpret_cast<const float4*>(qr + g * 8);
                    q1 = *reinterpret_cast<const float4*>(qr + g * 8 + 4);
                }
                const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                const float qs[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const int cnt = min(8, L - in_leaf);
                if (cnt == 8) {
                    if (in_leaf == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) r[j] = term<OP>(xs[j], qs[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) r[j] = accum<OP>(r[j], xs[j], qs[j]);
                    }
                    in_leaf += 8;
                    if (in_leaf == L)
                        res = comb<OP>(comb<OP>(comb<OP>(r[0], r[1]), comb<OP>(r[2], r[3])),
                                       comb<OP>(comb<OP>(r[4], r[5]), comb<OP>(r[6], r[7])));
                } else {
                    // sequential tail of the leaf (or a leaf shorter than 8)
                    if (in_leaf == 0) res = 0.0f;
                    else res = comb<OP>(comb<OP>(comb<OP>(r[0], r[1]), comb<OP>(r[2], r[3])),
                                        comb<OP>(comb<OP>(r[4], r[5]), comb<OP>(r[6], r[7])));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < cnt) res = comb<OP>(res, term<OP>(xs[j], qs[j]));
                    in_leaf += cnt;
                }
                if (in_leaf == L) {
                    // leaf finished: push, then fold pending partial sums right into left
                    St[sp * NT + tid] = res;
                    ++sp;
                    for (int f = 0; f < p.plan.folds[leaf]; ++f) {
                        const float b = St[(sp - 1) * NT + tid];
                        const float a = St[(sp - 2) * NT + tid];
                        St[(sp - 2) * NT + tid] = comb<OP>(a, b);
                        --sp;
                    }
                    ++leaf;
                    in_leaf = 0;
                    L = (leaf < p.plan.n_leaves) ? p.plan.leaf_len[leaf] : 0;
                }
            }
        }
        const float sum = St[tid];

        if (!live) continue;

        if (p.metric == METRIC_SQNORM) {
            p.dist_out[pos] = sum;
            p.nrm_out[pos] = sqrtf(sum);
            continue;
        }

        // ---- final distance value
        float d;
        switch (p.metric) {
            case VDB_EUCLIDEAN_DIRECT: d = sqrtf(sum); break;
            case VDB_EUCLIDEAN: case VDB_EUCLIDEAN_SQ: {
                const float sq = fmaxf(__fmaf_rn(-2.0f, sum, q_sq + p.row_sq[pos]), 0.0f);
                d = p.metric == VDB_EUCLIDEAN ? sqrtf(sq) : sq;
                break;
            }
            case VDB_COSINE: {
                if (q_nrm < 1e-8f) { d = 1.0f; break; }            // metrics.py:472-473
                float cn = p.row_nrm[pos];
                if (cn < 1e-8f) cn = 1.0f;                           // metrics.py:476
                const float sim = fminf(fmaxf(sum / (q_nrm * cn), -1.0f), 1.0f);
                d = 1.0f - sim;
                break;
            }
            case VDB_COSINE_PAIR: {
                const float den = q_nrm * p.row_nrm[pos];            // metrics.py:96-100
                const float sim = den < 1e-8f ? 0.0f : sum / den;
                d = 1.0f - sim;
                break;
            }
            case VDB_DOT: d = -sum; break;
            default: d = sum; break;    // direct squared L2, Manhattan, Chebyshev
        }

        if (p.dist_out) { p.dist_out[(int64_t)qi * p.n + pos] = d; }
        if (k == 0) continue;

        const int row = p.row_ids ? p.row_ids[pos] : (int)pos;
        if (p.mask && !((p.mask[row >> 5] >> (row & 31)) & 1u)) continue;
        if (p.use_radius && !(d <= p.radius)) continue;
        const bool above = (d > lbd) || (d == lbd && (key_hi | pos) > lbk);
        if (above && d < own_kth) own_kth = list_insert(Ld + tid, Li + tid, k, d, (int)pos);
    }

    if (k == 0 || p.part_d == nullptr) return;

    // ---- block reduction of the 256 lists to one list of k
    float* od = p.part_d + ((int64_t)unit * p.splits + split) * k;
    int* oi = p.part_i + ((int64_t)unit * p.splits + split) * k;
    block_reduce_lists(Ld, Li, k, tid, Red, od, oi);
}

// --------------------------------------------------------------------------
static size_t scan_lds_bytes(int ld, int k) {
    return (size_t)(NT * LW + ld + MAXSTACK * NT) * sizeof(float) + (size_t)k * NT * 8 + 12 * sizeof(int);
}

int scan_num_cand(const ScanJob& j, int* n_cand) {
    *n_cand = (j.list_off ? j.n_probe : 1) * j.splits;
    return VDB_OK;
}

static int launch_scan_params(ScanParams& p, int units, hipStream_t s) {
    const dim3 grid(units, p.splits);
    const size_t lds = scan_lds_bytes(p.ld, p.k);
    VDB_REQUIRE(lds <= 160 * 1024, "dimension %d with k=%d needs %zu bytes of LDS", p.dim, p.k, lds);
    KernelTimer timer(p.metric == METRIC_SQNORM ? nullptr : s, p.metric != METRIC_SQNORM);
#define VDB_LAUNCH_SCAN(O)                                                                     \
    case O: {                                                                                  \
        auto kern = scan_kernel<O>;                                                            \
        VDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);                                   \
        break;                                                                                 \
    }
    switch (op_of_metric(p.metric)) {
        VDB_LAUNCH_SCAN(OP_DOT)
        VDB_LAUNCH_SCAN(OP_SQDIFF)
        VDB_LAUNCH_SCAN(OP_ABSDIFF)
        VDB_LAUNCH_SCAN(OP_MAXABS)
        VDB_LAUNCH_SCAN(OP_SQ)
    }
#undef VDB_LAUNCH_SCAN
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_scan(const ScanJob& j, hipStream_t s) {
    VDB_REQUIRE(j.k >= 0 && j.k <= KMAX_FUSED, "scan supports k <= %d, got %d", KMAX_FUSED, j.k);
    VDB_REQUIRE(j.nq >= 1 && j.splits >= 1, "empty scan");
    ScanParams p{};
    p.data = j.data; p.ld = j.ld; p.dim = j.dim; p.row_sq = j.row_sq; p.row_nrm = j.row_nrm;
    p.plan = j.plan; p.metric = j.metric;
    p.q = j.q; p.ldq = j.ldq; p.nq = j.nq; p.q_sq = j.q_sq; p.q_nrm = j.q_nrm;
    p.n = j.n; p.list_off = j.list_off; p.probes = j.probes; p.n_probe = j.n_probe;
    p.row_ids = j.row_ids; p.mask = j.mask; p.lb_d = j.lb_d; p.lb_k = j.lb_k;
    p.k = j.k; p.splits = j.splits;
    p.part_d = j.part_d; p.part_i = j.part_i; p.dist_out = j.dist_out;
    p.radius = j.radius; p.use_radius = j.use_radius;
    const int units = j.list_off ? j.nq * j.n_probe : j.nq;
    return launch_scan_params(p, units, s);
}

int launch_row_norms(const float* data, int ld, int dim, int64_t n, const SumPlan& plan,
                     float* sq, float* nrm, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    ScanParams p{};
    p.data = data; p.ld = ld; p.dim = dim; p.plan = plan; p.metric = METRIC_SQNORM;
    p.nq = 1; p.n = n; p.k = 0;
    p.splits = (int)std::min<int64_t>((n + NT - 1) / NT, 4096);
    p.dist_out = sq; p.nrm_out = nrm;
    return launch_scan_params(p, 1, s);
}

}  // namespace vdb
