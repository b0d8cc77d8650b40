// One Lloyd accumulation step: per-cluster sums and member counts.
// index/ivf.py:177-182 and index/pq.py:160-165 add the rows one at a time in
// Python; here every row adds into fp64 accumulators with device atomics, so
// the order of additions is free but the fp64 sums agree with a sequential
// fp32/fp64 accumulation far below fp32 resolution.
#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float* __restrict__ x, int64_t n, int dim, int ldx,
                                                                const int* __restrict__ assign, int n_clusters,
                                                                double* __restrict__ sums,
                                                                unsigned long long* __restrict__ counts) {
    const int64_t total = n * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / dim;
        const int j = (int)(i - row * dim);
        const int c = assign[row];
        if (c < 0 || c >= n_clusters) continue;
        atomicAdd(&sums[(int64_t)c * dim + j], (double)x[row * ldx + j]);
        if (j == 0) atomicAdd(&counts[c], 1ull);
    }
}

int launch_kmeans_accumulate(const float* x, int64_t n, int dim, int ldx, const int* assign, int n_clusters,
                             double* sums, unsigned long long* counts, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    const int64_t total = n * dim;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(blocks), dim3(256), 0, s, x, n, dim, ldx, assign, n_clusters, sums, counts);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb

// ============================================================================================
// k-means on rows that stay in HBM (KMeans, index/ivf.py:102-237; the PQ and IVF-PQ sub-quantizers,
// index/pq.py:144-180, index/hybrid.py:174-203).
//
// A Lloyd update in the reference adds the rows of a cluster one at a time, in sample order, into
// fp32 sums (float64 sums from the second PQ step on).  To repeat exactly those additions the rows
// are first sorted by cluster with a STABLE counting sort (per-block histograms, a scan, a scatter
// that ranks equal keys by row order), and then one thread per (cluster, column) walks the cluster's
// members in ascending row order and adds them sequentially in the reference's accumulator type.
// Given equal assignments the new centroids are therefore bit-identical to the reference's, and no
// atomics are involved.
namespace vdb {

// ---- stable counting sort of the rows by cluster ---------------------------------------------
__global__ __launch_bounds__(256) void km_hist_kernel(const int* __restrict__ assign, int64_t n, int k, int64_t chunk,
                                                      int* __restrict__ hist) {
    extern __shared__ int h[];
    for (int i = threadIdx.x; i < k; i += 256) h[i] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = min(n, r0 + chunk);
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const int c = assign[r];
        if ((unsigned)c < (unsigned)k) atomicAdd(&h[c], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += 256) hist[(int64_t)blockIdx.x * k + i] = h[i];
}

// per cluster: exclusive prefix of its counts over the blocks (in place) and its total
__global__ void km_hist_scan_kernel(int* __restrict__ hist, int n_blocks, int k, int* __restrict__ totals) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    int run = 0;
    for (int b = 0; b < n_blocks; ++b) {
        const int v = hist[(int64_t)b * k + c];
        hist[(int64_t)b * k + c] = run;
        run += v;
    }
    totals[c] = run;
}

// exclusive scan of the cluster totals: off[0..k], one block
__global__ __launch_bounds__(1024) void km_cluster_scan_kernel(const int* __restrict__ totals, int k, int* __restrict__ off) {
    __shared__ int sh[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < k; c0 += 1024) {
        const int c = c0 + tid;
        const int v = c < k ? totals[c] : 0;
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = tid >= o ? sh[tid - o] : 0;
            __syncthreads();
            sh[tid] += a;
            __syncthreads();
        }
        if (c < k) off[c] = carry + sh[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (tid == 0) off[k] = carry;
}

// One wave per block walks its chunk 64 rows at a time; rows of one cluster keep their order: a
// row's place = first place of its cluster + members in earlier blocks + members earlier in this
// block (a running counter in LDS) + lanes below it in this step that carry the same cluster.
__global__ __launch_bounds__(64) void km_scatter_kernel(const int* __restrict__ assign, int64_t n, int k, int key_bits,
                                                        int64_t chunk, const int* __restrict__ hist,
                                                        const int* __restrict__ cluster_off, int* __restrict__ members) {
    extern __shared__ int cnt[];
    const int lane = threadIdx.x;
    for (int i = lane; i < k; i += 64) cnt[i] = cluster_off[i] + hist[(int64_t)blockIdx.x * k + i];
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = min(n, r0 + chunk);
    for (int64_t r = r0; r < r1; r += 64) {
        const int64_t row = r + lane;
        int c = row < r1 ? assign[row] : -1;
        const bool valid = (unsigned)c < (unsigned)k;
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < key_bits; ++bit) {
            const bool one = (c >> bit) & 1;
            const unsigned long long b = __ballot(valid && one);
            peers &= one ? b : ~b;
        }
        if (valid) {
            const int base = cnt[c];
            const int rank = __popcll(peers & ((1ull << lane) - 1ull));
            members[base + rank] = (int)row;
            if ((peers >> lane) == 1ull) cnt[c] = base + __popcll(peers);      // the highest lane of the group
        }
        __syncthreads();
    }
}

// ---- per-cluster sums in member (= sample) order, and the new centroids ------------------------
// FORM 0: KMeans (ivf.py:177-187)   fp32 sums, new = float(double(sum) / double(count)); an empty cluster is left zero
//         for the caller to reseed (:188-190)
// FORM 1: first PQ / IVF-PQ step (pq.py:160-169)  fp32 sums, new (float64) = double(sum) / max(count, 1)
// FORM 2: later PQ / IVF-PQ steps   float64 sums, new (float64) = sum / max(count, 1)
template <class ACC, int FORM>
__global__ __launch_bounds__(256) void km_sum_kernel(const float* __restrict__ x, int ldx, int dsub, int dp,
                                                     const int* __restrict__ members, const int* __restrict__ off, int k,
                                                     float* __restrict__ next32, double* __restrict__ next64,
                                                     long long* __restrict__ counts) {
    const int cpb = 256 / dp;
    const int cl = blockIdx.x * cpb + threadIdx.x / dp;
    const int col = blockIdx.y * dp + threadIdx.x % dp;
    if (cl >= k || col >= dsub) return;
    const int m0 = off[cl], m1 = off[cl + 1];
    ACC acc = 0;
    int m = m0;
    for (; m + 4 <= m1; m += 4) {
        const int a = members[m], b = members[m + 1], c = members[m + 2], d = members[m + 3];
        const float va = x[(int64_t)a * ldx + col], vb = x[(int64_t)b * ldx + col];
        const float vc = x[(int64_t)c * ldx + col], vd = x[(int64_t)d * ldx + col];
        acc = acc + (ACC)va; acc = acc + (ACC)vb; acc = acc + (ACC)vc; acc = acc + (ACC)vd;
    }
    for (; m < m1; ++m) acc = acc + (ACC)x[(int64_t)members[m] * ldx + col];
    const int cnt = m1 - m0;
    const int64_t at = (int64_t)cl * dsub + col;
    if (FORM == 0) {
        next32[at] = cnt > 0 ? (float)((double)acc / (double)cnt) : 0.0f;
    } else {
        const double v = (double)acc / (double)(cnt > 0 ? cnt : 1);
        next64[at] = v;
        next32[at] = (float)v;
    }
    if (col == 0) counts[cl] = cnt;
}

// np.sum((new - old) ** 2) (ivf.py:193): fp32 differences and squares, summed in float64
__global__ __launch_bounds__(1024) void km_shift_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t total,
                                                        double* __restrict__ out) {
    __shared__ double sh[1024];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += 1024) {
        const float d = a[i] - b[i];
        acc += (double)(d * d);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// dst[i][0..dsub) = x[rows[i]][0..dsub)   (i-th destination row may be remapped through dst_rows)
__global__ void km_gather_kernel(const float* __restrict__ x, int ldx, int dsub, const long long* __restrict__ rows,
                                 const int* __restrict__ dst_rows, int count, float* __restrict__ dst32,
                                 double* __restrict__ dst64) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * dsub) return;
    const int i = t / dsub, j = t - i * dsub;
    const float v = x[(int64_t)rows[i] * ldx + j];
    const int64_t at = (int64_t)(dst_rows ? dst_rows[i] : i) * dsub + j;
    dst32[at] = v;
    if (dst64) dst64[at] = (double)v;
}

// a float rounded to bf16 (nearest even), as a float
__device__ __forceinline__ float bf16_rounded(float f) {
    const uint32_t v = __float_as_uint(f);
    const uint32_t b = ((v & 0x7fffffffu) > 0x7f800000u) ? ((v >> 16) | 0x40u) : ((v + 0x7fffu + ((v >> 16) & 1u)) >> 16);
    return __uint_as_float(b << 16);
}

// one training row as the query of the pass that follows, with the norms the row store keeps for it (the values a
// norm pass over the copy would compute again)
__global__ void km_gather_query_kernel(const float* __restrict__ x, int ldx, int dsub, const long long* __restrict__ row,
                                       const float* __restrict__ x_sq, const float* __restrict__ x_nrm,
                                       float* __restrict__ q, float* __restrict__ q_sq, float* __restrict__ q_nrm,
                                       float* __restrict__ q16f, int q16f_n) {
    const long long r = row[0];
    for (int j = threadIdx.x; j < dsub; j += blockDim.x) q[j] = x[r * ldx + j];
    for (int j = threadIdx.x; j < q16f_n; j += blockDim.x) q16f[j] = j < dsub ? bf16_rounded(x[r * ldx + j]) : 0.0f;
    if (threadIdx.x == 0) { q_sq[0] = x_sq[r]; q_nrm[0] = x_nrm[r]; }
}

int launch_km_gather_query(const float* x, int ldx, int dsub, const long long* row, const float* x_sq, const float* x_nrm,
                           float* q, float* q_sq, float* q_nrm, hipStream_t s, float* q16f, int q16f_n) {
    hipLaunchKernelGGL(km_gather_query_kernel, dim3(1), dim3(256), 0, s, x, ldx, dsub, row, x_sq, x_nrm, q, q_sq, q_nrm, q16f, q16f_n);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// ---- float64 assignment of the later PQ steps (pq.py:156-157 with float64 centroids) -----------
// distance = (double(|x|^2 fp32) + |c|^2 float64) - (2x) . c in float64, no clamp; first minimum.
__global__ void km_csq64_kernel(const double* __restrict__ c, int k, int dsub, double* __restrict__ csq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const double* v = c + (int64_t)i * dsub;
    // np.sum(Y ** 2, axis=1) of a float64 row: pairwise order (8 strided accumulators per leaf of <= 128)
    double stack[16];
    int sp = 0;
    // leaves are produced by the same recursive split NumPy uses
    struct Seg { int lo, n; };
    Seg todo[32];
    int tp = 0;
    todo[tp++] = Seg{0, dsub};
    // iterative post-order: a negative n marks "fold the two results on the stack"
    while (tp > 0) {
        const Seg s = todo[--tp];
        if (s.n < 0) { const double b = stack[--sp], a = stack[--sp]; stack[sp++] = a + b; continue; }
        if (s.n > 128) {
            int h = s.n / 2;
            h -= h % 8;
            todo[tp++] = Seg{0, -1};
            todo[tp++] = Seg{s.lo + h, s.n - h};
            todo[tp++] = Seg{s.lo, h};
            continue;
        }
        double res;
        if (s.n < 8) {
            res = 0.0;
            for (int j = 0; j < s.n; ++j) res = res + v[s.lo + j] * v[s.lo + j];
        } else {
            double r[8];
            for (int j = 0; j < 8; ++j) r[j] = v[s.lo + j] * v[s.lo + j];
            int i2 = 8;
            for (; i2 + 8 <= s.n; i2 += 8)
                for (int j = 0; j < 8; ++j) r[j] = r[j] + v[s.lo + i2 + j] * v[s.lo + i2 + j];
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i2 < s.n; ++i2) res = res + v[s.lo + i2] * v[s.lo + i2];
        }
        stack[sp++] = res;
    }
    csq[i] = stack[0];
}

template <int DS>
__global__ __launch_bounds__(256) void km_assign_f64_kernel(const float* __restrict__ x, int ldx, int dsub, int64_t n,
                                                            const float* __restrict__ xsq, const double* __restrict__ cent,
                                                            const double* __restrict__ csq, int k, int lds_cent,
                                                            int* __restrict__ assign) {
    extern __shared__ double shc[];             // centroids (when they fit) and their squared norms
    double* scent = shc;
    double* scsq = shc + (lds_cent ? (int64_t)k * dsub : 0);
    if (lds_cent) for (int i = threadIdx.x; i < k * dsub; i += 256) scent[i] = cent[i];
    for (int i = threadIdx.x; i < k; i += 256) scsq[i] = csq[i];
    __syncthreads();
    const double* cc = lds_cent ? scent : cent;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const float* xr = x + row * ldx;
    double x2[DS > 0 ? DS : 1];
    if (DS > 0) {
#pragma unroll
        for (int j = 0; j < DS; ++j) x2[j] = (double)(2.0f * xr[j]);
    }
    const double xs = (double)xsq[row];
    double best = INFINITY;
    int best_c = 0;
    for (int c = 0; c < k; ++c) {
        const double* cv = cc + (int64_t)c * dsub;
        double dot = 0.0;
        if (DS > 0) {
#pragma unroll
            for (int j = 0; j < DS; ++j) dot = fma(x2[j], cv[j], dot);
        } else {
            for (int j = 0; j < dsub; ++j) dot = fma((double)(2.0f * xr[j]), cv[j], dot);
        }
        const double d = (xs + scsq[c]) - dot;
        if (d < best) { best = d; best_c = c; }
    }
    assign[row] = best_c;
}

// ---- k-means++ (ivf.py:204-226) -----------------------------------------------------------------
// nearest[i] = min over the chosen centroids of pairwise_euclidean(x_i, c)^2 kept as a running minimum
// (the square of the rounded distance, as the reference squares what pairwise_euclidean returns);
// sums of 1024 entries for the draw that follows.
static constexpr int PP_BLOCK = 1024;

// With q_sq / row_sq, `dist` holds the streamed dot products and the distance is finished here as the distance pass
// finishes it (sqrt(max(|q|^2 + |x|^2 - 2 q.x, 0)), metrics.py:456-462): one kernel and 8 MB of traffic less per round.
__global__ __launch_bounds__(256) void km_pp_update_kernel(const float* __restrict__ dist, int64_t n, int first,
                                                           float* __restrict__ nearest, double* __restrict__ block_sums,
                                                           const float* __restrict__ q_sq, const float* __restrict__ row_sq) {
    __shared__ double sh[256];
    const int64_t base = (int64_t)blockIdx.x * PP_BLOCK;
    double acc = 0.0;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t i = base + ps * 256 + threadIdx.x;
        if (i < n) {
            float d = dist[i];
            if (q_sq) d = sqrtf(fmaxf(__fmaf_rn(-2.0f, d, q_sq[0] + row_sq[i]), 0.0f));
            float v = d * d;
            if (!first) v = fminf(nearest[i], v);
            nearest[i] = v;
            acc += (double)v;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}

// The round's pass and update in one, over bf16 copies of the rows (tiled as the IVF scan's copy: the 16-byte pieces of
// 32 consecutive rows lie together).  A new centroid lowers the running minimum of few rows.  For every row the key
// A = |q|^2 + |x|^2 - 2 bf16(q).bf16(x) lies within S (key_slack) of the float32 value R the distance pass computes
// before its square root, and the update's value is sqrtf(R)^2 >= R (1 - 4 * 2^-24): a row with
// (A - S) * 0.999999 >= nearest[row] keeps its minimum and is not read in float32 at all.  The others (the rows near the
// new centroid, and those whose minimum lies within the slack) get the distance pass's own arithmetic: the streaming
// kernel's dot product of a one-leaf row (eight fused partial sums, elements j, j + 8, ...; ((r0+r1)+(r2+r3)) +
// ((r4+r5)+(r6+r7))), fma(-2, dot, |q|^2 + |x|^2), max 0, sqrtf, square.  Same minima, same block sums (a thread's four
// rows in float64, then the tree over the block) as km_pp_update_kernel.
// (KmBf16Pass: x16 = the tiled copy, `pieces` 16-byte pieces per row; x = the float32 rows, dim a multiple of 8 and at
// most 128; q16f = the new centroid's elements rounded to bf16, as floats, pieces * 8 of them.)
template <int PIECES>
__global__ __launch_bounds__(256) void km_pp_bf16_kernel(KmBf16Pass p) {
    __shared__ double sh[256];
    const float q_sq = p.q_sq[0], qn = p.q_nrm[0], big = p.max_nrm[0];
    const bool trusted = qn < 1e18f && big < 1e18f;                       // else every row takes the float32 way
    const float slack = trusted ? key_slack(p.dim, qn, big, q_sq) : 0.0f;
    const int64_t base = (int64_t)blockIdx.x * PP_BLOCK;
    double acc_sum = 0.0;
    uint4 w[PIECES];
    auto load_row = [&](int64_t i) {
        const int64_t at = (((i >> 5) * PIECES) * 32 + (i & 31)) * 8;      // piece 0 of row i; piece c lies c * 256 elements on
#pragma unroll
        for (int c = 0; c < PIECES; ++c) w[c] = *reinterpret_cast<const uint4*>(p.x16 + at + (int64_t)c * 256);
    };
    if (base + threadIdx.x < p.n) load_row(base + threadIdx.x);
#pragma unroll 1
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t i = base + ps * 256 + threadIdx.x;
        if (i < p.n) {
            float a = 0.0f;
#pragma unroll
            for (int c = 0; c < PIECES; ++c) {
                const uint32_t v[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
                // (the centroid's elements through the scalar cache, eight at a time: in vector registers all 128 of them
                // would leave a wave per SIMD)
                typedef int km_i8 __attribute__((ext_vector_type(8)));
                const km_i8 qw = *reinterpret_cast<const __attribute__((address_space(4))) km_i8*>(reinterpret_cast<uintptr_t>(p.q16f + c * 8));
                const float qs[8] = {__int_as_float(qw[0]), __int_as_float(qw[1]), __int_as_float(qw[2]), __int_as_float(qw[3]),
                                     __int_as_float(qw[4]), __int_as_float(qw[5]), __int_as_float(qw[6]), __int_as_float(qw[7])};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a = __fmaf_rn(__uint_as_float(v[e] << 16), qs[2 * e], a);
                    a = __fmaf_rn(__uint_as_float(v[e] & 0xffff0000u), qs[2 * e + 1], a);
                }
            }
            const int64_t next = i + 256;
            if (ps < 3 && next < p.n) load_row(next);                     // on its way while this row is decided
            const float x_sq = p.x_sq[i];
            const float old = p.nearest[i];
            const float key = __fmaf_rn(-2.0f, a, q_sq + x_sq);
            float v = old;
            if (!(trusted && (key - slack) * 0.999999f >= old)) {
                const float* xr = p.x + i * p.ldx;
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = 0.0f;
                for (int g = 0; g < p.dim; g += 8) {
                    const float4 xa = *reinterpret_cast<const float4*>(xr + g), xb = *reinterpret_cast<const float4*>(xr + g + 4);
                    const float4 qa = *reinterpret_cast<const float4*>(p.q + g), qb = *reinterpret_cast<const float4*>(p.q + g + 4);
                    r[0] = __fmaf_rn(xa.x, qa.x, r[0]); r[1] = __fmaf_rn(xa.y, qa.y, r[1]);
                    r[2] = __fmaf_rn(xa.z, qa.z, r[2]); r[3] = __fmaf_rn(xa.w, qa.w, r[3]);
                    r[4] = __fmaf_rn(xb.x, qb.x, r[4]); r[5] = __fmaf_rn(xb.y, qb.y, r[5]);
                    r[6] = __fmaf_rn(xb.z, qb.z, r[6]); r[7] = __fmaf_rn(xb.w, qb.w, r[7]);
                }
                const float dot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                const float d = sqrtf(fmaxf(__fmaf_rn(-2.0f, dot, q_sq + x_sq), 0.0f));
                v = fminf(old, d * d);
                if (v != old) p.nearest[i] = v;
            }
            acc_sum += (double)v;
        }
    }
    sh[threadIdx.x] = acc_sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.block_sums[blockIdx.x] = sh[0];
}

int launch_km_pp_bf16(const KmBf16Pass& p, hipStream_t s) {
    const dim3 grid((unsigned)km_pp_blocks(p.n));
    switch (p.pieces) {
        case 4: hipLaunchKernelGGL(km_pp_bf16_kernel<4>, grid, dim3(256), 0, s, p); break;
        case 8: hipLaunchKernelGGL(km_pp_bf16_kernel<8>, grid, dim3(256), 0, s, p); break;
        case 12: hipLaunchKernelGGL(km_pp_bf16_kernel<12>, grid, dim3(256), 0, s, p); break;
        case 16: hipLaunchKernelGGL(km_pp_bf16_kernel<16>, grid, dim3(256), 0, s, p); break;
        default: set_error("bf16 k-means++ pass: %d pieces per row", p.pieces); return VDB_ERR_INVALID;
    }
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// rng.choice(n, p = nearest / nearest.sum()) for the uniform draw u (mtrand: cdf = cumsum(p); cdf /= cdf[-1];
// index = searchsorted(cdf, u, side="right")): the first row whose cumulative share exceeds u.
// One block: scan of the block sums, then of the 1024 entries of the block the draw falls into.
__device__ __forceinline__ void km_pp_select_body(const float* __restrict__ nearest, int64_t n,
                                                  const double* __restrict__ block_sums, int64_t nb, double u,
                                                  long long* __restrict__ chosen, long long* __restrict__ bad_weights) {
    __shared__ double sh[1024];
    __shared__ double carry, total_s, before_s;
    __shared__ long long pick_block;
    const int tid = threadIdx.x;
    double part = 0.0;
    for (int64_t b = tid; b < nb; b += 1024) part += block_sums[b];
    sh[tid] = part;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) { total_s = sh[0]; carry = 0.0; pick_block = -1; before_s = 0.0; }
    __syncthreads();
    const double total = total_s;
    if (!(total > 0.0) || !(total < INFINITY)) {
        // every row coincides with a chosen centroid (or a distance is not a number): the weights are 0/0, and the
        // reference's rng.choice raises "probabilities contain NaN" (ivf.py:224).  Row 0 keeps the rounds that are
        // already enqueued inside the buffers; the caller reports the error.
        if (tid == 0) { chosen[0] = 0; *bad_weights = 1; }
        return;
    }
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        const int64_t b = b0 + tid;
        const double v = b < nb ? block_sums[b] : 0.0;
        __syncthreads();
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const double a = tid >= o ? sh[tid - o] : 0.0;
            __syncthreads();
            sh[tid] += a;
            __syncthreads();
        }
        const double incl = carry + sh[tid];
        const bool over = b < nb && (incl / total) > u;
        const bool prev_over = tid > 0 ? ((carry + sh[tid - 1]) / total) > u : false;
        if (over && !prev_over && pick_block < 0) { pick_block = b; before_s = incl - v; }
        __syncthreads();
        if (pick_block >= 0) break;
        if (tid == 1023) carry += sh[1023];
        __syncthreads();
    }
    __syncthreads();
    long long pb = pick_block;
    if (pb < 0) {                                   // u beyond every cumulative share (rounding): the last row
        if (tid == 0) chosen[0] = n - 1;
        return;
    }
    const int64_t i = pb * PP_BLOCK + tid;
    const double v = i < n ? (double)nearest[i] : 0.0;
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const double a = tid >= o ? sh[tid - o] : 0.0;
        __syncthreads();
        sh[tid] += a;
        __syncthreads();
    }
    const double before = before_s;
    const bool over = i < n && ((before + sh[tid]) / total) > u;
    const bool prev_over = tid > 0 ? ((before + sh[tid - 1]) / total) > u : false;
    if (over && !prev_over) chosen[0] = i;
    // the block sum said the draw falls in here; if rounding puts it just past the block's end, its last row
    const int64_t last = min<int64_t>(n - 1, pb * PP_BLOCK + (PP_BLOCK - 1));
    if (i == last && !over) chosen[0] = last;
}

// ... and, with `q`, the chosen row copied out as the query of the next round with its stored norms (the launch that
// would gather it otherwise)
__global__ __launch_bounds__(1024) void km_pp_select_kernel(const float* __restrict__ nearest, int64_t n,
                                                            const double* __restrict__ block_sums, int64_t nb, double u,
                                                            long long* __restrict__ chosen, long long* __restrict__ bad_weights,
                                                            const float* __restrict__ x, int ldx, int dsub,
                                                            const float* __restrict__ x_sq, const float* __restrict__ x_nrm,
                                                            float* __restrict__ q, float* __restrict__ q_sq, float* __restrict__ q_nrm,
                                                            float* __restrict__ q16f, int q16f_n) {
    km_pp_select_body(nearest, n, block_sums, nb, u, chosen, bad_weights);
    if (!q) return;
    __syncthreads();
    const long long r = __hip_atomic_load(chosen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // written by one thread above
    for (int j = threadIdx.x; j < dsub; j += 1024) q[j] = x[r * ldx + j];
    for (int j = threadIdx.x; j < q16f_n; j += 1024) q16f[j] = j < dsub ? bf16_rounded(x[r * ldx + j]) : 0.0f;
    if (threadIdx.x == 0) { q_sq[0] = x_sq[r]; q_nrm[0] = x_nrm[r]; }
}

// ---- launchers -------------------------------------------------------------------------------------
int km_sort_chunk(int64_t n, int64_t* chunk, int* n_blocks) {
    int64_t c = std::max<int64_t>(1024, (n + 2047) / 2048);
    c = (c + 63) / 64 * 64;
    *chunk = c;
    *n_blocks = (int)((n + c - 1) / c);
    return VDB_OK;
}

int launch_km_sort(const int* assign, int64_t n, int k, int* hist, int* totals, int* cluster_off, int* members, hipStream_t s) {
    VDB_REQUIRE(k >= 1 && k <= 32768, "device k-means handles up to 32768 clusters, got %d", k);
    VDB_REQUIRE(n < (int64_t)0x7fffffff, "device k-means handles up to 2^31-1 training rows");
    int64_t chunk; int nb;
    km_sort_chunk(n, &chunk, &nb);
    const size_t lds = (size_t)k * sizeof(int);
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(km_hist_kernel), lds));
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(km_scatter_kernel), lds));
    hipLaunchKernelGGL(km_hist_kernel, dim3(nb), dim3(256), lds, s, assign, n, k, chunk, hist);
    hipLaunchKernelGGL(km_hist_scan_kernel, dim3((k + 255) / 256), dim3(256), 0, s, hist, nb, k, totals);
    hipLaunchKernelGGL(km_cluster_scan_kernel, dim3(1), dim3(1024), 0, s, totals, k, cluster_off);
    int key_bits = 1;
    while ((1 << key_bits) < k) ++key_bits;
    hipLaunchKernelGGL(km_scatter_kernel, dim3(nb), dim3(64), lds, s, assign, n, k, key_bits, chunk, hist, cluster_off, members);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_km_sums(int form, const float* x, int ldx, int dsub, const int* members, const int* cluster_off, int k,
                   float* next32, double* next64, long long* counts, hipStream_t s) {
    int dp = 1;
    while (dp < dsub && dp < 256) dp <<= 1;
    const dim3 grid((k + 256 / dp - 1) / (256 / dp), (dsub + dp - 1) / dp);
    if (form == 0) hipLaunchKernelGGL((km_sum_kernel<float, 0>), grid, dim3(256), 0, s, x, ldx, dsub, dp, members, cluster_off, k, next32, next64, counts);
    else if (form == 1) hipLaunchKernelGGL((km_sum_kernel<float, 1>), grid, dim3(256), 0, s, x, ldx, dsub, dp, members, cluster_off, k, next32, next64, counts);
    else hipLaunchKernelGGL((km_sum_kernel<double, 2>), grid, dim3(256), 0, s, x, ldx, dsub, dp, members, cluster_off, k, next32, next64, counts);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_km_shift(const float* a, const float* b, int64_t total, double* out, hipStream_t s) {
    hipLaunchKernelGGL(km_shift_kernel, dim3(1), dim3(1024), 0, s, a, b, total, out);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_km_gather(const float* x, int ldx, int dsub, const long long* rows, const int* dst_rows, int count,
                     float* dst32, double* dst64, hipStream_t s) {
    if (count <= 0) return VDB_OK;
    hipLaunchKernelGGL(km_gather_kernel, dim3((count * dsub + 255) / 256), dim3(256), 0, s, x, ldx, dsub, rows, dst_rows, count,
                       dst32, dst64);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_km_assign_f64(const float* x, int ldx, int dsub, int64_t n, const float* xsq, const double* cent, double* csq,
                         int k, int* assign, hipStream_t s) {
    hipLaunchKernelGGL(km_csq64_kernel, dim3((k + 63) / 64), dim3(64), 0, s, cent, k, dsub, csq);
    const size_t cent_bytes = (size_t)k * dsub * sizeof(double);
    const int lds_cent = cent_bytes + (size_t)k * 8 <= 96 * 1024;
    const size_t lds = (lds_cent ? cent_bytes : 0) + (size_t)k * sizeof(double);
    const dim3 grid((unsigned)((n + 255) / 256));
#define VDB_KM_ASSIGN(DS)                                                                             \
    {                                                                                                 \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(km_assign_f64_kernel<DS>), lds));      \
        hipLaunchKernelGGL(km_assign_f64_kernel<DS>, grid, dim3(256), lds, s, x, ldx, dsub, n, xsq, cent, csq, k, lds_cent, assign); \
    }
    switch (dsub) {
        case 2: VDB_KM_ASSIGN(2) break;
        case 4: VDB_KM_ASSIGN(4) break;
        case 8: VDB_KM_ASSIGN(8) break;
        case 16: VDB_KM_ASSIGN(16) break;
        case 32: VDB_KM_ASSIGN(32) break;
        default: VDB_KM_ASSIGN(0) break;
    }
#undef VDB_KM_ASSIGN
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int64_t km_pp_blocks(int64_t n) { return (n + PP_BLOCK - 1) / PP_BLOCK; }

int launch_km_pp_step(const float* dist, int64_t n, bool first, float* nearest, double* block_sums, double u,
                      long long* chosen, long long* bad_weights, hipStream_t s, const float* q_sq, const float* row_sq,
                      const KmNextQuery* next) {
    const int64_t nb = km_pp_blocks(n);
    hipLaunchKernelGGL(km_pp_update_kernel, dim3((unsigned)nb), dim3(256), 0, s, dist, n, first ? 1 : 0, nearest, block_sums,
                       q_sq, row_sq);
    const KmNextQuery none{};
    const KmNextQuery& nx = next ? *next : none;
    hipLaunchKernelGGL(km_pp_select_kernel, dim3(1), dim3(1024), 0, s, nearest, n, block_sums, nb, u, chosen, bad_weights,
                       nx.x, nx.ldx, nx.dsub, nx.x_sq, nx.x_nrm, nx.q, nx.q_sq, nx.q_nrm, nx.q16f, nx.q16f_n);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// the selection of a round alone (the pass above has written the block sums)
int launch_km_pp_select(const float* nearest, int64_t n, const double* block_sums, double u, long long* chosen,
                        long long* bad_weights, const KmNextQuery& nx, hipStream_t s) {
    hipLaunchKernelGGL(km_pp_select_kernel, dim3(1), dim3(1024), 0, s, nearest, n, block_sums, km_pp_blocks(n), u, chosen,
                       bad_weights, nx.x, nx.ldx, nx.dsub, nx.x_sq, nx.x_nrm, nx.q, nx.q_sq, nx.q_nrm, nx.q16f, nx.q16f_n);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
