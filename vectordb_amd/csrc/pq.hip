// Product-quantisation kernels.
//
//  pq_lut_kernel    : the per-query distance tables of PQIndex.search
//                     (index/pq.py:270-283): table[m][c] = np.sum((codebook[m][c] - q_m)**2)
//  pq_encode_kernel : PQIndex._encode (index/pq.py:188-200): per sub-space the
//                     first codeword of minimum direct-form squared distance
//  pq_scan_kernel   : the ADC scan (index/pq.py:287-297): codes streamed from
//                     HBM one row per lane, the query's table in LDS, fp32 sum
//                     in sub-space order 0..m-1 (what Python's sum() does there),
//                     sqrt, filter bit tested in the scan, k best kept.
//
// All three reproduce the reference's fp32 results bit for bit: the sums over
// a sub-vector follow NumPy's pairwise order and nothing is contracted to fma.
#include <type_traits>
#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

// np.sum((a - b)**2) over n contiguous elements, n <= 128 (one pairwise leaf)
__device__ __forceinline__ static float np_leaf_sqdiff(const float* a, const float* b, int n) {
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) { const float d = a[i] - b[i]; r = r + d * d; }
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = a[j] - b[j]; r[j] = d * d; }
    int i = 8;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = a[i + j] - b[i + j]; r[j] = r[j] + d * d; }
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) { const float d = a[i] - b[i]; res = res + d * d; }
    return res;
}

// the same with the elements of a `stride` floats apart (a codebook laid out codeword-fastest in LDS)
__device__ __forceinline__ static float np_leaf_sqdiff_strided(const float* a, int stride, const float* b, int n) {
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) { const float d = a[i * stride] - b[i]; r = r + d * d; }
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = a[j * stride] - b[j]; r[j] = d * d; }
    int i = 8;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = a[(i + j) * stride] - b[i + j]; r[j] = r[j] + d * d; }
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) { const float d = a[i * stride] - b[i]; res = res + d * d; }
    return res;
}

// any n: walk the plan (leaves left to right, fold pending partial sums)
__device__ static float np_sum_sqdiff(const float* a, const float* b, int n, const SumPlan& plan) {
    if (n <= 128) return np_leaf_sqdiff(a, b, n);
    float st[8];
    int sp = 0, pos = 0;
    for (int l = 0; l < plan.n_leaves; ++l) {
        const int len = plan.leaf_len[l];
        const float v = np_leaf_sqdiff(a + pos, b + pos, len);
        pos += len;
        // static-index push/fold so the stack stays in registers
        int f = plan.folds[l];
        float top = v;
        while (f > 0) { top = st[sp - 1] + top; --sp; --f; }
        st[sp++] = top;
    }
    return st[0];
}

// ---------------------------------------------------------------- tables
__global__ __launch_bounds__(256) void pq_lut_kernel(const float* __restrict__ q, int ldq,
                                                     const float* __restrict__ book, int m, int ksub, int dsub,
                                                     SumPlan plan, float* __restrict__ lut, int* __restrict__ zero_counts,
                                                     float* __restrict__ mins, int* __restrict__ zero_more, int zero_more_n) {
    extern __shared__ __align__(16) float qs[];
    __shared__ float wave_min[4];
    const int sub = blockIdx.x, qi = blockIdx.y;
    if (zero_counts && sub == 0 && threadIdx.x == 0) zero_counts[qi] = 0;    // the scan's per-query queue lengths
    if (sub == 0 && qi == 0)                                                 // the byte-table pass's counters and flags
        for (int i = threadIdx.x; i < zero_more_n; i += blockDim.x) zero_more[i] = 0;
    for (int i = threadIdx.x; i < dsub; i += blockDim.x) qs[i] = q[(int64_t)qi * ldq + sub * dsub + i];
    __syncthreads();
    float least = INFINITY;
    for (int c = threadIdx.x; c < ksub; c += blockDim.x) {
        const float* cw = book + ((int64_t)sub * ksub + c) * dsub;
        const float v = np_sum_sqdiff(cw, qs, dsub, plan);
        lut[((int64_t)qi * m + sub) * ksub + c] = v;
        least = fminf(least, v);                 // fminf skips NaN entries
    }
    if (mins) {                                  // the smallest entry of the table row (the byte tables start from it)
        for (int o = 32; o > 0; o >>= 1) least = fminf(least, __shfl_xor(least, o));
        if ((threadIdx.x & 63) == 0) wave_min[threadIdx.x >> 6] = least;
        __syncthreads();
        if (threadIdx.x == 0) mins[(int64_t)qi * m + sub] = fminf(fminf(wave_min[0], wave_min[1]), fminf(wave_min[2], wave_min[3]));
    }
}

// ---------------------------------------------------------------- encode
// block = 256 rows of one sub-space.  The sub-space's codewords and the rows'
// sub-vectors are staged in LDS when they fit (CB_LDS / X_LDS), otherwise read
// from global memory through the caches.
template <bool CB_LDS, bool X_LDS>
__global__ __launch_bounds__(256) void pq_encode_kernel(const float* __restrict__ x, int64_t n, int ldx,
                                                        const float* __restrict__ book, int m, int ksub, int dsub,
                                                        SumPlan plan, uint8_t* __restrict__ codes) {
    extern __shared__ __align__(16) float sm[];
    float* cb_s = sm;                                       // [ksub][dsub]
    float* xs = sm + (CB_LDS ? ksub * dsub : 0);            // [256][dsub + 1]
    const int sub = blockIdx.y;
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const float* book_sub = book + (int64_t)sub * ksub * dsub;
    if (CB_LDS)
        for (int i = tid; i < ksub * dsub; i += 256) cb_s[i] = book_sub[i];
    if (X_LDS)
        for (int i = tid; i < 256 * dsub; i += 256) {
            const int r = i / dsub, j = i - r * dsub;
            const int64_t gr = min(row0 + r, n - 1);
            xs[r * (dsub + 1) + j] = x[gr * ldx + sub * dsub + j];
        }
    __syncthreads();
    const int64_t row = row0 + tid;
    if (row >= n) return;
    const float* cb = CB_LDS ? cb_s : book_sub;
    const float* mine = X_LDS ? xs + tid * (dsub + 1) : x + row * ldx + sub * dsub;
    float best = INFINITY;
    int best_c = 0;
    for (int c = 0; c < ksub; ++c) {
        const float d = np_sum_sqdiff(cb + c * dsub, mine, dsub, plan);
        if (d < best) { best = d; best_c = c; }       // strict: first minimum wins (np.argmin)
    }
    codes[row * m + sub] = (uint8_t)best_c;
}

// ---------------------------------------------------------------- scan
struct PqScanParams {
    const uint8_t* codes; int64_t n; int m; int ksub;
    const float* lut; int nq;        // [nq][m][ksub]
    const uint32_t* mask;
    const float* lb_d; const int64_t* lb_k;
    int k; int splits;
    float* part_d; int* part_i;      // [nq][splits * PQ_WAVES][k]: one list per wave, or (queue_len) a query's entries packed
    int* queue_len;                  // optional [nq]: a wave appends its list's entries at part_*[q][queue_len[q]...]
    float* dist_out;                 // optional [nq][n]
    const float* bound_d; int64_t bound_stride;   // optional: query q's k-th best is not above bound_d[q * bound_stride]
    const int* run_if;               // optional device flag: the launch does nothing while it is zero
};

static constexpr int PQ_WAVES = 4;       // waves of a split: one candidate list each

// M > 0: compile-time number of sub-spaces (multiple of 4); 0: run-time.
// Q queries share one pass over the codes: their tables are interleaved in LDS,
// T[(j * ksub + c) * Q + q], so one LDS read of Q floats serves all of them.
// KSUB > 0: compile-time codewords per sub-space (256 for 8-bit codes): a sub-space's table offset becomes the
// immediate offset of the LDS read instead of an addition per row.
// PQ_UNROLL: row tiles whose code loads are in flight together (measured on 100M codes: 4 beats 2 and 8; blocks of
// eight waves that share the tables of a four-query pass are slower than blocks of four).
template <int M, int Q, int KSUB = 0, int PQ_UNROLL = 4>
__global__ __launch_bounds__(256) void pq_scan_kernel(PqScanParams p) {
    extern __shared__ __align__(16) float sm[];
    if (p.run_if && *p.run_if == 0) return;
    const int m = M > 0 ? M : p.m;
    const int ksub = KSUB > 0 ? KSUB : p.ksub;
    float* T = sm;                                  // [m][ksub][Q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, q0 = blockIdx.y * Q;
    const int qcnt = min(Q, p.nq - q0);
    const int k = p.k;

    for (int i = tid; i < m * ksub; i += 256) {
#pragma unroll
        for (int q = 0; q < Q; ++q) T[i * Q + q] = p.lut[(int64_t)(q0 + min(q, qcnt - 1)) * m * ksub + i];
    }
    WaveTopK best[Q];
    float lbd[Q];
    int64_t lbk[Q];
    // A row can only enter a list if sqrt(sum) <= the list's k-th distance.  thr2 is a bound on the SUM that no such
    // row exceeds (the square of the k-th distance, a few ulps up; never below the square of the smallest distance
    // whose own square still is a normal number), so the correctly rounded root, ten instructions per row and
    // query, is only taken for the rare tile in which some lane passes it.
    // thr2g: the same kind of bound from a distance that is known not to be below the query's k-th best over ALL
    // rows (the k-th best of a sample of the rows, p.bound_d): it holds from the first tile on and for every wave.
    float thr2[Q], thr2g[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        best[q].init(k > 0 ? k : 1);
        lbd[q] = -INFINITY; lbk[q] = -1;
        thr2g[q] = INFINITY;
        if (p.bound_d) {
            const float t = p.bound_d[(int64_t)(q0 + min(q, qcnt - 1)) * p.bound_stride];
            thr2g[q] = fmaxf(__fmaf_rn(t * t, 4e-7f, t * t), 1e-36f);
        }
        thr2[q] = thr2g[q];
        if (p.lb_d && q < qcnt) { lbd[q] = p.lb_d[q0 + q]; lbk[q] = p.lb_k[q0 + q]; }
    }
    const bool roots_always = p.dist_out != nullptr || p.lb_d != nullptr;      // wave-uniform
    __syncthreads();

    const int64_t ntiles = (p.n + 255) / 256;
    const int64_t per = (ntiles + p.splits - 1) / p.splits;
    const int64_t t0 = min((int64_t)split * per, ntiles), t1 = min(t0 + per, ntiles);
    // the codes (and filter words) of the PQ_UNROLL tiles from tile tbx on
    constexpr int WN = M > 0 ? M / 4 : 1;
    // Rows past the end of the set (or of this block's range) are read from the last row instead: the loads stay
    // free of branches, so nothing waits for them before their first use a step later.
    const uint32_t* mask_src = p.mask ? p.mask : reinterpret_cast<const uint32_t*>(p.codes);
    auto fetch = [&](int64_t tbx, uint32_t (&wx)[PQ_UNROLL][WN], uint32_t (&mwx)[PQ_UNROLL]) {
#pragma unroll
        for (int u = 0; u < PQ_UNROLL; ++u) {
            const int64_t row = min((tbx + u) * 256 + tid, p.n - 1);
            if (M > 0) {
                if (M % 16 == 0) {
                    const uint4* src = reinterpret_cast<const uint4*>(p.codes + row * M);
#pragma unroll
                    for (int i = 0; i < M / 16; ++i) {
                        const uint4 v = src[i];
                        wx[u][4 * i] = v.x; wx[u][4 * i + 1] = v.y; wx[u][4 * i + 2] = v.z; wx[u][4 * i + 3] = v.w;
                    }
                } else if (M % 8 == 0) {
                    const uint2* src = reinterpret_cast<const uint2*>(p.codes + row * M);
#pragma unroll
                    for (int i = 0; i < M / 8; ++i) { const uint2 v = src[i]; wx[u][2 * i] = v.x; wx[u][2 * i + 1] = v.y; }
                } else {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.codes + row * M);
#pragma unroll
                    for (int i = 0; i < M / 4; ++i) wx[u][i] = src[i];
                }
            }
            mwx[u] = mask_src[p.mask ? (row >> 5) : 0];
        }
    };
    // One to three queries per pass: the codes of the coming step are on their way from HBM while this one's table
    // sums are taken, and lanes leave a row's sum early (below).  Four queries per pass keep the plain step: with
    // four sums per lane hardly any lane can leave, and the pass is bound by the LDS either way (measured on 100M
    // codes: 1.36 ms per eight queries plain, 1.58 ms with the early exit and the extra registers of the prefetch).
    constexpr bool AHEAD = Q < 4;
    uint32_t wn[PQ_UNROLL][WN], mwn[PQ_UNROLL];
    if constexpr (AHEAD) fetch(min(t0, ntiles - 1), wn, mwn);
    for (int64_t tb = t0; tb < t1; tb += PQ_UNROLL) {
        uint32_t w[PQ_UNROLL][WN];
        uint32_t mw[PQ_UNROLL];
        bool ok[PQ_UNROLL];
        if constexpr (!AHEAD) fetch(tb, wn, mwn);
#pragma unroll
        for (int u = 0; u < PQ_UNROLL; ++u) {
            ok[u] = (tb + u) < t1 && (tb + u) * 256 + tid < p.n;
            mw[u] = p.mask ? mwn[u] : 0xffffffffu;
#pragma unroll
            for (int i = 0; i < WN; ++i) w[u][i] = wn[u][i];
        }
        if constexpr (AHEAD) fetch(tb + PQ_UNROLL, wn, mwn);            // past the last step: the last row again, unused
        // ---- table sums in sub-space order, then the selection
        if constexpr (M >= 4 && AHEAD) if (!roots_always) {
            // The entries are squares, so a row's running sum only grows: once it is above the bound of every query
            // of the pass the row is out, and its lane sits out the remaining gathers.  The scan is bound by LDS
            // bank conflicts between the lanes of a gather, so every lane that leaves makes the gathers of the
            // others cheaper, and a tile whose lanes have all left skips them.  First half of the sub-spaces for
            // every tile of the step (all gathers in flight together), then two sub-spaces per test.
            constexpr int MM = M > 0 ? M : 4;
            constexpr int JA = MM / 2;
            typedef float pq_f2 __attribute__((ext_vector_type(2)));
            float acc[PQ_UNROLL][Q];
            pq_f2 a01[PQ_UNROLL], a23[PQ_UNROLL];          // Q == 4: the four sums as two packed pairs
            auto entry = [&](int u, int jj) { return (jj * ksub + ((w[u][jj >> 2] >> (8 * (jj & 3))) & 0xff)) * Q; };
            auto add = [&](int u, int jj) {
                if constexpr (Q == 4) {
                    const float4 tv = *reinterpret_cast<const float4*>(T + entry(u, jj));
                    a01[u] = a01[u] + pq_f2{tv.x, tv.y};
                    a23[u] = a23[u] + pq_f2{tv.z, tv.w};
                } else {
                    const float* tp = T + entry(u, jj);
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[u][q] = acc[u][q] + tp[q];
                }
            };
            auto sum_of = [&](int u, int q) -> float {
                if constexpr (Q == 4) return q == 0 ? a01[u].x : q == 1 ? a01[u].y : q == 2 ? a23[u].x : a23[u].y;
                else return acc[u][q];
            };
            auto in_reach = [&](int u) {
                bool r = false;
#pragma unroll
                for (int q = 0; q < Q; ++q) r = r || (sum_of(u, q) <= thr2[q]);
                return r;
            };
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u) {
                if constexpr (Q == 4) {
                    const float4 tv = *reinterpret_cast<const float4*>(T + entry(u, 0));
                    a01[u] = pq_f2{tv.x, tv.y}; a23[u] = pq_f2{tv.z, tv.w};
                } else {
                    const float* tp = T + entry(u, 0);
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[u][q] = tp[q];
                }
#pragma unroll
                for (int jj = 1; jj < JA; ++jj) add(u, jj);
            }
            bool alive[PQ_UNROLL];
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u) {
                const int64_t row = (tb + u) * 256 + tid;
                alive[u] = ok[u] && ((mw[u] >> (row & 31)) & 1u) && in_reach(u);
            }
            // Two more sub-spaces at a time for all tiles of the step.  A lane that has left keeps reading, but entry
            // 0 of the table: lanes that read one address are served together, so it no longer collides with
            // anyone, and the step stays free of divergent branches.  Its sums are not used again.
#pragma unroll
            for (int jj = JA; jj < MM; jj += 2) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < PQ_UNROLL; ++u) any = any || alive[u];
                if (__ballot(any) == 0ull) break;                       // wave-uniform: nothing left in reach
#pragma unroll
                for (int u = 0; u < PQ_UNROLL; ++u) {
                    const int e0 = alive[u] ? entry(u, jj) : 0, e1 = alive[u] ? entry(u, jj + 1) : 0;
                    if constexpr (Q == 4) {
                        const float4 t0 = *reinterpret_cast<const float4*>(T + e0), t1 = *reinterpret_cast<const float4*>(T + e1);
                        a01[u] = (a01[u] + pq_f2{t0.x, t0.y}) + pq_f2{t1.x, t1.y};
                        a23[u] = (a23[u] + pq_f2{t0.z, t0.w}) + pq_f2{t1.z, t1.w};
                    } else {
#pragma unroll
                        for (int q = 0; q < Q; ++q) acc[u][q] = (acc[u][q] + T[e0 + q]) + T[e1 + q];
                    }
                }
#pragma unroll
                for (int u = 0; u < PQ_UNROLL; ++u) alive[u] = alive[u] && in_reach(u);
            }
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u) {
                const int64_t row = (tb + u) * 256 + tid;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float a = sum_of(u, q);
                    if (__ballot(alive[u] && a <= thr2[q]) != 0ull) {
                        const float d = sqrtf(a);
                        best[q].template offer<false>(alive[u], d, (int)row, lane);
                        const float t = best[q].thr;
                        thr2[q] = fminf(thr2g[q], fmaxf(__fmaf_rn(t * t, 4e-7f, t * t), 1e-36f));
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < PQ_UNROLL; ++u) {
            const int64_t row = (tb + u) * 256 + tid;
            float acc[Q];
            if (M > 0 && Q == 4) {
                // the four queries' entries are one 16-byte read; two packed adds per sub-space (each element is
                // still its own sequential fp32 sum in sub-space order)
                typedef float pq_f2 __attribute__((ext_vector_type(2)));
                const float4 t0v = *reinterpret_cast<const float4*>(T + (w[u][0] & 0xff) * Q);
                pq_f2 a01 = pq_f2{t0v.x, t0v.y}, a23 = pq_f2{t0v.z, t0v.w};
#pragma unroll
                for (int j = 1; j < M; ++j) {
                    const float4 tv = *reinterpret_cast<const float4*>(T + (j * ksub + ((w[u][j >> 2] >> (8 * (j & 3))) & 0xff)) * Q);
                    a01 = a01 + pq_f2{tv.x, tv.y};
                    a23 = a23 + pq_f2{tv.z, tv.w};
                }
                acc[0] = a01.x; acc[1 % Q] = a01.y; acc[2 % Q] = a23.x; acc[3 % Q] = a23.y;
            } else if (M > 0) {
                const float* t0p = T + (w[u][0] & 0xff) * Q;
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = t0p[q];
#pragma unroll
                for (int j = 1; j < M; ++j) {
                    const float* tp = T + (j * ksub + ((w[u][j >> 2] >> (8 * (j & 3))) & 0xff)) * Q;
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[q] = acc[q] + tp[q];
                }
            } else {
                const uint8_t* src = p.codes + (ok[u] ? row : 0) * m;
                const float* t0p = T + src[0] * Q;
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = t0p[q];
                for (int j = 1; j < m; ++j) {
                    const float* tp = T + (j * ksub + src[j]) * Q;
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[q] = acc[q] + tp[q];
                }
            }
            const bool pass = ok[u] && ((mw[u] >> (row & 31)) & 1u);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (roots_always) {
                    const float d = sqrtf(acc[q]);
                    if (p.dist_out && ok[u] && q < qcnt) p.dist_out[(int64_t)(q0 + q) * p.n + row] = d;
                    if (k > 0) {
                        const bool above = (d > lbd[q]) || (d == lbd[q] && row > lbk[q]);
                        best[q].template offer<false>(pass && above, d, (int)row, lane);
                    }
                } else if (__ballot(pass && acc[q] <= thr2[q]) != 0ull) {
                    const float d = sqrtf(acc[q]);
                    best[q].template offer<false>(pass, d, (int)row, lane);
                    const float t = best[q].thr;
                    thr2[q] = fminf(thr2g[q], fmaxf(__fmaf_rn(t * t, 4e-7f, t * t), 1e-36f));
                }
            }
        }
    }
    if (k == 0) return;
    // one candidate list per wave; behind a bound most lists are empty, so with queue_len the waves pack what they
    // hold into the front of the query's area (in any order: the merge that follows does not need sorted input)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (q < qcnt) {
            if (p.queue_len) {
                const int cnt = __popcll(__ballot(lane < k && best[q].eid != EMPTY_IDX));     // the list's first cnt lanes
                if (cnt > 0) {
                    int at = 0;
                    if (lane == 0) at = atomicAdd(p.queue_len + q0 + q, cnt);
                    at = __builtin_amdgcn_readfirstlane(at);
                    const int64_t base = (int64_t)(q0 + q) * p.splits * PQ_WAVES * k + at;
                    if (lane < cnt) { p.part_d[base + lane] = best[q].e; p.part_i[base + lane] = best[q].eid; }
                }
            } else {
                const int64_t list = ((int64_t)(q0 + q) * p.splits + split) * PQ_WAVES + wave;
                best[q].store(p.part_d + list * k, p.part_i + list * k, lane);
            }
        }
    }
}

int pq_scan_lists_per_split() { return PQ_WAVES; }

// ---------------------------------------------------------------- IVF-PQ (index/hybrid.py)
// residual of every row against its coarse centroid (hybrid.py:151,269), element-wise fp32
__global__ void residual_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ cent, int ldc,
                                const int* __restrict__ list, int dim, int64_t n, float* __restrict__ out) {
    const int64_t total = n * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int c = (int)(i - r * dim);
        out[i] = x[r * ldx + c] - cent[(int64_t)list[r] * ldc + c];
    }
}

int launch_residual(const float* x, int64_t ldx, const float* cent, int ldc, const int* list, int dim, int64_t n,
                    float* out, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)std::min<int64_t>((n * dim + 255) / 256, 16384)), dim3(256), 0, s,
                       x, ldx, cent, ldc, list, dim, n, out);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// One block per (query, probed list): the table of the residual query q - centroid
// (hybrid.py:303-318) is built in LDS in NumPy order, then the list's codes are
// scanned one row per thread: fp32 sum in sub-space order (hybrid.py:320-326).
// The sums go to the query's candidate slots in visiting order WITHOUT their square root: the selection takes it,
// of the few candidates in reach of the k-th only (a correctly rounded root is twenty instructions per row here).
struct IvfPqScanParams {
    const float* q; int ldq; int dim;
    const float* cent; int ldc;
    const float* book; int m; int ksub; int dsub; SumPlan plan;
    const uint8_t* codes;                       // [n][m] in list order
    const int64_t* list_beg; const int64_t* list_end; int n_lists;
    const int* probes; int n_probe;
    const int64_t* cand_off; int64_t cand_stride;
    float* cand;
    // the units in list order (launch_ivf_units_by_list), or null: per unit its list, its query and the start of its
    // candidate slots in `cand`; the first unit of every block (and, behind the last block's, their number)
    const int* u_list; const int* u_q; const int64_t* u_out; const int* u_first;
};

static constexpr int IPQ1_NT = 1024;       // threads per (query, list) unit

__global__ __launch_bounds__(IPQ1_NT) void ivfpq_scan_kernel(IvfPqScanParams p) {
    extern __shared__ __align__(16) float sm[];
    float* T = sm;                               // [m][ksub]
    float* qres = T + p.m * p.ksub;              // [dim]
    const int tid = threadIdx.x;
    const int unit = blockIdx.x;
    const int qi = unit / p.n_probe, pr = unit - qi * p.n_probe;
    const int l = p.probes[unit];
    if (l < 0 || l >= p.n_lists) return;
    const int64_t pos0 = p.list_beg[l];
    const int64_t len = p.list_end[l] - pos0;
    if (len == 0) return;
    for (int i = tid; i < p.dim; i += IPQ1_NT) qres[i] = p.q[(int64_t)qi * p.ldq + i] - p.cent[(int64_t)l * p.ldc + i];
    __syncthreads();
    for (int e = tid; e < p.m * p.ksub; e += IPQ1_NT) {
        const int j = e / p.ksub;
        T[e] = np_sum_sqdiff(p.book + (int64_t)e * p.dsub, qres + j * p.dsub, p.dsub, p.plan);
    }
    __syncthreads();
    float* out = p.cand + (int64_t)qi * p.cand_stride + p.cand_off[(int64_t)qi * (p.n_probe + 1) + pr];
    if ((p.m & 3) == 0) {                        // four codes per load
        for (int64_t r = tid; r < len; r += IPQ1_NT) {
            const uint32_t* code = reinterpret_cast<const uint32_t*>(p.codes + (pos0 + r) * p.m);
            float acc = 0.0f;
            for (int j = 0; j < p.m; j += 4) {
                const uint32_t w = code[j >> 2];
                const float* Tj = T + j * p.ksub;
                const float t0 = Tj[w & 0xffu], t1 = Tj[p.ksub + ((w >> 8) & 0xffu)];
                const float t2 = Tj[2 * p.ksub + ((w >> 16) & 0xffu)], t3 = Tj[3 * p.ksub + (w >> 24)];
                acc = j == 0 ? t0 : acc + t0;
                acc = acc + t1;
                acc = acc + t2;
                acc = acc + t3;
            }
            out[r] = acc;
        }
        return;
    }
    for (int64_t r = tid; r < len; r += IPQ1_NT) {
        const uint8_t* code = p.codes + (pos0 + r) * p.m;
        float acc = T[code[0]];
        for (int j = 1; j < p.m; ++j) acc = acc + T[j * p.ksub + code[j]];
        out[r] = acc;
    }
}

// The same with the codebooks resident in LDS (m * ksub * dsub floats, 128 KiB at m = 16, dsub = 8): one block of
// sixteen waves per CU walks the (query, probed list) units.  A unit is a chain of dependent reads (probed list -> its
// bounds, its centroid -> its codes) in front of a microsecond of LDS work, and the codebooks leave room for one block
// per CU, so the chain is taken apart over the units: while unit i is summed, the codes of unit i + 1 (the first 3072
// rows, in registers), the bounds, query and centroid elements of unit i + 2 and the list number of unit i + 3 are on
// their way.  (One block per unit, nothing read ahead: 10 us per unit and CU, 1.3 ms per 1024 x 32 units.)
static constexpr int IPQ_NT = 1024;        // threads per block
static constexpr int IPQ_AHEAD = 3;        // rows per thread whose codes are read a unit ahead

template <int M>
__global__ __launch_bounds__(IPQ_NT) void ivfpq_scan_lds_kernel(IvfPqScanParams p, int n_units) {
    extern __shared__ __align__(16) float sm[];
    float* book = sm;                            // [M][dsub][ksub]: codeword-fastest, so the lanes of a table step read neighbours
    float* T = book + M * p.ksub * p.dsub;       // [M][ksub]
    float* qres = T + M * p.ksub;                // [dim]
    const int tid = threadIdx.x;
    const int book_n = M * p.ksub * p.dsub;
    for (int i = tid; i < book_n; i += IPQ_NT) {
        const int jc = i / p.dsub, d = i - jc * p.dsub;
        const int j = jc / p.ksub, c = jc - j * p.ksub;
        book[(j * p.dsub + d) * p.ksub + c] = p.book[i];
    }
    constexpr int W = M / 4;
    struct Named { int l, qi, pr; };                                     // a unit's list (-1: no such unit, or no such list)
    struct Placed { int64_t pos0, len, out; float q_el, c_el; };         // ... its rows, its output, this thread's elements
    auto name_of = [&](int64_t unit) {
        Named a{-1, 0, 0};
        if (unit < n_units) {
            a.qi = (int)(unit / p.n_probe);
            a.pr = (int)(unit - (int64_t)a.qi * p.n_probe);
            const int l = p.probes[unit];
            a.l = (l >= 0 && l < p.n_lists) ? l : -1;
        }
        return a;
    };
    auto place_of = [&](const Named& a) {
        Placed b{0, 0, 0, 0.0f, 0.0f};
        if (a.l >= 0) {
            b.pos0 = p.list_beg[a.l];
            b.len = p.list_end[a.l] - b.pos0;
            b.out = (int64_t)a.qi * p.cand_stride + p.cand_off[(int64_t)a.qi * (p.n_probe + 1) + a.pr];
            if (tid < p.dim) {
                b.q_el = p.q[(int64_t)a.qi * p.ldq + tid];
                b.c_el = p.cent[(int64_t)a.l * p.ldc + tid];
            }
        }
        return b;
    };
    auto codes_of = [&](const Placed& b, uint32_t (&w)[IPQ_AHEAD][W]) {
#pragma unroll
        for (int x = 0; x < IPQ_AHEAD; ++x) {
            const int64_t r = tid + x * IPQ_NT;
            if (r < b.len) {
                const uint32_t* code = reinterpret_cast<const uint32_t*>(p.codes + (b.pos0 + r) * M);
#pragma unroll
                for (int i = 0; i < W; ++i) w[x][i] = code[i];
            }
        }
    };
    auto distance_of = [&](const uint32_t (&w)[W]) {
        float acc = T[w[0] & 0xffu];
#pragma unroll
        for (int j = 1; j < M; ++j) acc = acc + T[j * p.ksub + ((w[j >> 2] >> (8 * (j & 3))) & 0xffu)];
        return acc;
    };
    const int64_t step = gridDim.x;
    int64_t unit = blockIdx.x;
    Named n0 = name_of(unit), n1 = name_of(unit + step), named = name_of(unit + 2 * step);
    Placed cur = place_of(n0), placed = place_of(n1);
    uint32_t w_cur[IPQ_AHEAD][W] = {}, w_next[IPQ_AHEAD][W] = {};
    codes_of(cur, w_cur);
    for (; unit < n_units; unit += step) {
        __syncthreads();                         // the table and the residual of the unit before are no longer read
        if (tid < p.dim) qres[tid] = cur.q_el - cur.c_el;
        // what the coming units need, on its way while this one is summed
        codes_of(placed, w_next);
        const Placed placed_next = place_of(named);
        const Named named_next = name_of(unit + 3 * step);
        __syncthreads();
        if (cur.len > 0) {                       // (the same for every thread of the block)
            for (int e = tid; e < M * p.ksub; e += IPQ_NT) {
                const int j = e / p.ksub, c = e - j * p.ksub;
                T[e] = np_leaf_sqdiff_strided(book + j * p.dsub * p.ksub + c, p.ksub, qres + j * p.dsub, p.dsub);   // dsub <= 128: one leaf
            }
        }
        __syncthreads();
        if (cur.len > 0) {
            float* out = p.cand + cur.out;
#pragma unroll
            for (int x = 0; x < IPQ_AHEAD; ++x) {
                const int64_t r = tid + x * IPQ_NT;
                if (r < cur.len) out[r] = distance_of(w_cur[x]);
            }
            for (int64_t r = tid + IPQ_AHEAD * IPQ_NT; r < cur.len; r += IPQ_NT) {
                const uint32_t* code = reinterpret_cast<const uint32_t*>(p.codes + (cur.pos0 + r) * M);
                uint32_t w[W];
#pragma unroll
                for (int i = 0; i < W; ++i) w[i] = code[i];
                out[r] = distance_of(w);
            }
        }
        cur = placed;
        placed = placed_next;
        named = named_next;
#pragma unroll
        for (int x = 0; x < IPQ_AHEAD; ++x)
#pragma unroll
            for (int i = 0; i < W; ++i) w_cur[x][i] = w_next[x][i];
    }
}

// The same again with two tables: while unit i is summed from one, the table of unit i + 1 is built in the other, so
// the table arithmetic and the LDS gathers of the sums run side by side and a unit costs one barrier instead of three.
// The second table takes the room of the residual query, which moves into registers: with 256 codewords and 1024
// threads a wave builds entries of a few sub-spaces only, so lane x * DSUB + d holds element d of the residual in the
// wave's x-th sub-space and the build reads it with a lane broadcast.  Both halves work on two values at a time with
// packed fp32 instructions (the IEEE operations of the one-value code, two per instruction): the build on codewords
// c and c + 128 of a sub-space, the sums on rows r and r + 64.  At m = 16, dsub = 8 the codebooks and the two tables
// fill the 160 KiB of LDS to the byte.
typedef float v2f __attribute__((ext_vector_type(2)));

// np.sum((a - b)**2) over DSUB elements for one codeword (V = float) or two at once (V = v2f, packed fp32 instructions:
// the same IEEE operations in the same order, two results per instruction); a(i) is element i of the codeword(s)
template <int DSUB, typename V, typename A>
__device__ __forceinline__ static V np_leaf_sqdiff_cols(A a, const float (&b)[DSUB]) {
    if constexpr (DSUB < 8) {
        V r = V(0.0f);
#pragma unroll
        for (int i = 0; i < DSUB; ++i) { const V d = a(i) - V(b[i]); r = r + d * d; }
        return r;
    } else {
        V r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const V d = a(j) - V(b[j]); r[j] = d * d; }
#pragma unroll
        for (int i = 8; i + 8 <= DSUB; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const V d = a(i + j) - V(b[i + j]); r[j] = r[j] + d * d; }
        }
        V res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = DSUB / 8 * 8; i < DSUB; ++i) { const V d = a(i) - V(b[i]); res = res + d * d; }
        return res;
    }
}

static constexpr int IPQ2_AHEAD = 4;       // rows per thread whose codes are read a unit ahead, with two tables

template <int M, int DSUB, bool BY_LIST>
__global__ __launch_bounds__(IPQ_NT) void ivfpq_scan_lds2_kernel(IvfPqScanParams p, int n_units) {
    constexpr int KS = 256;                      // codewords per sub-space (checked by the launcher)
    constexpr int W = M / 4;                     // 32-bit words of a row's codes
    constexpr int XS = M * KS / IPQ_NT;          // table entries per thread, one sub-space each
    constexpr int XP = XS >= 2 ? XS / 2 : 1;    // ... as pairs of codewords c, c + 128 of one sub-space where there are two
    constexpr int JW = XS >= 2 ? 7 : 8;          // a thread's sub-spaces are (tid >> JW) + (IPQ_NT >> JW) * x, x < XP
    constexpr int RL = XP * DSUB;                // lanes of a wave that hold an element of the residual
    static_assert(RL <= 64 && XS >= 1 && (XS == 1 || XS % 2 == 0), "the residual of a wave's sub-spaces must fit its lanes");
    constexpr int PAIRS = IPQ2_AHEAD / 2;
    static_assert(IPQ2_AHEAD % 2 == 0, "rows are summed two at a time");
    extern __shared__ __align__(16) float sm[];
    float* tables = sm;                          // [2][M][KS], at LDS address 0 (the kernel has no other LDS; the launcher
                                                 // checks): a table read is a byte * 4 plus a constant offset
    typedef __attribute__((address_space(3))) const float lds_float;
    float* book = tables + 2 * M * KS;           // [M][DSUB][KS], codeword-fastest
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < M * KS * DSUB; i += IPQ_NT) {
        const int jc = i / DSUB, d = i - jc * DSUB;
        const int j = jc / KS, c = jc - j * KS;
        book[(j * DSUB + d) * KS + c] = p.book[i];
    }
    const int res_col = ((tid >> JW) + (IPQ_NT >> JW) * (lane / DSUB)) * DSUB + lane % DSUB;     // (for lane < RL)
    // the rows a thread sums, two at a time so that the gathers of one hide behind the other's: pair x of wave w holds
    // rows x * 2048 + w * 128 + lane and 64 further on (a wave is then either busy in both or, at a list's end, idle)
    const int row0 = (tid >> 6) * 128 + lane;
    // What a unit needs from memory is read with per-lane loads of one address, not scalar loads: scalar loads share
    // their counter with the LDS reads and return in any order, so the first wait for a table entry would wait for
    // every scalar load on its way as well, a trip to memory per unit that nothing hides.  The zero below is one the
    // compiler has to take for a per-lane value.
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    auto lane_load = [&](auto* at) {             // *at, at the same for every lane
        typedef std::remove_reference_t<decltype(*at)> T;
        return *reinterpret_cast<T*>(reinterpret_cast<const char*>(at) + (uint32_t)lane_zero);
    };
    auto uniform64 = [](int64_t v) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    struct Named { int l, qi, pr; int64_t at; };                         // a unit's list as read (per lane; -1: no such unit)
    struct Placed { int64_t pos0, end, off; float q_el, c_el; int qi, l; bool ok; };   // its bounds and output as read (per lane)
    struct Ready { int64_t pos0; int len; int64_t out; int l; };        // ... and as the block uses them a unit later
    // BY_LIST: the units come in list order, a block takes a run of them, and a unit of the list of the unit before it
    // keeps that unit's codes (a list is probed by several queries of a batch: its codes are then read once per
    // block instead of once per query).  Otherwise unit u = query u / n_probe, probe u % n_probe, dealt round-robin.
    constexpr bool by_list = BY_LIST;
    int64_t unit, last;                          // this block's units: unit, unit + step, ... < last
    int step;
    if (by_list) {
        unit = __builtin_amdgcn_readfirstlane(lane_load(p.u_first + blockIdx.x));
        last = __builtin_amdgcn_readfirstlane(lane_load(p.u_first + blockIdx.x + 1));
        step = 1;
    } else {
        unit = blockIdx.x;
        last = n_units;
        step = (int)gridDim.x;
    }
    // the unit three ahead of the one being summed, counted on without a division per unit
    const int step_q = step / p.n_probe, step_r = step - step_q * p.n_probe;
    int64_t ahead = unit;
    int ahead_qi = (int)((uint32_t)unit / (uint32_t)p.n_probe), ahead_pr = (int)unit - ahead_qi * p.n_probe;
    auto name_next = [&]() {
        Named a{-1, ahead_qi, ahead_pr, ahead};
        if (ahead < last) {
            if (by_list) {
                a.l = lane_load(p.u_list + ahead);
                a.qi = lane_load(p.u_q + ahead);
            } else {
                a.l = lane_load(p.probes + ahead);
            }
        }
        ahead += step;
        ahead_qi += step_q;
        ahead_pr += step_r;
        if (ahead_pr >= p.n_probe) { ahead_pr -= p.n_probe; ++ahead_qi; }
        return a;
    };
    auto place_of = [&](const Named& a) {
        const int l = __builtin_amdgcn_readfirstlane(a.l);
        const int qi = by_list ? __builtin_amdgcn_readfirstlane(a.qi) : a.qi;
        Placed b{0, 0, 0, 0.0f, 0.0f, qi, l, false};
        b.ok = l >= 0 && l < p.n_lists;
        if (b.ok) {
            b.pos0 = lane_load(p.list_beg + l);
            b.end = lane_load(p.list_end + l);
            b.off = by_list ? lane_load(p.u_out + a.at) : lane_load(p.cand_off + ((int64_t)qi * (p.n_probe + 1) + a.pr));
            if (lane < RL) {
                b.q_el = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.q + (int64_t)qi * p.ldq) + (uint32_t)(res_col * 4));
                b.c_el = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.cent + (int64_t)l * p.ldc) + (uint32_t)(res_col * 4));
            }
        }
        return b;
    };
    auto ready_of = [&](const Placed& b) {
        Ready r{0, 0, 0, -1};
        if (b.ok) {
            r.pos0 = uniform64(b.pos0);
            r.len = (int)(uniform64(b.end) - r.pos0);                 // (a device holds fewer than 2^31 rows)
            r.out = by_list ? uniform64(b.off) : (int64_t)b.qi * p.cand_stride + uniform64(b.off);
            r.l = b.l;
        }
        return r;
    };
    auto codes_of = [&](const Ready& b, uint32_t (&w)[IPQ2_AHEAD][W]) {
        const uint8_t* rows = p.codes + b.pos0 * M;
#pragma unroll
        for (int x = 0; x < IPQ2_AHEAD; ++x) {
            const int r = (x >> 1) * (2 * IPQ_NT) + row0 + (x & 1) * 64;
            if (r < b.len) {
                const uint32_t* code = reinterpret_cast<const uint32_t*>(rows + (uint32_t)(r * M));
#pragma unroll
                for (int i = 0; i < W; ++i) w[x][i] = code[i];
            } else {
#pragma unroll
                for (int i = 0; i < W; ++i) w[x][i] = 0u;             // (summed all the same beside a row that is there)
            }
        }
    };
    auto table_of = [&](const Placed& b, float* T) {
        const int res = __float_as_int(b.q_el - b.c_el);
#pragma unroll
        for (int x = 0; x < XP; ++x) {
            float el[DSUB];
#pragma unroll
            for (int d = 0; d < DSUB; ++d) el[d] = __int_as_float(__builtin_amdgcn_readlane(res, x * DSUB + d));
            const int j = (tid >> JW) + (IPQ_NT >> JW) * x;
            if constexpr (XS >= 2) {
                const int c = tid & 127;
                const float* a = book + j * DSUB * KS + c;
                const v2f t = np_leaf_sqdiff_cols<DSUB, v2f>([&](int i) { return v2f{a[i * KS], a[i * KS + 128]}; }, el);
                T[j * KS + c] = t.x;
                T[j * KS + c + 128] = t.y;
            } else {
                const int c = tid & 255;
                const float* a = book + j * DSUB * KS + c;
                T[j * KS + c] = np_leaf_sqdiff_cols<DSUB, float>([&](int i) { return a[i * KS]; }, el);
            }
        }
    };
    auto byte_of = [](const uint32_t (&w)[W], int j) { return (w[j >> 2] >> (8 * (j & 3))) & 0xffu; };
    const uint32_t two = 2u;
    auto quad_of = [&](uint32_t w, int k) {      // 4 * byte k of w: a table offset in one instruction
        uint32_t r;
        if (k == 0)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
        else if (k == 1)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
        else if (k == 2)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
        else
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w));
        return r;
    };
    const Named n0 = name_next(), n1 = name_next();
    Named named = name_next();
    const Placed first = place_of(n0);
    Placed placed = place_of(n1);
    Ready cur = ready_of(first);
    uint32_t w_a[IPQ2_AHEAD][W] = {}, w_b[IPQ2_AHEAD][W] = {};
    codes_of(cur, w_a);
    __syncthreads();                             // the codebooks are in place
    if (cur.len > 0) table_of(first, tables);
    __syncthreads();
    // one unit, summed from table SIDE while the next one's table is built in the other
    auto unit_of = [&](auto side_tag, const uint32_t (&w_cur)[IPQ2_AHEAD][W], uint32_t (&w_next)[IPQ2_AHEAD][W]) {
        constexpr int SIDE = decltype(side_tag)::value;
        const float* T = tables + SIDE * (M * KS);
        const Ready next = ready_of(placed);
        auto build = [&]() {
            if (next.len > 0) table_of(placed, tables + (SIDE ^ 1) * (M * KS));   // (the same for every thread of the block)
        };
        auto sum_rows = [&]() {
        if (cur.len > 0) {
            float* out = p.cand + cur.out;
#pragma unroll
            for (int x = 0; x < PAIRS; ++x) {
                const int r = x * (2 * IPQ_NT) + row0;
                if (r < cur.len) {
                    // two rows side by side: one packed addition per sub-space
                    uint32_t at[2][M];
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        at[0][j] = quad_of(w_cur[2 * x][j >> 2], j & 3);
                        at[1][j] = quad_of(w_cur[2 * x + 1][j >> 2], j & 3);
                    }
                    v2f t[M];
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        t[j].x = *(lds_float*)(uintptr_t)(at[0][j] + (uint32_t)((SIDE * M + j) * KS * sizeof(float)));
                        t[j].y = *(lds_float*)(uintptr_t)(at[1][j] + (uint32_t)((SIDE * M + j) * KS * sizeof(float)));
                    }
                    v2f sum = t[0];
#pragma unroll
                    for (int j = 1; j < M; ++j) sum = sum + t[j];
                    const float a = sum.x, b = sum.y;
                    const bool both = r + 64 < cur.len;
                    const float da = a, db = b;
                    // (one value twice rather than a branch round the second row)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (uint32_t)(r * 4)) = da;
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (uint32_t)((both ? r + 64 : r) * 4)) = both ? db : da;
                }
            }
            for (int r = tid + IPQ2_AHEAD * IPQ_NT; r < cur.len; r += IPQ_NT) {
                const uint32_t* code = reinterpret_cast<const uint32_t*>(p.codes + (cur.pos0 + r) * M);
                uint32_t w[W];
#pragma unroll
                for (int i = 0; i < W; ++i) w[i] = code[i];
                float a = T[byte_of(w, 0)];
#pragma unroll
                for (int j = 1; j < M; ++j) a = a + T[j * KS + byte_of(w, j)];
                out[r] = a;
            }
        }
        };
        build();
        __builtin_amdgcn_sched_barrier(0);
        // what the coming units need, on its way while this one is summed (asked for here, not at the top, where
        // all sixteen waves would ask at once, straight after the barrier)
        if (by_list && next.l == cur.l && cur.len > 0) {       // the same list again: its codes are here
#pragma unroll
            for (int x = 0; x < IPQ2_AHEAD; ++x)
#pragma unroll
                for (int i = 0; i < W; ++i) w_next[x][i] = w_cur[x][i];
        } else {
            codes_of(next, w_next);
        }
        const Placed placed_next = place_of(named);
        const Named named_next = name_next();
        __builtin_amdgcn_sched_barrier(0);
        sum_rows();
        __syncthreads();                         // the next table is whole, this one is no longer read
        cur = next;
        placed = placed_next;
        named = named_next;
        unit += step;
    };
    while (unit < last) {                        // (the two sets of code registers change places, like the two tables)
        unit_of(std::integral_constant<int, 0>{}, w_a, w_b);
        if (unit >= last) break;
        unit_of(std::integral_constant<int, 1>{}, w_b, w_a);
    }
}

int ivfpq_scan_blocks() { return ctx().num_cus; }

// whether a scan of these shapes goes to the two-table kernel (which can take its units in list order)
bool ivfpq_scan_takes_units(int m, int ksub, int dsub, int dim, int nq, int n_probe) {
    static const bool off = getenv("VDB_IVFPQ_NO_LDS_BOOK") != nullptr || getenv("VDB_IVFPQ_ONE_TABLE") != nullptr;
    const int64_t n_units = (int64_t)nq * n_probe;
    const size_t lds_two = ((size_t)m * ksub * dsub + 2 * (size_t)m * ksub) * sizeof(float);
    return !off && n_units >= 4 * (int64_t)ctx().num_cus && n_units < (int64_t)0x7fffffff && ksub == 256 && lds_two <= 160 * 1024 &&
           (m == 4 || m == 8 || m == 16) && (dsub == 4 || dsub == 8 || dsub == 16) && m * dsub == dim;
}

int launch_ivfpq_scan(const float* q, int ldq, int dim, int nq, const float* cent, int ldc, const float* book, int m, int ksub,
                      int dsub, const SumPlan& plan, const uint8_t* codes, const int64_t* list_beg, const int64_t* list_end, int n_lists,
                      const int* probes, int n_probe, const int64_t* cand_off, int64_t cand_stride, float* cand, hipStream_t s,
                      const IvfPqUnits* units) {
    if (nq <= 0) return VDB_OK;
    IvfPqScanParams p{q, ldq, dim, cent, ldc, book, m, ksub, dsub, plan, codes, list_beg, list_end, n_lists, probes, n_probe,
                      cand_off, cand_stride, cand, nullptr, nullptr, nullptr, nullptr};
    const size_t lds = ((size_t)m * ksub + dim) * sizeof(float);
    VDB_REQUIRE(lds <= 160 * 1024, "%d sub-spaces x %d codewords need %zu bytes of LDS", m, ksub, lds);
    const size_t lds_book = lds + (size_t)m * ksub * dsub * sizeof(float);
    static const bool no_lds_book = getenv("VDB_IVFPQ_NO_LDS_BOOK") != nullptr;
    const int64_t n_units = (int64_t)nq * n_probe;
    const size_t lds_two = ((size_t)m * ksub * dsub + 2 * (size_t)m * ksub) * sizeof(float);
    VDB_REQUIRE(units == nullptr || ivfpq_scan_takes_units(m, ksub, dsub, dim, nq, n_probe), "units in list order are for the two-table scan");
    if (ivfpq_scan_takes_units(m, ksub, dsub, dim, nq, n_probe)) {
        if (units) { p.u_list = units->list; p.u_q = units->query; p.u_out = units->out; p.u_first = units->first; }
        const dim3 grid((unsigned)ctx().num_cus);
        KernelTimer timer(s);
        bool launched = true;
#define VDB_LAUNCH_IPQ2(MM, DD)                                                                               \
    if (m == MM && dsub == DD) {                                                                              \
        const void* kern = units ? reinterpret_cast<const void*>(ivfpq_scan_lds2_kernel<MM, DD, true>)        \
                                 : reinterpret_cast<const void*>(ivfpq_scan_lds2_kernel<MM, DD, false>);      \
        hipFuncAttributes fa;                                                                                 \
        VDB_HIP(hipFuncGetAttributes(&fa, kern));                                                             \
        VDB_REQUIRE(fa.sharedSizeBytes == 0, "the tables must start the kernel's LDS");                       \
        VDB_TRY(raise_dynamic_lds(kern, lds_two));                                                            \
        if (units) hipLaunchKernelGGL((ivfpq_scan_lds2_kernel<MM, DD, true>), grid, dim3(IPQ_NT), lds_two, s, p, (int)n_units);   \
        else hipLaunchKernelGGL((ivfpq_scan_lds2_kernel<MM, DD, false>), grid, dim3(IPQ_NT), lds_two, s, p, (int)n_units);        \
    } else
        VDB_LAUNCH_IPQ2(4, 4) VDB_LAUNCH_IPQ2(4, 8) VDB_LAUNCH_IPQ2(4, 16)
        VDB_LAUNCH_IPQ2(8, 4) VDB_LAUNCH_IPQ2(8, 8) VDB_LAUNCH_IPQ2(8, 16)
        VDB_LAUNCH_IPQ2(16, 4) VDB_LAUNCH_IPQ2(16, 8) launched = false;
#undef VDB_LAUNCH_IPQ2
        if (launched) {
            VDB_HIP(hipGetLastError());
            return VDB_OK;
        }
    }
    if (!no_lds_book && lds_book <= 160 * 1024 && (m == 4 || m == 8 || m == 16 || m == 32) && dim <= IPQ_NT && dsub <= 128 && n_units >= 4 * (int64_t)ctx().num_cus &&
        n_units < (int64_t)0x7fffffff) {
        const dim3 grid((unsigned)ctx().num_cus);
        KernelTimer timer(s);
#define VDB_LAUNCH_IPQ(MM)                                                                                   \
    case MM:                                                                                                  \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(ivfpq_scan_lds_kernel<MM>), lds_book));       \
        hipLaunchKernelGGL(ivfpq_scan_lds_kernel<MM>, grid, dim3(IPQ_NT), lds_book, s, p, (int)n_units);         \
        break;
        switch (m) {
            VDB_LAUNCH_IPQ(4)
            VDB_LAUNCH_IPQ(8)
            VDB_LAUNCH_IPQ(16)
            VDB_LAUNCH_IPQ(32)
        }
#undef VDB_LAUNCH_IPQ
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(ivfpq_scan_kernel),
                                lds));
    KernelTimer timer(s);
    hipLaunchKernelGGL(ivfpq_scan_kernel, dim3((unsigned)((int64_t)nq * n_probe)), dim3(IPQ1_NT), lds, s, p);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// ---------------------------------------------------------------- launchers
int launch_pq_lut(const float* q, int ldq, int nq, const float* book, int m, int ksub, int dsub,
                  const SumPlan& plan, float* lut, hipStream_t s, int* zero_counts, float* mins, int* zero_more, int zero_more_n) {
    if (nq <= 0) return VDB_OK;
    hipLaunchKernelGGL(pq_lut_kernel, dim3(m, nq), dim3(256), (size_t)dsub * sizeof(float), s,
                       q, ldq, book, m, ksub, dsub, plan, lut, zero_counts, mins, zero_more, zero_more_n);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_pq_encode(const float* x, int64_t n, int ldx, const float* book, int m, int ksub, int dsub,
                     const SumPlan& plan, uint8_t* codes, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    const size_t cb_bytes = (size_t)ksub * dsub * sizeof(float);
    const size_t x_bytes = 256 * (size_t)(dsub + 1) * sizeof(float);
    const size_t budget = 96 * 1024;
    const bool cb_lds = cb_bytes <= budget;
    const bool x_lds = (cb_lds ? cb_bytes : 0) + x_bytes <= budget;
    const size_t lds = (cb_lds ? cb_bytes : 0) + (x_lds ? x_bytes : 0);
    const dim3 grid((unsigned)((n + 255) / 256), m);
#define VDB_LAUNCH_ENC(C, X)                                                                   \
    {                                                                                          \
        auto kern = pq_encode_kernel<C, X>;                                                    \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern),                       \
                                    lds));    \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, n, ldx, book, m, ksub, dsub, plan, codes); \
    }
    if (cb_lds && x_lds) VDB_LAUNCH_ENC(true, true)
    else if (cb_lds) VDB_LAUNCH_ENC(true, false)
    else if (x_lds) VDB_LAUNCH_ENC(false, true)
    else VDB_LAUNCH_ENC(false, false)
#undef VDB_LAUNCH_ENC
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_pq_scan(const uint8_t* codes, int64_t n, int m, int ksub, const float* lut, int nq, int k,
                   const uint32_t* mask, const float* lb_d, const int64_t* lb_k, int splits,
                   float* part_d, int* part_i, float* dist_out, hipStream_t s, const float* bound_d, int64_t bound_stride,
                   bool timed, int* queue_len, const int* run_if, int max_q) {
    VDB_REQUIRE(k >= 0 && k <= 64, "scan supports k <= 64, got %d", k);
    VDB_REQUIRE(n < (int64_t)0x7fffffff, "at most 2^31-1 codes per device");
    PqScanParams p{codes, n, m, ksub, lut, nq, mask, lb_d, lb_k, k, splits, part_d, part_i, queue_len, dist_out, bound_d, bound_stride, run_if};
    const size_t table = (size_t)m * ksub * sizeof(float);
    VDB_REQUIRE(table <= 160 * 1024, "%d sub-spaces x %d codewords need %zu bytes of LDS", m, ksub, table);
    // queries per pass: as many tables as fit while two blocks still share a CU
    int Q = 1;
    if (nq >= 4 && 4 * table <= 80 * 1024) Q = 4;
    else if (nq >= 2 && 2 * table <= 80 * 1024) Q = 2;
    const bool fixed_m = m == 4 || m == 8 || m == 16 || m == 32 || m == 64;
    if (!fixed_m) Q = 1;
    Q = std::min(Q, std::max(max_q, 1));
    const size_t lds = table * Q;
    const dim3 grid(splits, (nq + Q - 1) / Q);
    KernelTimer timer(s, timed);
#define VDB_LAUNCH_PQ(MM, QQ)                                                                  \
    {                                                                                          \
        auto kern = ksub == 256 ? pq_scan_kernel<MM, QQ, 256> : pq_scan_kernel<MM, QQ, 0>;     \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));                  \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);                                  \
    }
#define VDB_LAUNCH_PQ_M(MM)                                                                    \
    case MM:                                                                                   \
        if (Q == 4) VDB_LAUNCH_PQ(MM, 4) else if (Q == 2) VDB_LAUNCH_PQ(MM, 2) else VDB_LAUNCH_PQ(MM, 1) \
        break;
    switch (m) {
        VDB_LAUNCH_PQ_M(4)
        VDB_LAUNCH_PQ_M(8)
        VDB_LAUNCH_PQ_M(16)
        VDB_LAUNCH_PQ_M(32)
        VDB_LAUNCH_PQ_M(64)
        default: VDB_LAUNCH_PQ(0, 1) break;
    }
#undef VDB_LAUNCH_PQ_M
#undef VDB_LAUNCH_PQ
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
