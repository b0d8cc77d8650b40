// PQ ADC scan for batches: up to eight queries per pass over the codes on one-byte tables, the survivors recomputed
// with the fp32 tables in the reference's arithmetic (index/pq.py:285-297: sequential fp32 sum over the sub-spaces,
// sqrt), so the results are those of pq_scan_kernel bit for bit.
//
// Why: with fp32 tables a pass for four queries reads 16 bytes of LDS per (row, sub-space) and is bound by the LDS
// (bank conflicts between the lanes of a gather).  One-byte entries put eight queries into one 8-byte LDS read and
// make the sums packed 16-bit adds.
//
// The bytes are lower bounds.  For query q with a distance B that k rows of the set are known not to exceed
// (the bound pass of the caller), a result row has sqrtf(S) <= B for its fp32 sum S, hence S <= thr (B squared, a
// few ulps up), and the real-number sum of its table entries is at most thr * (1 + 32 * 2^-24).  With
// min_j = the smallest entry of sub-space j,  thr' = thr * (1 + 4e-6) - sum_j min_j  and  step = thr' / 510,  the
// byte of an entry t is  clamp(floor((t - min_j) / step - 1e-9), 0, 255)  (float64; the 1e-9 covers the rounding of
// the quotient), so  byte * step <= t - min_j  and for a result row  sum_j byte_j <= thr' / step = 510: a row whose
// byte sum is above 510 is in no query's result and is dropped.  An entry of half of thr' and more saturates at 255
// (still a lower bound).  Flooring loses less than one step per sub-space, m / 1020 of thr' on average.
// Rows at or below 510 are queued per wave with the bits of the queries they passed for; pq_exact8_kernel takes the
// fp32 sums of exactly those and hands (distance, row) to the per-query selection.
//
// A query without a finite bound, a full wave segment or a full candidate area set the overflow flag: the caller's
// fp32 scan then runs for the batch (it is launched behind the flag either way).
#include <type_traits>
#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

typedef unsigned short pq_u16x2 __attribute__((ext_vector_type(2)));

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

static constexpr int P8_Q = 8;             // queries per pass
static constexpr int P8_LIMIT = 510;       // largest byte sum of a row that may be in a result (m * 255 < 2^16 for the sums)
static constexpr int P8_UNROLL = 2;        // row tiles of 256 whose code loads and table reads are in flight together
static constexpr int P8_DEPTH = 4;         // steps (of P8_UNROLL tiles) whose codes are in registers or on their way

// ---------------------------------------------------------------- byte tables
struct PqQuantParams {
    const float* lut; int nq; int m;            // [nq][m][256]
    const float* lut_mins;                      // [nq][m]
    const float* bound_d; int64_t bound_stride; // query q's k-th best distance is not above bound_d[q * bound_stride]
    uint2* t8;                                  // [groups][m][256]: byte b of an entry = query 8 * group + b
    float* thr;                                 // [nq]: the bound on the fp32 sum
    float* bound_keep;                          // [nq]: the bounds themselves, for a scan that follows (bound_d may be overwritten)
    int* overflow;
};

// one block per (group of eight queries, sub-space); thread c owns codeword c
__global__ __launch_bounds__(256) void pq_quantise_kernel(PqQuantParams p) {
    __shared__ double inv_step[P8_Q];
    __shared__ float least[P8_Q];
    const int g = blockIdx.x, j = blockIdx.y, c = threadIdx.x;
    const int m = p.m;
    if (c < P8_Q) {
        const int q = g * P8_Q + c;
        double inv = 0.0;                       // 0: a query that is not there (its bytes are 255)
        float mine = 0.0f;
        if (q < p.nq) {
            const float b = p.bound_d[(int64_t)q * p.bound_stride];
            const float t2 = fmaxf(__fmaf_rn(b * b, 4e-7f, b * b), 1e-36f);       // what the fp32 scan compares its sums with
            double rest = (double)t2 * (1.0 + 4e-6);
            for (int jj = 0; jj < m; ++jj) rest -= (double)p.lut_mins[(int64_t)q * m + jj];
            mine = p.lut_mins[(int64_t)q * m + j];
            const double step = rest / (double)P8_LIMIT;
            const bool usable = (b == b) && (t2 < INFINITY) && (step > 0.0) && (step < 1e300);
            if (j == 0) {
                p.thr[q] = t2;
                p.bound_keep[q] = b;
                // no usable bound: fewer than k rows passed the filter in the bound pass, NaN or infinite tables, or
                // a bound below the sum of the minima (which only rounding can produce)
                if (!usable) *p.overflow = 1;
            }
            inv = usable ? 1.0 / step : -1.0;   // -1: every byte 0
        }
        inv_step[c] = inv;
        least[c] = mine;
    }
    __syncthreads();
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int qi = 0; qi < P8_Q; ++qi) {
        const int q = g * P8_Q + qi;
        const double inv = inv_step[qi];
        uint32_t byte = 255u;
        if (inv < 0.0) byte = 0u;
        else if (inv > 0.0) {
            const double v = ((double)p.lut[((int64_t)q * m + j) * 256 + c] - (double)least[qi]) * inv;
            if (v < 256.0) byte = (uint32_t)max(0, (int)floor(v - 1e-9));
            else if (!(v == v)) byte = 0u;                  // NaN entry: the fp32 sums decide
        }
        w[qi >> 2] |= byte << (8 * (qi & 3));
    }
    p.t8[((int64_t)g * m + j) * 256 + c] = make_uint2(w[0], w[1]);
}

// ---------------------------------------------------------------- scan
struct PqScan8Params {
    const uint8_t* codes; int64_t n; int splits;
    const uint2* t8;
    const uint32_t* mask;
    int2* seg; int* seg_len; int seg_cap;       // one segment per wave: (row, bits of the group's queries it may belong to)
    int* overflow;
};

// 8 * (byte B of w): the byte offset of a code's entry in its sub-space's table, in one instruction (the byte
// select of the SDWA encoding; `three` is a register that holds 3)
template <int B>
__device__ __forceinline__ uint32_t code_offset(uint32_t w, uint32_t three) {
    uint32_t r;
    if constexpr (B == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
    else if constexpr (B == 1)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
    else if constexpr (B == 2)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
    return r;
}

// eight bytes of LDS at a byte offset from the start of the workgroup's LDS, and the offset of a pointer into it
__device__ __forceinline__ uint2 lds_entry_at(uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const unsigned long long lds_u64;
    const unsigned long long v = *reinterpret_cast<lds_u64*>((uintptr_t)off);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
#else
    (void)off;
    return make_uint2(0u, 0u);
#endif
}
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
#else
    (void)p;
    return 0u;
#endif
}

template <int M, bool MASKED>
__global__ __launch_bounds__(256) void pq_scan8_kernel(PqScan8Params p) {
    extern __shared__ __align__(16) uint2 T8[];             // [M][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, g = blockIdx.y;
    const int wave_id = (g * p.splits + split) * 4 + wave;
    if (*p.overflow != 0) {                                  // the fp32 scan answers this batch
        if (lane == 0) p.seg_len[wave_id] = 0;
        return;
    }
    // the table reads below address the LDS by offset alone (one instruction per code); this kernel has no other LDS
    // variables, so its dynamic block starts at 0, and if that ever changes the fp32 scan answers instead
    if (lds_offset_of(T8) != 0u) {
        if (tid == 0) *p.overflow = 1;
        if (lane == 0) p.seg_len[wave_id] = 0;
        return;
    }
    const uint2* src_t = p.t8 + (int64_t)g * M * 256;
    for (int i = tid; i < M * 256; i += 256) T8[i] = src_t[i];
    __syncthreads();

    // rows are numbered in 32 bits (the launcher refuses 2^31 codes and more): a row's address is one 64-bit
    // shift-and-add, its validity one compare
    const uint32_t ntiles = (uint32_t)((p.n + 255) / 256);
    const uint32_t per = (ntiles + p.splits - 1) / p.splits;
    const uint32_t t0 = min((uint32_t)split * per, ntiles), t1 = min(t0 + per, ntiles);
    const uint32_t row_end = (uint32_t)min((int64_t)t1 * 256, p.n);      // this block's rows end here
    const uint32_t last = row_end - 1u;                      // reads past the block's rows repeat its last row
    constexpr int WN = M / 4;
    int2* seg = p.seg + (int64_t)wave_id * p.seg_cap;
    int kept = 0;                                            // wave-uniform
    bool full = false;
    uint32_t three = 3;
    asm volatile("" : "+v"(three));                          // stays in a vector register

    // the codes (and filter words) of the P8_UNROLL tiles whose first row for this lane is `row0`; rows past the end
    // of the block's share read its last row again (a cache hit, never used: `row < row_end` fails for them)
    auto fetch = [&](uint32_t row0, uint32_t (&wx)[P8_UNROLL][WN], uint32_t (&mwx)[P8_UNROLL]) {
#pragma unroll
        for (int u = 0; u < P8_UNROLL; ++u) {
            const uint32_t row = min(row0 + u * 256u, last);
            const uint8_t* src8 = p.codes + (uint64_t)row * M;
            if constexpr (M % 16 == 0) {
                const uint4* src = reinterpret_cast<const uint4*>(src8);
#pragma unroll
                for (int i = 0; i < M / 16; ++i) {
                    const uint4 v = src[i];
                    wx[u][4 * i] = v.x; wx[u][4 * i + 1] = v.y; wx[u][4 * i + 2] = v.z; wx[u][4 * i + 3] = v.w;
                }
            } else if constexpr (M % 8 == 0) {
                const uint2* src = reinterpret_cast<const uint2*>(src8);
#pragma unroll
                for (int i = 0; i < M / 8; ++i) { const uint2 v = src[i]; wx[u][2 * i] = v.x; wx[u][2 * i + 1] = v.y; }
            } else {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(src8);
#pragma unroll
                for (int i = 0; i < M / 4; ++i) wx[u][i] = src[i];
            }
            if constexpr (MASKED) mwx[u] = p.mask[row >> 5];
        }
    };
    // one step: the table sums of the tiles in `w`, then the rows that may be in a result
    auto step = [&](uint32_t row0, const uint32_t (&w)[P8_UNROLL][WN], const uint32_t (&mw)[P8_UNROLL]) {
        // Per 32-bit word of an entry (four queries, one byte each): `all` adds the whole words, `odd` adds the odd
        // bytes as two 16-bit fields. Both are plain 32-bit sums, two sub-spaces per three-operand add: the fields of
        // `odd` cannot carry into each other (M * 255 < 2^16), and all - 256 * odd (mod 2^32) is exactly the even
        // bytes' sums in the same two fields, whatever `all` carried on the way.
        uint32_t all_lo[P8_UNROLL], odd_lo[P8_UNROLL], all_hi[P8_UNROLL], odd_hi[P8_UNROLL];
#pragma unroll
        for (int u = 0; u < P8_UNROLL; ++u) { all_lo[u] = 0; odd_lo[u] = 0; all_hi[u] = 0; odd_hi[u] = 0; }
        auto odd_bytes = [](uint32_t v) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pq_u16x2, v) >> 8); };
        auto look_up = [&](auto jc) {
            constexpr int j = 2 * decltype(jc)::value;
#pragma unroll
            for (int u = 0; u < P8_UNROLL; ++u) {
                const uint32_t off0 = code_offset<j & 3>(w[u][j >> 2], three);
                const uint32_t off1 = code_offset<(j + 1) & 3>(w[u][(j + 1) >> 2], three);
                const uint2 a = lds_entry_at(off0 + j * 2048);                 // the tables start at LDS offset 0
                const uint2 b = lds_entry_at(off1 + (j + 1) * 2048);
                all_lo[u] = all_lo[u] + a.x + b.x;  odd_lo[u] = odd_lo[u] + odd_bytes(a.x) + odd_bytes(b.x);
                all_hi[u] = all_hi[u] + a.y + b.y;  odd_hi[u] = odd_hi[u] + odd_bytes(a.y) + odd_bytes(b.y);
            }
            // the scheduler may not move table reads across every fourth sub-space: sixteen reads in flight are
            // enough, and all of them in flight cost a wave per SIMD in registers
            if constexpr ((j & 3) == 2) __builtin_amdgcn_sched_barrier(0);
        };
        static_for<M / 2>(look_up);
#pragma unroll
        for (int u = 0; u < P8_UNROLL; ++u) {
            const uint32_t row = row0 + u * 256u;
            const pq_u16x2 odd_l = __builtin_bit_cast(pq_u16x2, odd_lo[u]), odd_h = __builtin_bit_cast(pq_u16x2, odd_hi[u]);
            const pq_u16x2 even_lo = __builtin_bit_cast(pq_u16x2, all_lo[u] - (odd_lo[u] << 8));
            const pq_u16x2 even_hi = __builtin_bit_cast(pq_u16x2, all_hi[u] - (odd_hi[u] << 8));
            const pq_u16x2 mn2 = __builtin_elementwise_min(__builtin_elementwise_min(even_lo, odd_l),
                                                           __builtin_elementwise_min(even_hi, odd_h));
            const unsigned mn = min((unsigned)mn2.x, (unsigned)mn2.y);
            bool pass = row < row_end && mn <= (unsigned)P8_LIMIT;
            if constexpr (MASKED) pass = pass && ((mw[u] >> (row & 31u)) & 1u);
            const unsigned long long votes = __ballot(pass);
            if (votes != 0ull) {                                                   // rare
                const int cnt = __popcll(votes);
                if (kept + cnt > p.seg_cap) {
                    full = true;
                } else if (pass) {
                    // queries 0, 2 (even bytes of the low word), 1, 3 (odd), then the high word's 4, 6, 5, 7
                    unsigned bits = 0;
                    bits |= (even_lo.x <= P8_LIMIT) ? 1u : 0u;   bits |= (odd_l.x <= P8_LIMIT) ? 2u : 0u;
                    bits |= (even_lo.y <= P8_LIMIT) ? 4u : 0u;   bits |= (odd_l.y <= P8_LIMIT) ? 8u : 0u;
                    bits |= (even_hi.x <= P8_LIMIT) ? 16u : 0u;  bits |= (odd_h.x <= P8_LIMIT) ? 32u : 0u;
                    bits |= (even_hi.y <= P8_LIMIT) ? 64u : 0u;  bits |= (odd_h.y <= P8_LIMIT) ? 128u : 0u;
                    seg[kept + __popcll(votes & ((1ull << lane) - 1ull))] = make_int2((int)row, (int)bits);
                }
                if (!full) kept += cnt;
            }
        }
    };

    // P8_DEPTH sets of code registers take turns: the codes of a step are asked for P8_DEPTH - 1 steps before their
    // sums are taken (a step lasts about half a microsecond, a read from HBM under load longer), with no register
    // copies between steps
    constexpr uint32_t STEP = P8_UNROLL * 256u;
    uint32_t w[P8_DEPTH][P8_UNROLL][WN], mk[P8_DEPTH][P8_UNROLL];
#pragma unroll
    for (int s = 0; s < P8_DEPTH; ++s)
#pragma unroll
        for (int u = 0; u < P8_UNROLL; ++u) mk[s][u] = 0xffffffffu;
    uint32_t row0 = t0 * 256u + (uint32_t)tid;
    uint32_t tb = t0;
    if (t0 >= t1) goto done;
#pragma unroll
    for (int s = 0; s < P8_DEPTH - 1; ++s) fetch(row0 + s * STEP, w[s], mk[s]);
    for (;;) {
#pragma unroll
        for (int s = 0; s < P8_DEPTH; ++s) {
            fetch(row0 + (P8_DEPTH - 1) * STEP, w[(s + P8_DEPTH - 1) % P8_DEPTH], mk[(s + P8_DEPTH - 1) % P8_DEPTH]);
            step(row0, w[s], mk[s]);
            row0 += STEP;
            tb += P8_UNROLL;
            if (tb >= t1) goto done;                         // wave-uniform
        }
    }
done:
    if (lane == 0) {
        p.seg_len[wave_id] = kept;
        if (full) *p.overflow = 1;
    }
}

// ---------------------------------------------------------------- the survivors in fp32
struct PqExact8Params {
    const uint8_t* codes; int m;
    const float* lut; int nq;                   // [nq][m][256]
    const float* thr;                           // [nq]
    const int2* seg; const int* seg_len; int seg_cap; int n_seg; int segs_per_group;
    float* area_d; int* area_i; int area;       // [nq][area]
    int* queue_len; int len_stride;
    int* overflow;
};

// A block per 32 consecutive segments of a query group, eight per wave, the kept rows of a wave's segments spread
// over its lanes.  Nearly every kept row is kept for one query, so the lanes take their queries one set bit at a
// time, each lane its own query's table (as many rounds as the busiest lane has bits, not eight).  Rows within their
// query's bound are collected in LDS with their rank among the block's rows of that query; then one atomic per
// (block, query) reserves the block's room in the query's candidate area.  (An atomic per row, or per wave and
// query, makes a few thousand returning atomics on each of eight addresses: 50 microseconds.)
static constexpr int PX_SEGS = 8;          // segments per wave
static constexpr int PX_CAP = 1024;        // rows within their bounds that a block can hold

template <int M>
__global__ __launch_bounds__(256) void pq_exact8_kernel(PqExact8Params p) {
    __shared__ float sh_d[PX_CAP];
    __shared__ int sh_row[PX_CAP];
    __shared__ int sh_meta[PX_CAP];          // query of the group << 16 | rank among the block's rows of that query
    __shared__ int sh_cnt[P8_Q], sh_base[P8_Q], sh_total, sh_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y, q0 = g * P8_Q;
    if (tid < P8_Q) sh_cnt[tid] = 0;
    if (tid == 0) { sh_total = 0; sh_skip = *p.overflow; }
    __syncthreads();
    if (sh_skip != 0) return;                                // the fp32 scan answers this batch
    const int first = (blockIdx.x * 4 + wave) * PX_SEGS;     // of this wave's segments, within the group
    const int s0 = g * p.segs_per_group + first;
    // lane l < PX_SEGS: the rows kept in segment first + l, then the running totals over the lanes
    int incl = (lane < PX_SEGS && first + lane < p.segs_per_group) ? p.seg_len[s0 + lane] : 0;
#pragma unroll
    for (int off = 1; off < PX_SEGS; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    int ends[PX_SEGS];
#pragma unroll
    for (int i = 0; i < PX_SEGS; ++i) ends[i] = __builtin_amdgcn_readlane(incl, i);
    const int total = ends[PX_SEGS - 1];
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int i = i0 + lane;
        const bool have = i < total;
        int which = 0, begin = 0;
#pragma unroll
        for (int j = 0; j < PX_SEGS - 1; ++j)
            if (i >= ends[j]) { which = j + 1; begin = ends[j]; }
        const int2 e = have ? p.seg[(int64_t)(s0 + which) * p.seg_cap + (i - begin)] : make_int2(0, 0);
        uint32_t w[M / 4];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(p.codes + (int64_t)e.x * M);
#pragma unroll
        for (int j = 0; j < M / 4; ++j) w[j] = src[j];
        unsigned bits = have ? ((unsigned)e.y & 0xffu) : 0u;   // a set bit implies q0 + qi < nq (absent queries have bytes of 255)
        while (__ballot(bits != 0u) != 0ull) {
            const bool mine = bits != 0u;
            const int qi = mine ? __builtin_ctz(bits) : 0;                   // idle lanes read query q0's table (unused)
            bits &= bits - 1u;
            const float* t = p.lut + (int64_t)(q0 + qi) * M * 256;
            float acc = t[w[0] & 0xffu];
#pragma unroll
            for (int j = 1; j < M; ++j) acc = acc + t[j * 256 + ((w[j >> 2] >> (8 * (j & 3))) & 0xffu)];
            if (mine && acc <= p.thr[q0 + qi]) {
                const int slot = atomicAdd(&sh_total, 1);
                if (slot < PX_CAP) {
                    sh_d[slot] = sqrtf(acc);
                    sh_row[slot] = e.x;
                    sh_meta[slot] = (qi << 16) | atomicAdd(&sh_cnt[qi], 1);
                }
            }
        }
    }
    __syncthreads();
    const int held = sh_total;
    if (held == 0) return;
    if (held > PX_CAP) {                                     // (the ranks of the rows that did fit are of no use then)
        if (tid == 0) *p.overflow = 1;
        return;
    }
    if (tid < P8_Q && sh_cnt[tid] > 0) sh_base[tid] = atomicAdd(p.queue_len + (int64_t)(q0 + tid) * p.len_stride, sh_cnt[tid]);
    __syncthreads();
    for (int t = tid; t < held; t += 256) {
        const int meta = sh_meta[t], qi = meta >> 16;
        const int at = sh_base[qi] + (meta & 0xffff);
        if (at >= p.area) {
            *p.overflow = 1;
        } else {
            const int64_t o = (int64_t)(q0 + qi) * p.area + at;
            p.area_d[o] = sh_d[t];
            p.area_i[o] = sh_row[t];
        }
    }
}

// ---------------------------------------------------------------- launcher
bool pq_scan8_supports(int m, int ksub) { return ksub == 256 && (m == 4 || m == 8 || m == 16 || m == 32); }
int pq_scan8_groups(int nq) { return (nq + P8_Q - 1) / P8_Q; }
int pq_scan8_splits(int64_t n, int nq) {
    const int64_t ntiles = (n + 255) / 256;
    const int groups = pq_scan8_groups(nq);
    return (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + 15) / 16, (8 * (int64_t)ctx().num_cus + groups - 1) / groups));
}
size_t pq_scan8_table_bytes(int nq, int m) { return (size_t)pq_scan8_groups(nq) * m * 256 * sizeof(uint2); }

int launch_pq_scan8(const PqScan8Job& j, hipStream_t s) {
    VDB_REQUIRE(pq_scan8_supports(j.m, 256), "byte-table scan: m in {4, 8, 16, 32}");
    VDB_REQUIRE(j.n > 0 && j.n < (int64_t)0x7fffffff && j.nq > 0, "byte-table scan: 1 .. 2^31-1 codes, at least one query");
    const int groups = pq_scan8_groups(j.nq);
    const int n_seg = groups * j.splits * 4;
    PqQuantParams qp{j.lut, j.nq, j.m, j.lut_mins, j.bound_d, j.bound_stride, reinterpret_cast<uint2*>(j.t8), j.thr, j.bound_keep,
                     j.overflow};
    hipLaunchKernelGGL(pq_quantise_kernel, dim3(groups, j.m), dim3(256), 0, s, qp);
    PqScan8Params sp{j.codes, j.n, j.splits, reinterpret_cast<const uint2*>(j.t8), j.mask,
                     reinterpret_cast<int2*>(j.seg), j.seg_len, j.seg_cap, j.overflow};
    const size_t lds = (size_t)j.m * 256 * sizeof(uint2);
    {
        KernelTimer timer(s, j.timed);
#define VDB_LAUNCH_P8(MM)                                                                      \
    case MM: {                                                                                 \
        auto kern = j.mask ? pq_scan8_kernel<MM, true> : pq_scan8_kernel<MM, false>;           \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));                  \
        hipLaunchKernelGGL(kern, dim3(j.splits, groups), dim3(256), lds, s, sp);               \
    } break;
        switch (j.m) {
            VDB_LAUNCH_P8(4)
            VDB_LAUNCH_P8(8)
            VDB_LAUNCH_P8(16)
            VDB_LAUNCH_P8(32)
        }
#undef VDB_LAUNCH_P8
    }
    PqExact8Params ep{j.codes, j.m, j.lut, j.nq, j.thr, reinterpret_cast<const int2*>(j.seg), j.seg_len, j.seg_cap, n_seg,
                      j.splits * 4, j.area_d, j.area_i, j.area, j.queue_len, j.len_stride, j.overflow};
    const dim3 xgrid((j.splits * 4 + 4 * PX_SEGS - 1) / (4 * PX_SEGS), groups);
    switch (j.m) {
        case 4: hipLaunchKernelGGL(pq_exact8_kernel<4>, xgrid, dim3(256), 0, s, ep); break;
        case 8: hipLaunchKernelGGL(pq_exact8_kernel<8>, xgrid, dim3(256), 0, s, ep); break;
        case 16: hipLaunchKernelGGL(pq_exact8_kernel<16>, xgrid, dim3(256), 0, s, ep); break;
        case 32: hipLaunchKernelGGL(pq_exact8_kernel<32>, xgrid, dim3(256), 0, s, ep); break;
    }
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
