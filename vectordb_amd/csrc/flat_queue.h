// Shared by the candidate-queue kernels of the Flat index (flat_queue.hip: 128 x 128 tiles,
// two blocks per CU; flat_queue_wide.hip: 256 x 256 tiles for bf16 rows, one block per CU).
#pragma once

#include "flat_math.h"

namespace vdb {

struct QueueParams {
    const void* rows; int64_t n; int ld_bytes;     // row pitch in bytes (multiple of 16); fp32 or bf16 elements
    const float* row_sq; const float* row_nrm;
    const void* q; int nq;                         // same element type and pitch as the rows
    const float* q_sq; const float* q_nrm;
    const uint32_t* mask;
    const float* thr;                              // per query: an upper bound of its k-th best distance
    // Optional, per query: how far a dot product of this launch may lie from the one the bound was derived for
    // (bf16 copies of fp32 rows and queries as a filter in front of an exact fp32 selection).  The filter is
    // loosened by that much, so no row that the exact dot product would pass is dropped.
    const float* slack;
    int n_tiles, n_slabs, tiles_per_slab, n_qtiles;
    int cap;                                       // queue slots per (query, slab)
    float2* queue;                                 // [nq][n_slabs][cap]  (dot product, row number bits)
    int* counts;                                   // [nq][n_slabs]       survivors seen (may exceed cap)
    // bound pass (flat_queue_fast_kernel<..., MINPASS = true>): visited tile t is database tile
    // t * tile_stride, and every lane's smallest distance goes to min_out[(query * n_slabs + slab) * 2 + half]
    int tile_stride;
    float* min_out;
    // nearest-row pass (flat_queue_fast_kernel<..., MODE = QMODE_ARGMIN>): every lane's smallest and second smallest
    // key and the row of the smallest go to arg_out[(query * n_slabs + slab) * 2 + half] = (key1, key2, row bits, -)
    float4* arg_out;
};

enum { QMODE_QUEUE = 0, QMODE_MINPASS = 1, QMODE_ARGMIN = 2 };

// filter_bound<>() for dot products that are only known to within `slack`
template <int METRIC>
__device__ __forceinline__ float loosen_bound(float bound, float slack) {
    if (!(fabsf(bound) < INFINITY)) return bound;                           // passes all or nothing already
    if (METRIC == VDB_DOT || METRIC == VDB_COSINE) return bound - slack;    // pass: dot >= bound (* |c|)
    return (bound + 2.0f * slack) * 1.000001f;                              // pass: (|q|^2+|c|^2) - 2 dot <= bound
}

// Smallest "key" of the 16 values of one accumulator, for the bound pass.  Keys order like the
// distances of one query: -dot (dot), (|q|^2+|c|^2) - 2 dot (both Euclidean forms, the exact
// pre-clamp term of finish<>()), -(dot / |c|) (cosine; `rs` then holds 1/|c|).
template <int METRIC>
__device__ __forceinline__ float min_key16(const f32x16& a, const float* rs, int s, int half, float q_term, uint32_t avail) {
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 t = make_float4(0, 0, 0, 0);
        if (METRIC != VDB_DOT) t = *reinterpret_cast<const float4*>(rs + s * 32 + 8 * g + 4 * half);
        const float rts[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = a[g * 4 + j];
            float key = (METRIC == VDB_DOT) ? -d : (METRIC == VDB_COSINE) ? -(d * rts[j]) : __fmaf_rn(-2.0f, d, q_term + rts[j]);
            if (!((avail >> (g * 4 + j)) & 1u)) key = INFINITY;
            v[g * 4 + j] = key;
        }
    }
    float m = v[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) m = fminf(m, v[e]);
    return m;
}

// The lane's running smallest key b1 (at row bi) and second smallest b2 over the 16 values of one accumulator,
// Euclidean forms only: key = the pre-clamp term of finish<>().  Rows reach a lane in ascending order and the
// comparison is strict, so of equal keys the first row stays.  v_med3 keeps (b1, b2) sorted: the new b2 is the
// median of (b1, key, b2).
template <int METRIC, bool MASKED>
__device__ __forceinline__ void argmin16(const f32x16& a, const float* rs, int s, int half, float q_term, uint32_t avail,
                                         int row0, float& b1, float& b2, int& bi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 qq = {q_term, q_term}, m2 = {-2.0f, -2.0f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(rs + s * 32 + 8 * g + 4 * half);
        // two keys per packed add and packed fma (the same roundings as the scalar forms)
        const f2 s01 = qq + f2{t.x, t.y}, s23 = qq + f2{t.z, t.w};
        const f2 k01 = __builtin_elementwise_fma(m2, f2{a[g * 4], a[g * 4 + 1]}, s01);
        const f2 k23 = __builtin_elementwise_fma(m2, f2{a[g * 4 + 2], a[g * 4 + 3]}, s23);
        const float keys[4] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float key = keys[j];
            if (MASKED && !((avail >> (g * 4 + j)) & 1u)) key = INFINITY;
            const bool less = key < b1;
            b2 = __builtin_amdgcn_fmed3f(b1, key, b2);
            bi = less ? row0 + 8 * g + j : bi;
            b1 = fminf(b1, key);
        }
    }
}

// per-lane availability bits (value e = 4g + j <-> row 8g + 4*half + j of the 32-row sub-tile at r0)
__device__ __forceinline__ uint32_t avail16(const uint32_t* mask, int64_t r0, int64_t n, int half) {
    uint32_t bits = 0xffffffffu;
    if (mask) bits = (r0 < n) ? mask[r0 >> 5] : 0u;
    const int64_t left = n - r0;
    if (left < 32) bits = left <= 0 ? 0u : (bits & ((1u << left) - 1u));
    bits >>= (4 * half);
    return (bits & 0xFu) | ((bits >> 4) & 0xF0u) | ((bits >> 8) & 0xF00u) | ((bits >> 12) & 0xF000u);
}

// Survivors of one finished 128 x 128 tile -> queue.  `acc[s]` holds the dot products of this
// lane's query with rows rbase + 32 s + 8 g + 4 half + j (value index 4 g + j); `rs` the row terms
// of the tile.  A packed test per accumulator (one fma per value, a max/min tree, one compare)
// decides whether any of its 16 values can pass; only then are the values looked at one by one.
template <int METRIC>
__device__ __forceinline__ void queue_epilogue(const f32x16 (&acc)[4], const float* rs, int64_t rbase, int64_t n,
                                               const uint32_t* mask, int half, float bound, float q_term, int* cnt,
                                               float2* my_queue, int cap) {
    const bool masked = (rbase + TM > n) || mask != nullptr;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[16];
        if (METRIC == VDB_DOT) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc[s][e];
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(rs + s * 32 + 8 * g + 4 * half);
                const float rts[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // cosine: dot - bound*|c| >= 0 ; L2: (|q|^2+|c|^2) - 2 dot <= bound   (one rounding each)
                    v[g * 4 + j] = (METRIC == VDB_COSINE) ? __fmaf_rn(-bound, rts[j], acc[s][g * 4 + j])
                                                          : __fmaf_rn(-2.0f, acc[s][g * 4 + j], q_term + rts[j]);
                }
            }
        }
        const bool want_max = (METRIC == VDB_DOT || METRIC == VDB_COSINE);
        float m = v[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) m = want_max ? fmaxf(m, v[e]) : fminf(m, v[e]);
        const float lim = (METRIC == VDB_COSINE) ? 0.0f : bound;
        const bool any = want_max ? (m >= lim) : (m <= lim);
        if (any) {
            uint32_t hits = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool pass = want_max ? (v[e] >= lim) : (v[e] <= lim);
                hits |= pass ? (1u << e) : 0u;
            }
            if (masked) hits &= avail16(mask, rbase + s * 32, n, half);
            while (hits) {
                const int e = __ffs(hits) - 1;
                hits &= hits - 1;
                const int off = 8 * (e >> 2) + (e & 3) + 4 * half;
                const float dot = pick16(acc[s], e);
                const int row = (int)(rbase + s * 32 + off);
                const int slot = atomicAdd(cnt, 1);
                if (slot < cap) my_queue[slot] = make_float2(dot, __int_as_float(row));
            }
        }
    }
}


// ---- 256 x 256 tiles (flat_queue_wide.hip)
static constexpr int WM = 256;      // database rows per tile
static constexpr int WQ = 256;      // queries per tile
static constexpr int WNT = 512;     // threads per block (8 waves: 2 row blocks x 4 query blocks)
// bf16 rows with whole 128-byte K steps and whole readable tiles take the wide kernel
bool queue_use_wide(const Rows& r);
// slabs of 256-row tiles x 256-query tiles: one block per CU
void plan_grid_wide(int n_tiles, int nq, int num_cus, int* n_qtiles, int* n_slabs, int* tiles_per_slab);
int launch_queue_wide(int metric, QueueParams& p, hipStream_t s);

}  // namespace vdb
