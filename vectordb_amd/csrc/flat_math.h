// Tile geometry and per-metric arithmetic shared by the matrix-core kernels of
// the Flat index (flat_gemm.hip: per-thread candidate lists; flat_queue.hip:
// candidate queues).
#pragma once

#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int TM = 128;      // database rows per tile
static constexpr int TQ = 128;      // queries per tile
static constexpr int TKB = 128;     // bytes of every row per K step: 32 fp32 or 64 bf16
static constexpr int LW = 36;       // padded LDS row width in dwords (128 B of data + 16 B)
static constexpr int NT = LIST_NT;  // threads per block


// ---- final distance value from the dot product, per metric ---------------
template <int METRIC>
__device__ __forceinline__ float finish(float dot, float row_term, float q_term) {
    if (METRIC == VDB_DOT) {
        return -dot;
    } else if (METRIC == VDB_COSINE) {
        // q_term = |q| (or +inf for a zero query -> similarity 0 -> distance 1);
        // row_term = |c| with the <1e-8 -> 1 substitution already applied
        float sim = dot / (q_term * row_term);
        sim = fminf(fmaxf(sim, -1.0f), 1.0f);
        return 1.0f - sim;
    } else {
        // (|q|^2 + |c|^2) - 2*dot : 2*dot is exact, so one fma rounds like the
        // reference's separate multiply and subtract
        float sq = fmaxf(__fmaf_rn(-2.0f, dot, q_term + row_term), 0.0f);
        return METRIC == VDB_EUCLIDEAN ? sqrtf(sq) : sq;
    }
}

// ---- conservative filter: never rejects a value whose exact distance is <= thr
template <int METRIC>
__device__ __forceinline__ float filter_bound(float thr, float q_term) {
    if (METRIC == VDB_DOT) return -thr;                                     // pass: dot >= bound
    if (METRIC == VDB_EUCLIDEAN_SQ) return thr;                             // pass: raw <= bound
    if (METRIC == VDB_EUCLIDEAN) return thr * thr * 1.000001f;              // sqrt_rn(x) <= thr  =>  x <= thr^2 (1 + 2^-22)
    // cosine, pass: dot >= bound * |c|.  1 - clip(dot/(|q||c|)) <= thr implies
    // dot/(|q||c|) >= 1 - thr - 5e-7 after every rounding on both sides.
    if (q_term == INFINITY) return (1.0f <= thr) ? -INFINITY : INFINITY;    // zero query: every distance is exactly 1
    if (thr >= 2.0f) return -INFINITY;
    return ((1.0f - thr) - 3e-6f) * q_term;
}

template <int METRIC>
__device__ __forceinline__ bool filter_pass(float dot, float row_term, float q_term, float bound) {
    if (METRIC == VDB_DOT) return dot >= bound;
    if (METRIC == VDB_COSINE) return dot >= bound * row_term;
    return __fmaf_rn(-2.0f, dot, q_term + row_term) <= bound;
}

// element e (lane-varying) of a 16-register accumulator: binary tree of selects
__device__ __forceinline__ float pick16(const f32x16& a, int e) {
    const bool b0 = e & 1, b1 = e & 2, b2 = e & 4, b3 = e & 8;
    const float p0 = b0 ? a[1] : a[0], p1 = b0 ? a[3] : a[2], p2 = b0 ? a[5] : a[4], p3 = b0 ? a[7] : a[6];
    const float p4 = b0 ? a[9] : a[8], p5 = b0 ? a[11] : a[10], p6 = b0 ? a[13] : a[12], p7 = b0 ? a[15] : a[14];
    const float q0 = b1 ? p1 : p0, q1 = b1 ? p3 : p2, q2 = b1 ? p5 : p4, q3 = b1 ? p7 : p6;
    const float r0 = b2 ? q1 : q0, r1 = b2 ? q3 : q2;
    return b3 ? r1 : r0;
}


// slabs of row tiles x query tiles: about two blocks per CU in total
static inline void plan_grid(int n_tiles, int nq, int num_cus, int* n_qtiles, int* n_slabs, int* tiles_per_slab) {
    *n_qtiles = (nq + TQ - 1) / TQ;
    int want = (2 * num_cus + *n_qtiles - 1) / *n_qtiles;
    if (want < 1) want = 1;
    if (want > n_tiles) want = n_tiles;
    *tiles_per_slab = (n_tiles + want - 1) / want;
    *n_slabs = (n_tiles + *tiles_per_slab - 1) / *tiles_per_slab;
}

}  // namespace vdb
