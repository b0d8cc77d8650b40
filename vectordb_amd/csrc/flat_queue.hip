// Batched distance + top-k for the Flat index on the matrix cores, candidate-queue form.
//
// Same contraction as flat_gemm.hip (128 queries x 128 database rows per tile,
// v_mfma_f32_32x32x2_f32 or v_mfma_f32_32x32x16_bf16, distances as in
// distance/metrics.py:453-484), for searches that already hold a per-query upper
// bound of the k-th best distance (the k-th best over a strided sample of the
// rows).  With that bound almost no value survives the conservative filter, so
// the kernel keeps no sorted lists: a survivor's raw dot product and row number
// are appended to a small queue per (query, slab of rows) and the exact distance,
// the ordering and the final k best (index/flat.py:354-362) are left to
// queue_select_kernel.  The LDS that the lists would take holds a second staging
// buffer instead: one barrier per K step, the next step's rows are written to LDS
// and the step after that is fetched while the current one is multiplied.
//
// A queue that overflows is reported through a flag; the caller then repeats the
// search with the list kernel, so the result never depends on the queue size.
#include "flat_queue.h"

#include <type_traits>

namespace vdb {

template <int METRIC, bool BF16>
__global__ __launch_bounds__(NT, 2) void flat_queue_kernel(QueueParams p) {
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                                   // [2][TM][LW]  database rows, two stages
    float* Bs = As + 2 * TM * LW;                       // [2][TQ][LW]  queries, two stages
    float* Rs = Bs + 2 * TQ * LW;                       // [2][TM]      per-row term (|c|^2 or |c|), by tile parity
    int* Cnt = reinterpret_cast<int*>(Rs + 2 * TM);     // [TQ]         survivors per query of this block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;

    // blocks that work on the same slab of rows (one per query tile) get the
    // same blockIdx % 8, i.e. the same XCD and L2, and adjacent dispatch slots
    int slab, qt;
    if (p.n_slabs >= 8) {
        const int per_group = 8 * p.n_qtiles;
        const int grp = blockIdx.x / per_group;
        const int rem = blockIdx.x - grp * per_group;
        slab = grp * 8 + (rem & 7);
        qt = rem >> 3;
        if (slab >= p.n_slabs) return;
    } else {
        qt = blockIdx.x / p.n_slabs;
        slab = blockIdx.x - qt * p.n_slabs;
    }
    const int tile0 = slab * p.tiles_per_slab;
    const int tile1 = min(tile0 + p.tiles_per_slab, p.n_tiles);

    const int my_q = qt * TQ + wave * 32 + col;
    const bool q_ok = my_q < p.nq;
    const int my_q_ld = min(my_q, p.nq - 1);
    float q_term = 0.0f;
    if (METRIC == VDB_COSINE) {
        q_term = p.q_nrm[my_q_ld];
        if (q_term < 1e-8f) q_term = INFINITY;      // metrics.py:472-473: all ones
    } else if (METRIC != VDB_DOT) {
        q_term = p.q_sq[my_q_ld];
    }
    // lanes past the last query must never pass the filter
    float bound = filter_bound<METRIC>(p.thr[my_q_ld], q_term);
    if (p.slack) bound = loosen_bound<METRIC>(bound, p.slack[my_q_ld]);
    if (!q_ok) bound = (METRIC == VDB_DOT || METRIC == VDB_COSINE) ? INFINITY : -INFINITY;
    float2* my_queue = p.queue + ((int64_t)my_q_ld * p.n_slabs + slab) * p.cap;
    if (tid < TQ) Cnt[tid] = 0;

    const int ksteps = (p.ld_bytes + TKB - 1) / TKB;
    const int srow = tid >> 3;          // staging: row within a 32-row pass
    const int skc = (tid & 7) * 4;      // staging: dword offset inside the K step (16-byte pieces)
    const int64_t q_base = (int64_t)qt * TQ;
    const bool k_tail = (p.ld_bytes % TKB) != 0;
    const int ldw = p.ld_bytes >> 2;    // row pitch in dwords
    uint32_t offb[4], offa[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t gq = min(q_base + ps * 32 + srow, (int64_t)p.nq - 1) - q_base;
        offb[ps] = (uint32_t)(gq * ldw + skc);
        offa[ps] = (uint32_t)((ps * 32 + srow) * ldw + skc);
    }
    const float* rows_w = reinterpret_cast<const float*>(p.rows);
    const float* qbase_ptr = reinterpret_cast<const float*>(p.q) + q_base * ldw;
    float4 ra[4], rb[4];
    float rterm = 0.0f;

    // fetch K step `ks` of tile `tile` (rows and queries) into registers
    auto issue_loads = [&](int tile, int ks) {
        const int64_t rbase = (int64_t)tile * TM;
        const float* abase = rows_w + rbase * ldw + ks * (TKB / 4);
        const float* bbase = qbase_ptr + ks * (TKB / 4);
        if (rbase + TM > p.n) {             // last tile: rows past the end re-read the last row (masked out later)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                offa[ps] = (uint32_t)(min((int64_t)(ps * 32 + srow), p.n - 1 - rbase) * ldw + skc);
        }
        if (k_tail && ks == ksteps - 1) {
            const bool kin = ks * (TKB / 4) + skc < ldw;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                ra[ps] = kin ? *reinterpret_cast<const float4*>(abase + offa[ps]) : make_float4(0, 0, 0, 0);
                rb[ps] = kin ? *reinterpret_cast<const float4*>(bbase + offb[ps]) : make_float4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                ra[ps] = *reinterpret_cast<const float4*>(abase + offa[ps]);
                rb[ps] = *reinterpret_cast<const float4*>(bbase + offb[ps]);
            }
        }
        if (METRIC != VDB_DOT && ks == 0 && tid < TM) {
            const int64_t gr = min(rbase + tid, p.n - 1);
            rterm = (METRIC == VDB_COSINE) ? p.row_nrm[gr] : p.row_sq[gr];
        }
    };
    // registers -> LDS stage `buf`
    auto stage_write = [&](int buf, int tile, int ks) {
        float* a = As + buf * (TM * LW);
        float* b = Bs + buf * (TQ * LW);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = ps * 32 + srow;
            *reinterpret_cast<float4*>(a + r * LW + skc) = ra[ps];
            *reinterpret_cast<float4*>(b + r * LW + skc) = rb[ps];
        }
        if (METRIC != VDB_DOT && ks == 0 && tid < TM) {
            float t = rterm;
            if (METRIC == VDB_COSINE && t < 1e-8f) t = 1.0f;        // metrics.py:476
            Rs[(tile & 1) * TM + tid] = t;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.0f;

    auto epilogue = [&](int tile) {
        queue_epilogue<METRIC>(acc, Rs + (tile & 1) * TM, (int64_t)tile * TM, p.n, p.mask, half, bound, q_term,
                               &Cnt[wave * 32 + col], my_queue, p.cap);
    };

    const int n_steps = (tile1 - tile0) * ksteps;
    // cursors: c = step being multiplied, w = the step after it (written to LDS meanwhile),
    // l = the step after that (fetched meanwhile)
    int ct = tile0, ck = 0;
    int wt = ct, wk = ck + 1;
    if (wk == ksteps) { wk = 0; ++wt; }
    int lt = wt, lk = wk + 1;
    if (lk == ksteps) { lk = 0; ++lt; }

    if (n_steps > 0) {
        issue_loads(ct, ck);
        stage_write(0, ct, ck);
        if (n_steps > 1) issue_loads(wt, wk);
    }
    __syncthreads();

    for (int st = 0; st < n_steps; ++st) {
        const int b = st & 1;
        const float* bq = Bs + b * (TQ * LW) + (wave * 32 + col) * LW + half * 4;
        const float* ar = As + b * (TM * LW) + col * LW + half * 4;
        float4 bf0[2], af0[2][4], bf1[2], af1[2][4];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            bf0[c2] = *reinterpret_cast<const float4*>(bq + c2 * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) af0[c2][s] = *reinterpret_cast<const float4*>(ar + s * 32 * LW + c2 * 8);
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            bf1[c2] = *reinterpret_cast<const float4*>(bq + (2 + c2) * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) af1[c2][s] = *reinterpret_cast<const float4*>(ar + s * 32 * LW + (2 + c2) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (BF16) {
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af0[c2][s]),
                                                                     __builtin_bit_cast(bf16x8, bf0[c2]), acc[s], 0, 0, 0);
                } else {
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[c2][s].x, bf0[c2].x, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[c2][s].y, bf0[c2].y, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[c2][s].z, bf0[c2].z, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[c2][s].w, bf0[c2].w, acc[s], 0, 0, 0);
                }
            }
        }
        // the next step's operands have had a whole step to arrive: park them in the other
        // stage and start fetching the step after, all behind the matrix instructions above
        if (st + 1 < n_steps) {
            stage_write(1 - b, wt, wk);
            if (st + 2 < n_steps) issue_loads(lt, lk);
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (BF16) {
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af1[c2][s]),
                                                                     __builtin_bit_cast(bf16x8, bf1[c2]), acc[s], 0, 0, 0);
                } else {
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[c2][s].x, bf1[c2].x, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[c2][s].y, bf1[c2].y, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[c2][s].z, bf1[c2].z, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[c2][s].w, bf1[c2].w, acc[s], 0, 0, 0);
                }
            }
        }
        if (ck == ksteps - 1) {
            epilogue(ct);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] = 0.0f;
        }
        __syncthreads();
        ct = wt; ck = wk;
        wt = lt; wk = lk;
        if (++lk == ksteps) { lk = 0; ++lt; }
    }

    if (tid < TQ && q_base + tid < p.nq) p.counts[(q_base + tid) * p.n_slabs + slab] = Cnt[tid];
}

// --------------------------------------------------------------------------
// The same kernel for the common shape: row pitch a multiple of 128 bytes and a row store
// whose capacity covers whole tiles (rows past n are read and masked out, never clamped).
// A step then has no branch and no vector ALU work: its fragment reads, the 8 LDS writes of
// the next step's operands and the 8 global loads of the step after are spread, in program
// order, between this wave's own matrix instructions.  (Vector ALU and LDS/VMEM issue of a
// wave crawls while the other block's wave on the same SIMD runs dependent matrix
// instructions back to back, so nothing is left for a separate phase to do.)
#define VDB_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int METRIC, bool BF16, int MODE>
__global__ __launch_bounds__(NT, 2) void flat_queue_fast_kernel(QueueParams p) {
    constexpr bool MINPASS = MODE == QMODE_MINPASS;
    constexpr bool ARGMIN = MODE == QMODE_ARGMIN;
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                                   // [2][TM][LW]
    float* Bs = As + 2 * TM * LW;                       // [2][TQ][LW]
    float* Rs = Bs + 2 * TQ * LW;                       // [2][TM]
    int* Cnt = reinterpret_cast<int*>(Rs + 2 * TM);     // [TQ]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;

    int slab, qt;
    if (p.n_slabs >= 8) {
        const int per_group = 8 * p.n_qtiles;
        const int grp = blockIdx.x / per_group;
        const int rem = blockIdx.x - grp * per_group;
        slab = grp * 8 + (rem & 7);
        qt = rem >> 3;
        if (slab >= p.n_slabs) return;
    } else {
        qt = blockIdx.x / p.n_slabs;
        slab = blockIdx.x - qt * p.n_slabs;
    }
    const int tile0 = slab * p.tiles_per_slab;
    const int tile1 = min(tile0 + p.tiles_per_slab, p.n_tiles);

    const int my_q = qt * TQ + wave * 32 + col;
    const bool q_ok = my_q < p.nq;
    const int my_q_ld = min(my_q, p.nq - 1);
    float q_term = 0.0f;
    if (METRIC == VDB_COSINE) {
        q_term = p.q_nrm[my_q_ld];
        if (q_term < 1e-8f) q_term = INFINITY;
    } else if (METRIC != VDB_DOT) {
        q_term = p.q_sq[my_q_ld];
    }
    float bound = 0.0f;
    float2* my_queue = nullptr;
    float best = INFINITY;                              // bound pass: this lane's smallest key
    float best2 = INFINITY;                             // nearest-row pass: its second smallest, and the row of the smallest
    int best_row = -1;
    if (MODE == QMODE_QUEUE) {
        bound = filter_bound<METRIC>(p.thr[my_q_ld], q_term);
        if (p.slack) bound = loosen_bound<METRIC>(bound, p.slack[my_q_ld]);
        if (!q_ok) bound = (METRIC == VDB_DOT || METRIC == VDB_COSINE) ? INFINITY : -INFINITY;
        my_queue = p.queue + ((int64_t)my_q_ld * p.n_slabs + slab) * p.cap;
    }
    const int tstride = MINPASS ? p.tile_stride : 1;    // visited tile t is database tile t * tstride
    int* my_cnt = &Cnt[wave * 32 + col];
    if (tid < TQ) Cnt[tid] = 0;

    const int ksteps = p.ld_bytes / TKB;
    const int srow = tid >> 3;
    const int skc = (tid & 7) * 4;
    const int64_t q_base = (int64_t)qt * TQ;
    const int ldw = p.ld_bytes >> 2;
    uint32_t offb[4], offa[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t gq = min(q_base + ps * 32 + srow, (int64_t)p.nq - 1) - q_base;
        offb[ps] = (uint32_t)(gq * ldw + skc) * 4u;                 // byte offsets from a wave-uniform base
        offa[ps] = (uint32_t)((ps * 32 + srow) * ldw + skc) * 4u;
    }
    const char* rows_w = reinterpret_cast<const char*>(p.rows);
    const char* qbase_ptr = reinterpret_cast<const char*>(p.q) + q_base * p.ld_bytes;
    const float* row_term = (METRIC == VDB_COSINE) ? p.row_nrm : p.row_sq;
    float* wr_a = As + srow * LW + skc;                 // this thread's staging slots (stage 0, pass 0)
    float* wr_b = Bs + srow * LW + skc;
    const float* rd_a = As + col * LW + half * 4;       // this lane's fragment rows (stage 0)
    const float* rd_b = Bs + (wave * 32 + col) * LW + half * 4;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;          // the operands in flight (plain variables: they must stay in registers)

    f32x16 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.0f;

    const int n_steps = (tile1 - tile0) * ksteps;
    int ck = 0, ct = tile0;                             // the step being multiplied
    // uniform bases of the step to fetch next (two steps ahead of the one being multiplied)
    int lk = 0, lt = tile0;
#define VDB_LOAD_STEP()                                                                     \
    do {                                                                                    \
        const char* abase = rows_w + (int64_t)min(lt, tile1 - 1) * tstride * TM * p.ld_bytes + lk * TKB; \
        const char* bbase = qbase_ptr + lk * TKB;                                           \
        ra0 = *reinterpret_cast<const float4*>(abase + offa[0]); rb0 = *reinterpret_cast<const float4*>(bbase + offb[0]); \
        ra1 = *reinterpret_cast<const float4*>(abase + offa[1]); rb1 = *reinterpret_cast<const float4*>(bbase + offb[1]); \
        ra2 = *reinterpret_cast<const float4*>(abase + offa[2]); rb2 = *reinterpret_cast<const float4*>(bbase + offb[2]); \
        ra3 = *reinterpret_cast<const float4*>(abase + offa[3]); rb3 = *reinterpret_cast<const float4*>(bbase + offb[3]); \
        if (++lk == ksteps) { lk = 0; ++lt; }                                               \
    } while (0)

    // prologue: step 0 into stage 0, step 1 on its way, row terms of the first tile
    if (n_steps > 0) {
        VDB_LOAD_STEP();
        *reinterpret_cast<float4*>(wr_a + 0 * 32 * LW) = ra0; *reinterpret_cast<float4*>(wr_b + 0 * 32 * LW) = rb0;
        *reinterpret_cast<float4*>(wr_a + 1 * 32 * LW) = ra1; *reinterpret_cast<float4*>(wr_b + 1 * 32 * LW) = rb1;
        *reinterpret_cast<float4*>(wr_a + 2 * 32 * LW) = ra2; *reinterpret_cast<float4*>(wr_b + 2 * 32 * LW) = rb2;
        *reinterpret_cast<float4*>(wr_a + 3 * 32 * LW) = ra3; *reinterpret_cast<float4*>(wr_b + 3 * 32 * LW) = rb3;
        if (METRIC != VDB_DOT && tid < TM) {
            float t = row_term[(int64_t)tile0 * tstride * TM + tid];
            if (METRIC == VDB_COSINE && t < 1e-8f) t = 1.0f;
            if (MINPASS && METRIC == VDB_COSINE) t = 1.0f / t;
            Rs[(tile0 & 1) * TM + tid] = t;
        }
        VDB_LOAD_STEP();
    }
    __syncthreads();

#define VDB_MFMA4(CH, S)                                                                                           \
    do {                                                                                                           \
        if (BF16) {                                                                                                \
            acc[S] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[CH][S]),                \
                                                             __builtin_bit_cast(bf16x8, bf[CH]), acc[S], 0, 0, 0); \
        } else {                                                                                                   \
            acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CH][S].x, bf[CH].x, acc[S], 0, 0, 0);                 \
            acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CH][S].y, bf[CH].y, acc[S], 0, 0, 0);                 \
            acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CH][S].z, bf[CH].z, acc[S], 0, 0, 0);                 \
            acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CH][S].w, bf[CH].w, acc[S], 0, 0, 0);                 \
        }                                                                                                          \
    } while (0)
#define VDB_RD_A(CH, S) af[CH][S] = *reinterpret_cast<const float4*>(ra_ + (S) * 32 * LW + (CH) * 8)
#define VDB_RD_B(CH) bf[CH] = *reinterpret_cast<const float4*>(rb_ + (CH) * 8)
#define VDB_WR_A(PS) *reinterpret_cast<float4*>(wa_ + (PS) * 32 * LW) = ra##PS
#define VDB_WR_B(PS) *reinterpret_cast<float4*>(wb_ + (PS) * 32 * LW) = rb##PS

    // Nearest-row pass: the finished tile's values are taken one accumulator at a time at the start of the next
    // tile's first step, each just before that accumulator's first matrix instructions of the new tile, so the vector
    // ALU work runs under the matrix instructions already issued.  (Taken at the end of the tile, all four waves of a
    // block, and the block sharing the CU, which runs in step with it, leave the matrix cores idle for its length:
    // a third of the time at four K steps per tile.)  Rows of one K step (ksteps == 1) keep the plain order: the next
    // tile's row terms are written at the end of the same step.
    const bool defer = ARGMIN && ksteps >= 2;
    auto nearest_of = [&](int s, int t) {
        if constexpr (ARGMIN && (METRIC == VDB_EUCLIDEAN || METRIC == VDB_EUCLIDEAN_SQ)) {
            const int64_t rbase = (int64_t)t * TM;
            if ((rbase + TM > p.n) || p.mask != nullptr)                    // the last, partial tile (or a filter)
                argmin16<METRIC, true>(acc[s], Rs + (t & 1) * TM, s, half, q_term, avail16(p.mask, rbase + s * 32, p.n, half),
                                       (int)rbase + s * 32 + 4 * half, best, best2, best_row);
            else
                argmin16<METRIC, false>(acc[s], Rs + (t & 1) * TM, s, half, q_term, 0xFFFFu, (int)rbase + s * 32 + 4 * half,
                                        best, best2, best_row);
        }
    };
#define VDB_HEAD(S)                                                     \
    do {                                                                \
        if constexpr (ARGMIN) {                                         \
            if (defer && ck == 0) {                                     \
                if (ct > tile0) nearest_of(S, ct - 1);                  \
                for (int e_ = 0; e_ < 16; ++e_) acc[S][e_] = 0.0f;      \
            }                                                           \
        }                                                               \
    } while (0)

    // two steps per trip so that the stage in use is a literal: every LDS address is base + immediate
    for (int st = 0; st < n_steps; st += 2) {
#pragma unroll
      for (int B_ = 0; B_ < 2; ++B_) {
        if (st + B_ >= n_steps) break;
        const float* ra_ = rd_a + B_ * (TM * LW);
        const float* rb_ = rd_b + B_ * (TQ * LW);
        float* wa_ = wr_a + (1 - B_) * (TM * LW);
        float* wb_ = wr_b + (1 - B_) * (TQ * LW);
        float4 bf[4], af[4][4];
        VDB_RD_B(0); VDB_RD_A(0, 0); VDB_RD_A(0, 1); VDB_RD_A(0, 2); VDB_RD_A(0, 3);
        VDB_FENCE();
        VDB_HEAD(0); VDB_MFMA4(0, 0); VDB_RD_B(1); VDB_RD_A(1, 0); VDB_FENCE();
        VDB_HEAD(1); VDB_MFMA4(0, 1); VDB_RD_A(1, 1); VDB_RD_A(1, 2); VDB_FENCE();
        VDB_HEAD(2); VDB_MFMA4(0, 2); VDB_RD_A(1, 3); VDB_RD_B(2); VDB_FENCE();
        VDB_HEAD(3); VDB_MFMA4(0, 3); VDB_RD_A(2, 0); VDB_RD_A(2, 1); VDB_FENCE();
        VDB_MFMA4(1, 0); VDB_RD_A(2, 2); VDB_RD_A(2, 3); VDB_FENCE();
        VDB_MFMA4(1, 1); VDB_RD_B(3); VDB_RD_A(3, 0); VDB_FENCE();
        VDB_MFMA4(1, 2); VDB_RD_A(3, 1); VDB_RD_A(3, 2); VDB_FENCE();
        VDB_MFMA4(1, 3); VDB_RD_A(3, 3); VDB_FENCE();
        // (the last two steps of a block park and fetch a step nobody multiplies: the last tile again)
        VDB_MFMA4(2, 0); VDB_WR_A(0); VDB_WR_B(0); VDB_FENCE();
        VDB_MFMA4(2, 1); VDB_WR_A(1); VDB_WR_B(1); VDB_FENCE();
        VDB_MFMA4(2, 2); VDB_WR_A(2); VDB_WR_B(2); VDB_FENCE();
        VDB_MFMA4(2, 3); VDB_WR_A(3); VDB_WR_B(3); VDB_FENCE();
        {
            const char* abase = rows_w + (int64_t)min(lt, tile1 - 1) * tstride * TM * p.ld_bytes + lk * TKB;
            const char* bbase = qbase_ptr + lk * TKB;
            // keep the per-thread offsets 32 bits wide here, so that every load is "uniform base +
            // 32-bit lane offset" in one instruction and no vector ALU work is left in the step
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) { asm("" : "+v"(offa[ps])); asm("" : "+v"(offb[ps])); }
            VDB_MFMA4(3, 0); ra0 = *reinterpret_cast<const float4*>(abase + offa[0]); rb0 = *reinterpret_cast<const float4*>(bbase + offb[0]); VDB_FENCE();
            VDB_MFMA4(3, 1); ra1 = *reinterpret_cast<const float4*>(abase + offa[1]); rb1 = *reinterpret_cast<const float4*>(bbase + offb[1]); VDB_FENCE();
            VDB_MFMA4(3, 2); ra2 = *reinterpret_cast<const float4*>(abase + offa[2]); rb2 = *reinterpret_cast<const float4*>(bbase + offb[2]); VDB_FENCE();
            VDB_MFMA4(3, 3); ra3 = *reinterpret_cast<const float4*>(abase + offa[3]); rb3 = *reinterpret_cast<const float4*>(bbase + offb[3]); VDB_FENCE();
            if (++lk == ksteps) { lk = 0; ++lt; }
        }
        if (ck == ksteps - 1) {
            // row terms of the next tile travel while this tile's survivors are picked
            float nxt = 1.0f;
            const bool has_next = ct + 1 < tile1;
            if (METRIC != VDB_DOT && has_next && tid < TM) nxt = row_term[(int64_t)(ct + 1) * tstride * TM + tid];
            if (MINPASS) {
                const int64_t rbase = (int64_t)ct * tstride * TM;
                const bool masked = (rbase + TM > p.n) || p.mask != nullptr;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t avail = masked ? avail16(p.mask, rbase + s * 32, p.n, half) : 0xFFFFu;
                    best = fminf(best, min_key16<METRIC>(acc[s], Rs + (ct & 1) * TM, s, half, q_term, avail));
                }
            } else if (ARGMIN) {
                if (!defer) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) nearest_of(s, ct);
                }
            } else {
                queue_epilogue<METRIC>(acc, Rs + (ct & 1) * TM, (int64_t)ct * TM, p.n, p.mask, half, bound, q_term,
                                       my_cnt, my_queue, p.cap);
            }
            if (METRIC != VDB_DOT && has_next && tid < TM) {
                if (METRIC == VDB_COSINE && nxt < 1e-8f) nxt = 1.0f;
                if (MINPASS && METRIC == VDB_COSINE) nxt = 1.0f / nxt;
                Rs[((ct + 1) & 1) * TM + tid] = nxt;
            }
            if (!(ARGMIN && defer)) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[s][e] = 0.0f;
            }
            ck = 0; ++ct;
        } else {
            ++ck;
        }
        __syncthreads();
      }
    }
    if (ARGMIN && defer && n_steps > 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) nearest_of(s, tile1 - 1);
    }
#undef VDB_HEAD
#undef VDB_LOAD_STEP
#undef VDB_MFMA4
#undef VDB_RD_A
#undef VDB_RD_B
#undef VDB_WR_A
#undef VDB_WR_B

    if (MINPASS) {
        // key -> distance, the way finish<>() ends; cosine goes through a reciprocal, hence the margin
        float d = best;
        if (best < INFINITY) {
            if (METRIC == VDB_EUCLIDEAN) d = sqrtf(fmaxf(best, 0.0f));
            else if (METRIC == VDB_EUCLIDEAN_SQ) d = fmaxf(best, 0.0f);
            else if (METRIC == VDB_COSINE) d = (1.0f - fminf(fmaxf(-best / q_term, -1.0f), 1.0f)) + 4e-6f;
        }
        if (q_ok) p.min_out[((int64_t)my_q * p.n_slabs + slab) * 2 + half] = d;
        return;
    }
    if (ARGMIN) {
        if (q_ok) p.arg_out[((int64_t)my_q * p.n_slabs + slab) * 2 + half] = make_float4(best, best2, __int_as_float(best_row), 0.0f);
        return;
    }
    if (tid < TQ && q_base + tid < p.nq) p.counts[(q_base + tid) * p.n_slabs + slab] = Cnt[tid];
}

// --------------------------------------------------------------------------
// Per query: exact distances of the queued survivors (the same arithmetic as
// finish<>() in the list kernel), the k best under the (distance, row) order.
struct QueueSelectParams {
    const float2* queue; const int* counts; int n_slabs, cap;
    const float* row_sq; const float* row_nrm; const float* q_sq; const float* q_nrm;
    int nq, k;
    float* out_d; int64_t* out_i; int* out_i32;
    int* overflow;                  // set to 1 when a queue held more survivors than it has slots
    int* ovf_mark;                  // optional [nq]: 1 for the queries whose queues did
};

template <int METRIC>
__global__ __launch_bounds__(256) void queue_select_kernel(QueueSelectParams p) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= p.nq) return;
    float q_term = 0.0f;
    if (METRIC == VDB_COSINE) {
        q_term = p.q_nrm[q];
        if (q_term < 1e-8f) q_term = INFINITY;
    } else if (METRIC != VDB_DOT) {
        q_term = p.q_sq[q];
    }
    WaveTopK top;
    top.init(p.k);
    bool over = false;
    for (int s0 = 0; s0 < p.n_slabs; s0 += 64) {
        // lane l walks the queue of slab s0 + l
        const int s = s0 + lane;
        int c = 0;
        if (s < p.n_slabs) {
            c = p.counts[(int64_t)q * p.n_slabs + s];
            if (c > p.cap) { over = true; c = p.cap; }
        }
        int cmax = c;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        const float2* e = p.queue + ((int64_t)q * p.n_slabs + min(s, p.n_slabs - 1)) * p.cap;
        for (int j0 = 0; j0 < cmax; j0 += 4) {
            float2 v[4];
            float rt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (j0 + u < c) ? e[j0 + u] : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rt[u] = 1.0f;
                if (METRIC != VDB_DOT && j0 + u < c) {
                    const int row = __float_as_int(v[u].y);
                    rt[u] = (METRIC == VDB_COSINE) ? p.row_nrm[row] : p.row_sq[row];
                    if (METRIC == VDB_COSINE && rt[u] < 1e-8f) rt[u] = 1.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = finish<METRIC>(v[u].x, rt[u], q_term);
                top.offer(j0 + u < c, d, __float_as_int(v[u].y), lane);
            }
        }
    }
    if (__ballot(over) != 0ull && lane == 0) {
        *p.overflow = 1;
        if (p.ovf_mark) p.ovf_mark[q] = 1;
    }
    if (lane < p.k) {
        const bool empty = top.eid == EMPTY_IDX;
        const int64_t at = (int64_t)q * p.k + lane;
        const int64_t row = empty ? -1 : (int64_t)top.eid;
        if (p.out_d) p.out_d[at] = empty ? INFINITY : top.e;
        if (p.out_i) p.out_i[at] = row;
        if (p.out_i32) p.out_i32[at] = (int)row;
    }
}

// --------------------------------------------------------------------------
// Selection behind a bf16 filter pass.  The queue of a query holds (dot product of the bf16 copies, row) for every
// row whose copy passed the loosened bound; the dot product of the fp32 values lies within `slack` of it.  In a
// key that is linear in the dot product and ordered like the distance (-dot; -dot / (|q||c|); (|q|^2+|c|^2) - 2 dot)
// the two differ by at most E.  With K the k-th smallest approximate key of the queue, k rows have an exact key of
// at most K + E, so every row of the exact k best (ties at the k-th distance included, and everything at or below the
// value where the distance is clamped) has an approximate key of at most max(K, clamp) + 2E.  Those few rows get
// their fp32 dot product recomputed: 32 rows at a time as the A operand of the same v_mfma_f32_32x32x2_f32 chain,
// in the same order over the columns, as flat_queue_kernel / flat_gemm_kernel run it, the query in every column of
// B.  Distances and their (distance, row) order are therefore bit for bit those of the fp32 search.
struct PrefilterSelectParams {
    float2* queue; const int* counts; int n_slabs, cap;
    const float* rows; int ld;
    const float* q; int ldq;
    const float* row_sq; const float* row_nrm; const float* q_sq; const float* q_nrm;
    const float* slack;
    int nq, k;
    float* out_d; int64_t* out_i; int* out_i32;
    int* overflow;
    int* ovf_mark;
};

template <int METRIC>
__global__ __launch_bounds__(256) void prefilter_select_kernel(PrefilterSelectParams p) {
    __shared__ int CandRows[4][96];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= p.nq) return;                      // no block-wide barrier below: every wave works alone
    volatile int* cand = CandRows[wave];
    float q_term = 0.0f;
    if (METRIC == VDB_COSINE) {
        q_term = p.q_nrm[q];
        if (q_term < 1e-8f) q_term = INFINITY;
    } else if (METRIC != VDB_DOT) {
        q_term = p.q_sq[q];
    }
    const float slack = p.slack[q];
    bool over = false;

    // ---- pass 1: approximate keys (written back over the dot products) and a value K that at least k rows of the
    // queue do not exceed.  With enough lanes at work K is the k-th smallest of the lanes' own minima (different rows,
    // so k rows lie at or below it; at most a row or two looser than the k-th smallest key, and one sort of 64 values
    // instead of one merge per batch of entries); otherwise the k-th smallest key itself.
    WaveTopK approx;
    approx.init(p.k);
    const bool by_lane_minima = min(p.n_slabs, 64) >= 2 * p.k;
    float lane_min = INFINITY;
    int lane_row = EMPTY_IDX;
    for (int s0 = 0; s0 < p.n_slabs; s0 += 64) {
        const int s = s0 + lane;
        int c = 0;
        if (s < p.n_slabs) {
            c = p.counts[(int64_t)q * p.n_slabs + s];
            if (c > p.cap) { over = true; c = p.cap; }
        }
        int cmax = c;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        float2* e = p.queue + ((int64_t)q * p.n_slabs + min(s, p.n_slabs - 1)) * p.cap;
        for (int j0 = 0; j0 < cmax; j0 += 4) {
            float2 v[4];
            float rt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (j0 + u < c) ? e[j0 + u] : make_float2(0.0f, __int_as_float(EMPTY_IDX));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rt[u] = 1.0f;
                if (METRIC != VDB_DOT && j0 + u < c) {
                    const int row = __float_as_int(v[u].y);
                    rt[u] = (METRIC == VDB_COSINE) ? p.row_nrm[row] : p.row_sq[row];
                    if (METRIC == VDB_COSINE && rt[u] < 1e-8f) rt[u] = 1.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool have = j0 + u < c;
                float key = (METRIC == VDB_DOT) ? -v[u].x
                          : (METRIC == VDB_COSINE) ? -(v[u].x / (q_term * rt[u]))
                                                   : __fmaf_rn(-2.0f, v[u].x, q_term + rt[u]);
                if (have) e[j0 + u].x = key;
                if (by_lane_minima) {
                    if (have && key < lane_min) { lane_min = key; lane_row = __float_as_int(v[u].y); }
                } else {
                    approx.offer(have, have ? key : INFINITY, __float_as_int(v[u].y), lane);
                }
            }
        }
    }
    if (by_lane_minima) approx.offer(lane_min < INFINITY, lane_min, lane_row, lane);
    // the largest approximate key a row of the exact k best can have
    float limit = approx.thr;
    if (METRIC == VDB_DOT) {
        limit = limit + 2.0f * slack;
    } else if (METRIC == VDB_COSINE) {
        // E = slack / |q| plus the roundings of the quotient; distances are clamped to [0, 2]: keys <= -1 tie at 0
        const float E = slack / q_term + 1e-6f;
        limit = fmaxf(limit, -1.0f) + 2.0f * E;
        if (limit >= 1.0f - 1e-6f && limit < INFINITY) over = true;     // ties at distance 2: every row would be needed
    } else {
        // E = 2 slack; squared distances are clamped at 0
        limit = fmaxf(limit, 0.0f) + 4.0f * slack;
        limit = limit + fabsf(limit) * 2e-6f;
    }

    // ---- pass 2: rows within the limit, 32 at a time through the fp32 matrix instruction chain
    WaveTopK top;
    top.init(p.k);
    const int col = lane & 31, half = lane >> 5;
    const float* qrow = p.q + (int64_t)q * p.ldq + half * 4;
    const int ksteps = (p.ld * 4 + TKB - 1) / TKB;
    auto refine = [&](int cnt) {
        const int my_row = cand[col < cnt ? col : 0];
        const float* a = p.rows + (int64_t)my_row * p.ld + half * 4;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        // The rows are gathered from all over the store: every column step's loads take a trip to HBM.  Four steps
        // are kept in flight (a ring of four operand sets, the loop unrolled by four so that the set in use is a
        // literal); a step past the last one loads nothing and multiplies nothing.
        float4 fa[4][4], fb[4][4];
        auto fetch = [&](int ks, float4 (&xa)[4], float4 (&xb)[4]) {
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                const int off = ks * (TKB / 4) + c2 * 8;
                const bool in = ks < ksteps && off + half * 4 < p.ld;   // the pitch is a multiple of 4 floats: all or nothing
                xa[c2] = in ? *reinterpret_cast<const float4*>(a + off) : make_float4(0, 0, 0, 0);
                xb[c2] = in ? *reinterpret_cast<const float4*>(qrow + off) : make_float4(0, 0, 0, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(u, fa[u], fb[u]);
        for (int ks0 = 0; ks0 < ksteps; ks0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks0 + u < ksteps) {
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][c2].x, fb[u][c2].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][c2].y, fb[u][c2].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][c2].z, fb[u][c2].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][c2].w, fb[u][c2].w, acc, 0, 0, 0);
                    }
                }
                fetch(ks0 + u + 4, fa[u], fb[u]);
            }
        }
        // row i of the tile sits in value 4 (i / 8) + i % 4 of the lanes whose half is (i / 4) % 2; every column
        // holds the same query, so lane (column i, that half) takes row i
        const bool own = half == ((col >> 2) & 1) && col < cnt;
        const float dot = pick16(acc, 4 * (col >> 3) + (col & 3));
        float rt = 1.0f;
        if (METRIC != VDB_DOT) {
            rt = (METRIC == VDB_COSINE) ? p.row_nrm[my_row] : p.row_sq[my_row];
            if (METRIC == VDB_COSINE && rt < 1e-8f) rt = 1.0f;
        }
        top.offer(own, finish<METRIC>(dot, rt, q_term), my_row, lane);
    };
    int nc = 0;
    for (int s0 = 0; s0 < p.n_slabs; s0 += 64) {
        const int s = s0 + lane;
        int c = 0;
        if (s < p.n_slabs) c = min(p.counts[(int64_t)q * p.n_slabs + s], p.cap);
        int cmax = c;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        const float2* e = p.queue + ((int64_t)q * p.n_slabs + min(s, p.n_slabs - 1)) * p.cap;
        for (int j0 = 0; j0 < cmax; j0 += 4) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (j0 + u < c) ? e[j0 + u] : make_float2(INFINITY, 0.0f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool want = j0 + u < c && v[u].x <= limit;
                const unsigned long long m = __ballot(want);
                if (m == 0) continue;
                if (want) cand[nc + __popcll(m & ((1ull << lane) - 1ull))] = __float_as_int(v[u].y);
                nc += __popcll(m);
                while (nc >= 32) {
                    refine(32);
                    const int moved = (lane < nc - 32) ? cand[32 + lane] : 0;
                    if (lane < nc - 32) cand[lane] = moved;
                    nc -= 32;
                }
            }
        }
    }
    if (nc > 0) refine(nc);

    if (__ballot(over) != 0ull && lane == 0) {
        *p.overflow = 1;
        if (p.ovf_mark) p.ovf_mark[q] = 1;
    }
    if (lane < p.k) {
        const bool empty = top.eid == EMPTY_IDX;
        const int64_t at = (int64_t)q * p.k + lane;
        const int64_t row = empty ? -1 : (int64_t)top.eid;
        if (p.out_d) p.out_d[at] = empty ? INFINITY : top.e;
        if (p.out_i) p.out_i[at] = row;
        if (p.out_i32) p.out_i32[at] = (int)row;
    }
}

// slack[q]: the largest |dot(bf16(q), bf16(c)) - dot(q, c)| over the rows c, both sums as the matrix cores form them.
// bf16 keeps eight significant bits: rounding to nearest moves an element by at most 2^-8 of itself (or 2^-126 where
// the format runs out of exponent), so the sum of products moves by at most (2^-7 + 2^-16) sum|q_i c_i| <=
// 0.0078278 |q||c|; the two accumulations in fp32 add at most dim 2^-22 |q||c| between them.  Cosine compares dot >= bound |c|, the others use the largest
// row norm.  Products that could overflow, or norms that are not numbers, send the search to the fp32 kernel.
__global__ void prefilter_slack_kernel(int metric, const float* __restrict__ q_nrm, const float* __restrict__ max_row_nrm,
                                       int nq, int dim, float* __restrict__ slack, float* __restrict__ widen_thr,
                                       int* __restrict__ overflow, int* __restrict__ ovf_mark) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float qn = q_nrm[q];
    const float big = *max_row_nrm;
    const float rel = (0.0078278f + (float)dim * 2.4e-7f) * 1.0001f;
    const float prod = qn * big;
    if (!(prod < 1e37f)) {
        *overflow = 1;
        if (ovf_mark) ovf_mark[q] = 1;
    }
    const float tiny = sqrtf((float)dim) * 1.2e-38f * (qn + big) + 1e-30f;
    const float sl = (metric == VDB_COSINE) ? rel * qn : rel * prod + tiny;
    slack[q] = sl;
    if (widen_thr) {
        // a distance formed from a dot product that is off by at most `sl`: the exact one is at most ...
        float t = widen_thr[q];
        if (t < INFINITY) {
            if (metric == VDB_DOT) t = t + sl + fabsf(t) * 1e-6f;                          // -dot
            else if (metric == VDB_COSINE) t = t + (qn >= 1e-8f ? rel : 0.0f) + 2e-6f;     // 1 - dot / (|q||c|), clamped
            else if (metric == VDB_EUCLIDEAN_SQ) t = (t + 2.0f * sl) * 1.000001f;          // max(s - 2 dot, 0)
            else t = sqrtf(t * t * 1.000001f + 2.0f * sl) * 1.000001f;                     // its square root
            widen_thr[q] = t;
        }
    }
}

__global__ void max_norm_kernel(const float* __restrict__ nrm, int64_t n0, int64_t n1, float* __restrict__ out) {
    // non-negative floats (and NaNs above them) order like their bit patterns
    int m = 0;
    for (int64_t i = n0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (int64_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_int(nrm[i]) & 0x7fffffff);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(reinterpret_cast<int*>(out), m);
}

int launch_prefilter_slack(int metric, const float* q_nrm, const float* max_row_nrm, int nq, int dim, float* slack,
                           float* widen_thr, int* overflow, hipStream_t s, int* ovf_mark) {
    hipLaunchKernelGGL(prefilter_slack_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, metric, q_nrm, max_row_nrm, nq, dim,
                       slack, widen_thr, overflow, ovf_mark);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_max_norm(const float* row_nrm, int64_t n0, int64_t n1, float* max_nrm, hipStream_t s) {
    if (n1 <= n0) return VDB_OK;
    const int blocks = (int)std::min<int64_t>((n1 - n0 + 255) / 256, 1024);
    hipLaunchKernelGGL(max_norm_kernel, dim3(blocks), dim3(256), 0, s, row_nrm, n0, n1, max_nrm);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// Bound pass: per query, the k-th smallest of its per-lane minima over the sampled tiles.  Every
// minimum is the distance of a real row, and minima of different lanes belong to different rows, so
// the k-th smallest of them is at least the k-th best distance of the whole row set.
__global__ __launch_bounds__(256) void kth_of_minima_kernel(const float* __restrict__ mins, int n_lists, int nq, int k,
                                                            float* __restrict__ thr, int* __restrict__ clear,
                                                            int* __restrict__ clear_marks) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (clear && blockIdx.x == 0 && threadIdx.x == 0) *clear = 0;      // the overflow flag of the pass that follows
    if (q >= nq) return;
    if (clear_marks && lane == 0) clear_marks[q] = 0;                   // and its per-query marks
    WaveTopK top;
    top.init(k);
    for (int l0 = 0; l0 < n_lists; l0 += 64) {
        const int l = l0 + lane;
        const float v = l < n_lists ? mins[(int64_t)q * n_lists + l] : INFINITY;
        top.offer(l < n_lists && v < INFINITY, v, l, lane);
    }
    if (lane == 0) thr[q] = top.thr;        // +inf while fewer than k lanes saw a row
}

bool flat_queue_fast_ok(const Rows& r) {
    static const bool no_fast = getenv("VDB_NO_QFAST") != nullptr;
    const int ld_bytes = r.bf16 ? r.ld16 * 2 : r.ld * 4;
    return !no_fast && ld_bytes % TKB == 0 && r.cap >= ((r.n + WM - 1) / WM) * WM;
}

int flat_min_sample_lists(int sample_tiles, int nq, int k, int* n_slabs) {
    int nqt, tps;
    plan_grid(sample_tiles, nq, ctx().num_cus, &nqt, n_slabs, &tps);
    (void)k;
    return 2 * *n_slabs;
}

template <bool BF16>
static int launch_min_pass(int metric, QueueParams& p, hipStream_t s) {
    const int groups = (p.n_slabs + 7) / 8;
    const dim3 grid(p.n_slabs >= 8 ? groups * 8 * p.n_qtiles : p.n_slabs * p.n_qtiles);
    const size_t lds = (size_t)(2 * TM * LW + 2 * TQ * LW + 2 * TM + TQ) * sizeof(float);
#define VDB_LAUNCH_MIN(M)                                                                      \
    case M: {                                                                                  \
        auto kern = flat_queue_fast_kernel<M, BF16, QMODE_MINPASS>;                            \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern),                       \
                                    lds));    \
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);                                   \
        break;                                                                                 \
    }
    switch (metric) {
        VDB_LAUNCH_MIN(VDB_EUCLIDEAN)
        VDB_LAUNCH_MIN(VDB_EUCLIDEAN_SQ)
        VDB_LAUNCH_MIN(VDB_COSINE)
        VDB_LAUNCH_MIN(VDB_DOT)
        default:
            set_error("metric %d has no matrix-core path", metric);
            return VDB_ERR_INVALID;
    }
#undef VDB_LAUNCH_MIN
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_flat_min_sample(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm, int nq, int k,
                           const uint32_t* mask, int sample_tiles, int tile_stride, float* mins, float* thr, int* clear_flag, hipStream_t s,
                           int* clear_marks) {
    VDB_REQUIRE(flat_queue_fast_ok(r) && sample_tiles >= 1 && k >= 1 && k <= 64, "bad bound pass arguments");
    QueueParams p{};
    p.rows = r.bf16 ? (const void*)r.data16 : (const void*)r.data; p.n = r.n;
    p.ld_bytes = r.bf16 ? r.ld16 * 2 : r.ld * 4; p.row_sq = r.sq; p.row_nrm = r.nrm;
    p.q = q; p.nq = nq; p.q_sq = q_sq; p.q_nrm = q_nrm; p.mask = mask;
    p.n_tiles = sample_tiles; p.tile_stride = tile_stride; p.min_out = mins;
    plan_grid(p.n_tiles, nq, ctx().num_cus, &p.n_qtiles, &p.n_slabs, &p.tiles_per_slab);
    VDB_TRY(r.bf16 ? launch_min_pass<true>(metric, p, s) : launch_min_pass<false>(metric, p, s));
    hipLaunchKernelGGL(kth_of_minima_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, mins, 2 * p.n_slabs, nq, k, thr, clear_flag, clear_marks);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// Nearest row of every query (k = 1, Euclidean forms, no filter): the branch-free kernel with a per-lane running
// (smallest key, second smallest, row) instead of queues; then per query the smallest over its lanes' entries.
// The keys are the list kernel's own pre-clamp values (same matrix instructions in the same order, same fma), so the
// smallest key names the row the list kernel's argmin of sqrt(max(key, 0)) names, unless keys are so close that the
// clamp or the square root can make them equal (then the lower row wins there): a second key within 2^-20 of the
// smallest, or a smallest key that is not positive, puts the query on a list (*count entries wanted, the first `lim`
// kept), and the caller searches the listed ones again with the list kernel.
__global__ __launch_bounds__(256) void argmin_finish_kernel(const float4* __restrict__ arg, int lists, int nq, int take_sqrt,
                                                            float* __restrict__ out_d, int64_t* __restrict__ out_i,
                                                            int* __restrict__ out_i32, int k_stride,
                                                            int* __restrict__ list, int* __restrict__ count, int lim) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float b1 = INFINITY, b2 = INFINITY;
    int bi = -1;
    for (int l = 0; l < lists; ++l) {
        const float4 e = arg[(int64_t)q * lists + l];
        const int row = __float_as_int(e.z);
        const bool less = e.x < b1 || (e.x == b1 && row >= 0 && (bi < 0 || row < bi));
        // the second smallest of everything seen: the old smallest if it is displaced, else the entry's smallest
        const float other = less ? b1 : e.x;
        b2 = fminf(fminf(b2, e.y), other);
        if (less) { b1 = e.x; bi = row; }
    }
    const bool unsure = !(b1 > 0.0f) || !(b2 > b1 + fabsf(b1) * 9.5367431640625e-07f) || bi < 0;
    if (unsure) {                                   // (in any order: every query of the list gets its own result back)
        const int slot = atomicAdd(count, 1);
        if (slot < lim) list[slot] = q;
    }
    const float sq = fmaxf(b1, 0.0f);
    const int64_t at = (int64_t)q * k_stride;
    if (out_d) out_d[at] = bi < 0 ? INFINITY : (take_sqrt ? sqrtf(sq) : sq);
    if (out_i) out_i[at] = bi;
    if (out_i32) out_i32[at] = bi;
}

int flat_argmin_lists(const Rows& r, int nq) {
    int nqt, ns, tps;
    plan_grid((int)((r.n + TM - 1) / TM), nq, ctx().num_cus, &nqt, &ns, &tps);
    return 2 * ns;
}

int launch_flat_argmin(const Rows& r, int metric, const void* q, const float* q_sq, int nq, void* arg_buf,
                       float* out_d, int64_t* out_i, int* out_i32, int out_stride, int* list, int* count, int lim, hipStream_t s) {
    VDB_REQUIRE(!r.bf16 && flat_queue_fast_ok(r) && (metric == VDB_EUCLIDEAN || metric == VDB_EUCLIDEAN_SQ) && nq >= 1 && r.n >= 1,
                "bad nearest-row pass arguments");
    QueueParams p{};
    p.rows = r.data; p.n = r.n; p.ld_bytes = r.ld * 4; p.row_sq = r.sq; p.row_nrm = r.nrm;
    p.q = q; p.nq = nq; p.q_sq = q_sq; p.q_nrm = nullptr; p.mask = nullptr;
    p.n_tiles = (int)((r.n + TM - 1) / TM); p.tile_stride = 1;
    p.arg_out = reinterpret_cast<float4*>(arg_buf);
    plan_grid(p.n_tiles, nq, ctx().num_cus, &p.n_qtiles, &p.n_slabs, &p.tiles_per_slab);
    const int groups = (p.n_slabs + 7) / 8;
    const dim3 grid(p.n_slabs >= 8 ? groups * 8 * p.n_qtiles : p.n_slabs * p.n_qtiles);
    const size_t lds = (size_t)(2 * TM * LW + 2 * TQ * LW + 2 * TM + TQ) * sizeof(float);
    {
        KernelTimer timer(s, true);
        if (metric == VDB_EUCLIDEAN) {
            auto kern = flat_queue_fast_kernel<VDB_EUCLIDEAN, false, QMODE_ARGMIN>;
            VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
            hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
        } else {
            auto kern = flat_queue_fast_kernel<VDB_EUCLIDEAN_SQ, false, QMODE_ARGMIN>;
            VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
            hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
        }
    }
    hipLaunchKernelGGL(argmin_finish_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, p.arg_out, 2 * p.n_slabs, nq,
                       metric == VDB_EUCLIDEAN ? 1 : 0, out_d, out_i, out_i32, out_stride, list, count, lim);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int flat_queue_plan(const Rows& r, int nq, int k, int sample_stride, int* n_slabs, int* cap, bool loosened) {
    int nqt, tps;
    if (queue_use_wide(r)) plan_grid_wide((int)((r.n + WM - 1) / WM), nq, ctx().num_cus, &nqt, n_slabs, &tps);
    else plan_grid((int)((r.n + TM - 1) / TM), nq, ctx().num_cus, &nqt, n_slabs, &tps);
    // survivors per query: about sample_stride * k in expectation (the bound is the k-th best of
    // one row in sample_stride); four times that plus a constant, spread over the slabs
    // Behind a bf16 filter both the bound and the test are loosened by the rounding slack: two to three times as
    // many survivors on Gaussian rows, and a query whose sampled bound is loose on top of that must still fit the
    // slots of every slab (a full queue costs the whole batch a second search).
    const int64_t total = loosened ? 16LL * sample_stride * k + 2048 : 4LL * sample_stride * k + 1024;
    int c = (int)((total + *n_slabs - 1) / *n_slabs);
    c = std::max(32, (c + 7) / 8 * 8);
    *cap = c;
    return VDB_OK;
}

template <bool BF16>
static int launch_queue(int metric, QueueParams& p, bool fast, hipStream_t s) {
    const int groups = (p.n_slabs + 7) / 8;
    const dim3 grid(p.n_slabs >= 8 ? groups * 8 * p.n_qtiles : p.n_slabs * p.n_qtiles);
    const size_t lds = (size_t)(2 * TM * LW + 2 * TQ * LW + 2 * TM + TQ) * sizeof(float);
    KernelTimer timer(s, true);
#define VDB_LAUNCH_QUEUE(M)                                                                    \
    case M: {                                                                                  \
        auto kern = fast ? flat_queue_fast_kernel<M, BF16, QMODE_QUEUE> : flat_queue_kernel<M, BF16>; \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern),                       \
                                    lds));    \
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);                                   \
        break;                                                                                 \
    }
    switch (metric) {
        VDB_LAUNCH_QUEUE(VDB_EUCLIDEAN)
        VDB_LAUNCH_QUEUE(VDB_EUCLIDEAN_SQ)
        VDB_LAUNCH_QUEUE(VDB_COSINE)
        VDB_LAUNCH_QUEUE(VDB_DOT)
        default:
            set_error("metric %d has no matrix-core path", metric);
            return VDB_ERR_INVALID;
    }
#undef VDB_LAUNCH_QUEUE
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_flat_queue(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm, int nq, int k,
                      const uint32_t* mask, const float* thr, int n_slabs, int cap, void* queue, int* counts,
                      int* overflow, float* out_d, int64_t* out_i, int* out_i32, hipStream_t s, const PrefilterArgs* pf,
                      int* ovf_mark) {
    VDB_REQUIRE(k >= 1 && k <= 64, "queue selection supports 1 <= k <= 64, got %d", k);
    VDB_REQUIRE(r.n >= 1 && r.n < (1LL << 31) && nq >= 1 && thr != nullptr, "bad queue search arguments");
    QueueParams p{};
    p.rows = r.bf16 ? (const void*)r.data16 : (const void*)r.data; p.n = r.n;
    p.ld_bytes = r.bf16 ? r.ld16 * 2 : r.ld * 4; p.row_sq = r.sq; p.row_nrm = r.nrm;
    p.q = q; p.nq = nq; p.q_sq = q_sq; p.q_nrm = q_nrm; p.mask = mask; p.thr = thr;
    p.slack = pf ? pf->slack : nullptr;
    const bool wide = queue_use_wide(r);
    if (wide) {
        p.n_tiles = (int)((r.n + WM - 1) / WM);
        plan_grid_wide(p.n_tiles, nq, ctx().num_cus, &p.n_qtiles, &p.n_slabs, &p.tiles_per_slab);
    } else {
        p.n_tiles = (int)((r.n + TM - 1) / TM);
        plan_grid(p.n_tiles, nq, ctx().num_cus, &p.n_qtiles, &p.n_slabs, &p.tiles_per_slab);
    }
    VDB_REQUIRE(p.n_slabs == n_slabs, "queue plan mismatch (%d slabs planned, %d now)", n_slabs, p.n_slabs);
    p.cap = cap; p.queue = reinterpret_cast<float2*>(queue); p.counts = counts;
    // whole tiles readable and no partial K step: the branch-free kernel
    const bool fast = flat_queue_fast_ok(r);
    if (wide) VDB_TRY(launch_queue_wide(metric, p, s));
    else VDB_TRY(r.bf16 ? launch_queue<true>(metric, p, fast, s) : launch_queue<false>(metric, p, fast, s));

    if (pf) {
        PrefilterSelectParams fp{};
        fp.queue = p.queue; fp.counts = counts; fp.n_slabs = n_slabs; fp.cap = cap;
        fp.rows = pf->rows32; fp.ld = pf->ld32; fp.q = pf->q32; fp.ldq = pf->ldq32;
        fp.row_sq = r.sq; fp.row_nrm = r.nrm; fp.q_sq = q_sq; fp.q_nrm = q_nrm; fp.slack = pf->slack;
        fp.nq = nq; fp.k = k; fp.out_d = out_d; fp.out_i = out_i; fp.out_i32 = out_i32; fp.overflow = overflow; fp.ovf_mark = ovf_mark;
        const dim3 fgrid((nq + 3) / 4);
        switch (metric) {
            case VDB_EUCLIDEAN: hipLaunchKernelGGL(prefilter_select_kernel<VDB_EUCLIDEAN>, fgrid, dim3(256), 0, s, fp); break;
            case VDB_EUCLIDEAN_SQ: hipLaunchKernelGGL(prefilter_select_kernel<VDB_EUCLIDEAN_SQ>, fgrid, dim3(256), 0, s, fp); break;
            case VDB_COSINE: hipLaunchKernelGGL(prefilter_select_kernel<VDB_COSINE>, fgrid, dim3(256), 0, s, fp); break;
            default: hipLaunchKernelGGL(prefilter_select_kernel<VDB_DOT>, fgrid, dim3(256), 0, s, fp); break;
        }
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    QueueSelectParams sp{};
    sp.queue = p.queue; sp.counts = counts; sp.n_slabs = n_slabs; sp.cap = cap;
    sp.row_sq = r.sq; sp.row_nrm = r.nrm; sp.q_sq = q_sq; sp.q_nrm = q_nrm; sp.nq = nq; sp.k = k;
    sp.out_d = out_d; sp.out_i = out_i; sp.out_i32 = out_i32; sp.overflow = overflow; sp.ovf_mark = ovf_mark;
    const dim3 sgrid((nq + 3) / 4);
    switch (metric) {
        case VDB_EUCLIDEAN: hipLaunchKernelGGL(queue_select_kernel<VDB_EUCLIDEAN>, sgrid, dim3(256), 0, s, sp); break;
        case VDB_EUCLIDEAN_SQ: hipLaunchKernelGGL(queue_select_kernel<VDB_EUCLIDEAN_SQ>, sgrid, dim3(256), 0, s, sp); break;
        case VDB_COSINE: hipLaunchKernelGGL(queue_select_kernel<VDB_COSINE>, sgrid, dim3(256), 0, s, sp); break;
        default: hipLaunchKernelGGL(queue_select_kernel<VDB_DOT>, sgrid, dim3(256), 0, s, sp); break;
    }
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
