// Candidate-queue kernel on 256 x 256 tiles, for bf16 rows.
//
// With v_mfma_f32_32x32x16_bf16 a 128 x 128 tile finishes a 128-byte K step in a
// quarter of the fp32 time while it stages the same 32 KB per step: the kernel of
// flat_queue.hip is then bound by the bytes it moves (L2 -> LDS, LDS -> registers),
// not by the matrix cores.  Here a block of 8 waves owns 256 database rows x 256
// queries: a K step stages 64 KB for four times the matrix work (half the bytes per
// flop), a wave multiplies 128 rows x 64 queries (8 accumulators, 6 fragment reads
// per 8 matrix instructions instead of 5 per 4).  One block per CU (147 KB of LDS for
// two stages), so both waves of a SIMD belong to it and meet every barrier together:
// the step's barrier sits before its last quarter of matrix instructions, and the
// first fragments of the coming step are fetched behind it.  The global loads of the
// step after next are issued right behind the barrier and parked in LDS in the quarter
// before the next one: three quarters of a step in flight.  The kernel is bound by
// that delivery (64 KB per step and CU out of L2 and HBM), not by the matrix cores:
// with three of four matrix instructions removed a step takes 82 % of its full time.
//
// Everything else — the conservative filter, the queues, queue_select_kernel, the
// overflow flag and the list-kernel fallback — is shared with flat_queue.hip
// (distance/metrics.py:453-484, index/flat.py:354-362).
#include "flat_queue.h"

namespace vdb {

#define VDB_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int METRIC, bool BF16>
__global__ __launch_bounds__(WNT, 1) void flat_queue_wide_kernel(QueueParams p) {
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                                   // [2][WM][LW]  database rows, two stages
    float* Bs = As + 2 * WM * LW;                       // [2][WQ][LW]  queries, two stages
    float* Rs = Bs + 2 * WQ * LW;                       // [2][WM]      per-row term, by tile parity
    int* Cnt = reinterpret_cast<int*>(Rs + 2 * WM);     // [WQ]         survivors per query of this block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int wr = wave >> 2;                           // which 128 rows of the tile
    const int wq = wave & 3;                            // which 64 queries of the tile

    // blocks that share a slab of rows (one per query tile) get the same blockIdx % 8: one XCD, one L2
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

    // a lane owns two queries: column `col` of each of its wave's two 32-query groups
    float q_term[2], bound[2];
    float2* my_queue[2];
    int* my_cnt[2];
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
        const int ql = wq * 64 + qd * 32 + col;
        const int my_q = qt * WQ + ql;
        const int my_q_ld = min(my_q, p.nq - 1);
        q_term[qd] = 0.0f;
        if (METRIC == VDB_COSINE) {
            q_term[qd] = p.q_nrm[my_q_ld];
            if (q_term[qd] < 1e-8f) q_term[qd] = INFINITY;       // metrics.py:472-473: all ones
        } else if (METRIC != VDB_DOT) {
            q_term[qd] = p.q_sq[my_q_ld];
        }
        bound[qd] = filter_bound<METRIC>(p.thr[my_q_ld], q_term[qd]);
        if (p.slack) bound[qd] = loosen_bound<METRIC>(bound[qd], p.slack[my_q_ld]);
        if (my_q >= p.nq) bound[qd] = (METRIC == VDB_DOT || METRIC == VDB_COSINE) ? INFINITY : -INFINITY;
        my_queue[qd] = p.queue + ((int64_t)my_q_ld * p.n_slabs + slab) * p.cap;
        my_cnt[qd] = &Cnt[ql];
    }
    if (tid < WQ) Cnt[tid] = 0;

    const int ksteps = p.ld_bytes / TKB;
    const int srow = tid >> 3;          // staging: row within a 64-row pass
    const int skc = (tid & 7) * 4;      // staging: dword offset inside the K step (16-byte pieces)
    const int64_t q_base = (int64_t)qt * WQ;
    const int ldw = p.ld_bytes >> 2;
    uint32_t offb[4], offa[4];          // byte offsets from a wave-uniform base
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t gq = min(q_base + ps * 64 + srow, (int64_t)p.nq - 1) - q_base;
        offb[ps] = (uint32_t)(gq * ldw + skc) * 4u;
        offa[ps] = (uint32_t)((ps * 64 + srow) * ldw + skc) * 4u;
    }
    const char* rows_w = reinterpret_cast<const char*>(p.rows);
    const char* qbase_ptr = reinterpret_cast<const char*>(p.q) + q_base * p.ld_bytes;
    const float* row_term = (METRIC == VDB_COSINE) ? p.row_nrm : p.row_sq;
    float* wr_a = As + srow * LW + skc;                           // this thread's staging slots (stage 0, pass 0)
    float* wr_b = Bs + srow * LW + skc;
    const float* rd_a = As + (wr * 128 + col) * LW + half * 4;    // this lane's fragment rows (stage 0)
    const float* rd_b = Bs + (wq * 64 + col) * LW + half * 4;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;                // the operands in flight

    f32x16 acc[2][4];                                             // [query group][32-row sub-tile]
#pragma unroll
    for (int qd = 0; qd < 2; ++qd)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[qd][r][e] = 0.0f;

    const int n_steps = (tile1 - tile0) * ksteps;
    int ck = 0, ct = tile0;             // the step being multiplied
    int lk = 0, lt = tile0;             // the step to fetch next
#define VDB_LOAD_STEP()                                                                        \
    do {                                                                                       \
        const char* abase = rows_w + (int64_t)min(lt, tile1 - 1) * WM * p.ld_bytes + lk * TKB; \
        const char* bbase = qbase_ptr + lk * TKB;                                              \
        ra0 = *reinterpret_cast<const float4*>(abase + offa[0]); rb0 = *reinterpret_cast<const float4*>(bbase + offb[0]); \
        ra1 = *reinterpret_cast<const float4*>(abase + offa[1]); rb1 = *reinterpret_cast<const float4*>(bbase + offb[1]); \
        ra2 = *reinterpret_cast<const float4*>(abase + offa[2]); rb2 = *reinterpret_cast<const float4*>(bbase + offb[2]); \
        ra3 = *reinterpret_cast<const float4*>(abase + offa[3]); rb3 = *reinterpret_cast<const float4*>(bbase + offb[3]); \
        if (++lk == ksteps) { lk = 0; ++lt; }                                                  \
    } while (0)

    if (n_steps > 0) {
        VDB_LOAD_STEP();
        *reinterpret_cast<float4*>(wr_a + 0 * 64 * LW) = ra0; *reinterpret_cast<float4*>(wr_b + 0 * 64 * LW) = rb0;
        *reinterpret_cast<float4*>(wr_a + 1 * 64 * LW) = ra1; *reinterpret_cast<float4*>(wr_b + 1 * 64 * LW) = rb1;
        *reinterpret_cast<float4*>(wr_a + 2 * 64 * LW) = ra2; *reinterpret_cast<float4*>(wr_b + 2 * 64 * LW) = rb2;
        *reinterpret_cast<float4*>(wr_a + 3 * 64 * LW) = ra3; *reinterpret_cast<float4*>(wr_b + 3 * 64 * LW) = rb3;
        if (METRIC != VDB_DOT && tid < WM) {
            float t = row_term[(int64_t)tile0 * WM + tid];
            if (METRIC == VDB_COSINE && t < 1e-8f) t = 1.0f;        // metrics.py:476
            Rs[(tile0 & 1) * WM + tid] = t;
        }
        VDB_LOAD_STEP();
    }
    __syncthreads();

    // fragment sets: X holds the K chunk being multiplied, Y the next one
    float4 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;
    xb0 = *reinterpret_cast<const float4*>(rd_b); xb1 = *reinterpret_cast<const float4*>(rd_b + 32 * LW);
    xa0 = *reinterpret_cast<const float4*>(rd_a); xa1 = *reinterpret_cast<const float4*>(rd_a + 32 * LW);
    xa2 = *reinterpret_cast<const float4*>(rd_a + 64 * LW); xa3 = *reinterpret_cast<const float4*>(rd_a + 96 * LW);

#define VDB_MM(A, B, QD, R)                                                                                       \
    do {                                                                                                          \
        if (BF16) {                                                                                               \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A),                   \
                                                                 __builtin_bit_cast(bf16x8, B), acc[QD][R], 0, 0, 0); \
        } else {                                                                                                  \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, acc[QD][R], 0, 0, 0);                     \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, acc[QD][R], 0, 0, 0);                     \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, acc[QD][R], 0, 0, 0);                     \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, acc[QD][R], 0, 0, 0);                     \
        }                                                                                                         \
    } while (0)
    // fragment reads of K chunk CH from the stage at (PA, PB)
#define VDB_RA(DST, PA, CH, R) DST = *reinterpret_cast<const float4*>(PA + (R) * 32 * LW + (CH) * 8)
#define VDB_RB(DST, PB, CH, QD) DST = *reinterpret_cast<const float4*>(PB + (QD) * 32 * LW + (CH) * 8)

    for (int st = 0; st < n_steps; st += 2) {
#pragma unroll
      for (int B_ = 0; B_ < 2; ++B_) {          // two steps per trip: the stage in use is a literal
        if (st + B_ >= n_steps) break;
        const float* pa = rd_a + B_ * (WM * LW);
        const float* pb = rd_b + B_ * (WQ * LW);
        const float* na = rd_a + (1 - B_) * (WM * LW);      // the coming step's stage
        const float* nb = rd_b + (1 - B_) * (WQ * LW);
        float* wa_ = wr_a + (1 - B_) * (WM * LW);
        float* wb_ = wr_b + (1 - B_) * (WQ * LW);

        // chunk 0 (set X) while chunk 1 travels into set Y
        VDB_MM(xa0, xb0, 0, 0); VDB_RB(yb0, pb, 1, 0); VDB_FENCE();
        VDB_MM(xa0, xb1, 1, 0); VDB_RB(yb1, pb, 1, 1); VDB_FENCE();
        VDB_MM(xa1, xb0, 0, 1); VDB_RA(ya0, pa, 1, 0); VDB_FENCE();
        VDB_MM(xa1, xb1, 1, 1); VDB_RA(ya1, pa, 1, 1); VDB_FENCE();
        VDB_MM(xa2, xb0, 0, 2); VDB_RA(ya2, pa, 1, 2); VDB_FENCE();
        VDB_MM(xa2, xb1, 1, 2); VDB_RA(ya3, pa, 1, 3); VDB_FENCE();
        VDB_MM(xa3, xb0, 0, 3); VDB_FENCE();
        VDB_MM(xa3, xb1, 1, 3); VDB_FENCE();
#define VDB_WA(I) *reinterpret_cast<float4*>(wa_ + (I) * 64 * LW) = ra##I
#define VDB_WB(I) *reinterpret_cast<float4*>(wb_ + (I) * 64 * LW) = rb##I
#define VDB_GA(I) ra##I = *reinterpret_cast<const float4*>(abase + offa[I])
#define VDB_GB(I) rb##I = *reinterpret_cast<const float4*>(bbase + offb[I])
        // chunk 1 (set Y) while chunk 2 travels into set X
        VDB_MM(ya0, yb0, 0, 0); VDB_RB(xb0, pb, 2, 0); VDB_FENCE();
        VDB_MM(ya0, yb1, 1, 0); VDB_RB(xb1, pb, 2, 1); VDB_FENCE();
        VDB_MM(ya1, yb0, 0, 1); VDB_RA(xa0, pa, 2, 0); VDB_FENCE();
        VDB_MM(ya1, yb1, 1, 1); VDB_RA(xa1, pa, 2, 1); VDB_FENCE();
        VDB_MM(ya2, yb0, 0, 2); VDB_RA(xa2, pa, 2, 2); VDB_FENCE();
        VDB_MM(ya2, yb1, 1, 2); VDB_RA(xa3, pa, 2, 3); VDB_FENCE();
        VDB_MM(ya3, yb0, 0, 3); VDB_FENCE();
        VDB_MM(ya3, yb1, 1, 3); VDB_FENCE();
        // chunk 2 (set X) while chunk 3 travels into set Y; the coming step's rows and queries are
        // parked in the quarter before the barrier, one store per matrix instruction, so that their
        // loads (issued behind the previous barrier) have three quarters of a step to arrive
        VDB_MM(xa0, xb0, 0, 0); VDB_RB(yb0, pb, 3, 0); VDB_WA(0); VDB_FENCE();
        VDB_MM(xa0, xb1, 1, 0); VDB_RB(yb1, pb, 3, 1); VDB_WB(0); VDB_FENCE();
        VDB_MM(xa1, xb0, 0, 1); VDB_RA(ya0, pa, 3, 0); VDB_WA(1); VDB_FENCE();
        VDB_MM(xa1, xb1, 1, 1); VDB_RA(ya1, pa, 3, 1); VDB_WB(1); VDB_FENCE();
        VDB_MM(xa2, xb0, 0, 2); VDB_RA(ya2, pa, 3, 2); VDB_WA(2); VDB_FENCE();
        VDB_MM(xa2, xb1, 1, 2); VDB_RA(ya3, pa, 3, 3); VDB_WB(2); VDB_FENCE();
        VDB_MM(xa3, xb0, 0, 3); VDB_WA(3); VDB_FENCE();
        VDB_MM(xa3, xb1, 1, 3); VDB_WB(3); VDB_FENCE();
        // The step's only barrier: everyone has read this stage (chunk 3 is in registers) and
        // written the other one.  Chunk 3 is multiplied behind it while the step after next is
        // fetched and the first fragments of the coming step travel into set X.
        __syncthreads();
        {
            const char* abase = rows_w + (int64_t)min(lt, tile1 - 1) * WM * p.ld_bytes + lk * TKB;
            const char* bbase = qbase_ptr + lk * TKB;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) { asm("" : "+v"(offa[ps])); asm("" : "+v"(offb[ps])); }   // keep the lane offsets 32 bits wide
            VDB_MM(ya0, yb0, 0, 0); VDB_RB(xb0, nb, 0, 0); VDB_GA(0); VDB_FENCE();
            VDB_MM(ya0, yb1, 1, 0); VDB_RB(xb1, nb, 0, 1); VDB_GB(0); VDB_FENCE();
            VDB_MM(ya1, yb0, 0, 1); VDB_RA(xa0, na, 0, 0); VDB_GA(1); VDB_FENCE();
            VDB_MM(ya1, yb1, 1, 1); VDB_RA(xa1, na, 0, 1); VDB_GB(1); VDB_FENCE();
            VDB_MM(ya2, yb0, 0, 2); VDB_RA(xa2, na, 0, 2); VDB_GA(2); VDB_FENCE();
            VDB_MM(ya2, yb1, 1, 2); VDB_RA(xa3, na, 0, 3); VDB_GB(2); VDB_FENCE();
            VDB_MM(ya3, yb0, 0, 3); VDB_GA(3); VDB_FENCE();
            VDB_MM(ya3, yb1, 1, 3); VDB_GB(3); VDB_FENCE();
            if (++lk == ksteps) { lk = 0; ++lt; }
        }
        if (ck == ksteps - 1) {
            // row terms of the next tile travel while this tile's survivors are picked
            float nxt = 1.0f;
            const bool has_next = ct + 1 < tile1;
            if (METRIC != VDB_DOT && has_next && tid < WM) nxt = row_term[(int64_t)(ct + 1) * WM + tid];
            const float* rs = Rs + (ct & 1) * WM + wr * 128;
            const int64_t rbase = (int64_t)ct * WM + wr * 128;
            queue_epilogue<METRIC>(acc[0], rs, rbase, p.n, p.mask, half, bound[0], q_term[0], my_cnt[0], my_queue[0], p.cap);
            queue_epilogue<METRIC>(acc[1], rs, rbase, p.n, p.mask, half, bound[1], q_term[1], my_cnt[1], my_queue[1], p.cap);
            if (METRIC != VDB_DOT && has_next && tid < WM) {
                if (METRIC == VDB_COSINE && nxt < 1e-8f) nxt = 1.0f;
                Rs[((ct + 1) & 1) * WM + tid] = nxt;
            }
#pragma unroll
            for (int qd = 0; qd < 2; ++qd)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[qd][r][e] = 0.0f;
            ck = 0; ++ct;
        } else {
            ++ck;
        }
      }
    }
    __syncthreads();
#undef VDB_LOAD_STEP
#undef VDB_MM
#undef VDB_RA
#undef VDB_RB
#undef VDB_WA
#undef VDB_WB
#undef VDB_GA
#undef VDB_GB

    if (tid < WQ && q_base + tid < p.nq) p.counts[(q_base + tid) * p.n_slabs + slab] = Cnt[tid];
}

// The same tile and schedule with the staging done by the LDS-DMA path: a K step's 64 KB go from global memory
// straight into LDS (global_load_lds_dwordx4, 1 KB per wave instruction), so no staging registers, no LDS store
// instructions and no second trip of the data through the vector registers.  A transfer lands in lane order, so the
// staged rows are unpadded (128 bytes) and a row's eight 16-byte pieces are stored permuted instead: piece c of
// row r sits at position c ^ ((r >> 1) & 7), which keeps the 16-byte fragment reads of 32 consecutive rows free of
// bank conflicts.  The lane that fills position l & 7 of row r fetches piece (l & 7) ^ ((r >> 1) & 7).
static constexpr int LWD = 32;      // dwords per staged row in the DMA form
// a wave-uniform pointer, in scalar registers (the halves go through uint32_t: the builtin returns a signed int,
// and widening that directly would smear the low half's sign bit over the high half)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | (uint64_t)lo);
}
template <int METRIC>
__global__ __launch_bounds__(WNT, 1) void flat_queue_wide_dma_kernel(QueueParams p) {
    constexpr bool BF16 = true;
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                                   // [2][WM][LWD]  database rows, two stages
    float* Bs = As + 2 * WM * LWD;                       // [2][WQ][LWD]  queries, two stages
    float* Rs = Bs + 2 * WQ * LWD;                       // [2][WM]      per-row term, by tile parity
    int* Cnt = reinterpret_cast<int*>(Rs + 2 * WM);     // [WQ]         survivors per query of this block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int col = lane & 31;
    const int wr = wave >> 2;                           // which 128 rows of the tile
    const int wq = wave & 3;                            // which 64 queries of the tile

    // blocks that share a slab of rows (one per query tile) get the same blockIdx % 8: one XCD, one L2
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

    // a lane owns two queries: column `col` of each of its wave's two 32-query groups
    float q_term[2], bound[2];
    float2* my_queue[2];
    int* my_cnt[2];
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
        const int ql = wq * 64 + qd * 32 + col;
        const int my_q = qt * WQ + ql;
        const int my_q_ld = min(my_q, p.nq - 1);
        q_term[qd] = 0.0f;
        if (METRIC == VDB_COSINE) {
            q_term[qd] = p.q_nrm[my_q_ld];
            if (q_term[qd] < 1e-8f) q_term[qd] = INFINITY;       // metrics.py:472-473: all ones
        } else if (METRIC != VDB_DOT) {
            q_term[qd] = p.q_sq[my_q_ld];
        }
        bound[qd] = filter_bound<METRIC>(p.thr[my_q_ld], q_term[qd]);
        if (p.slack) bound[qd] = loosen_bound<METRIC>(bound[qd], p.slack[my_q_ld]);
        if (my_q >= p.nq) bound[qd] = (METRIC == VDB_DOT || METRIC == VDB_COSINE) ? INFINITY : -INFINITY;
        my_queue[qd] = p.queue + ((int64_t)my_q_ld * p.n_slabs + slab) * p.cap;
        my_cnt[qd] = &Cnt[ql];
    }
    if (tid < WQ) Cnt[tid] = 0;

    const int ksteps = p.ld_bytes / TKB;
    const int64_t q_base = (int64_t)qt * WQ;
    // staging: wave w brings rows [32 w, 32 w + 32) of the row tile and of the query tile, eight rows per transfer
    uint32_t offb[4], offa[4];          // byte offsets from a wave-uniform base
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = wave * 32 + ps * 8 + (lane >> 3);
        const uint32_t piece = (uint32_t)((lane & 7) ^ ((r >> 1) & 7)) * 16u;
        const int64_t gq = min(q_base + r, (int64_t)p.nq - 1) - q_base;
        offb[ps] = (uint32_t)(gq * p.ld_bytes) + piece;
        offa[ps] = (uint32_t)(r * p.ld_bytes) + piece;
    }
    // stage 0, this wave's rows: LDS byte addresses, wave-uniform (kept in scalar registers)
    const unsigned lds_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)As + (unsigned)(wave * 32 * LWD) * 4u);
    const unsigned lds_b = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Bs + (unsigned)(wave * 32 * LWD) * 4u);
    const char* rows_w = reinterpret_cast<const char*>(p.rows);
    const char* qbase_ptr = reinterpret_cast<const char*>(p.q) + q_base * p.ld_bytes;
    const float* row_term = (METRIC == VDB_COSINE) ? p.row_nrm : p.row_sq;
    const float* rd_a = As + (wr * 128 + col) * LWD;              // this lane's fragment rows (stage 0)
    const float* rd_b = Bs + (wq * 64 + col) * LWD;
    int xo[4];                                                    // where the lane's half of K chunk 0..3 sits in a staged row
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) xo[ch] = ((ch * 2 + half) ^ ((col >> 1) & 7)) * 4;

    f32x16 acc[2][4];                                             // [query group][32-row sub-tile]
#pragma unroll
    for (int qd = 0; qd < 2; ++qd)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[qd][r][e] = 0.0f;

    const int n_steps = (tile1 - tile0) * ksteps;
    int ck = 0, ct = tile0;             // the step being multiplied
    int lk = 0, lt = tile0;             // the step to fetch next
    // one K step into stage ST: eight 1 KB transfers per wave
#define VDB_DMA_BASES()                                                                          \
        const void* abase = uniform_ptr(rows_w + (int64_t)min(lt, tile1 - 1) * WM * p.ld_bytes + lk * TKB); \
        const void* bbase = uniform_ptr(qbase_ptr + lk * TKB);
#define VDB_DA(I, ST) glds16_s(abase, offa[I], lds_a + (unsigned)((ST) * WM * LWD + (I) * 8 * LWD) * 4u)
#define VDB_DB(I, ST) glds16_s(bbase, offb[I], lds_b + (unsigned)((ST) * WQ * LWD + (I) * 8 * LWD) * 4u)
#define VDB_LOAD_STEP(ST)                                                                        \
    do {                                                                                         \
        VDB_DMA_BASES()                                                                          \
        VDB_DA(0, ST); VDB_DB(0, ST); VDB_DA(1, ST); VDB_DB(1, ST);                              \
        VDB_DA(2, ST); VDB_DB(2, ST); VDB_DA(3, ST); VDB_DB(3, ST);                              \
        if (++lk == ksteps) { lk = 0; ++lt; }                                                    \
    } while (0)

    // A transfer is only ever issued for a step that will be multiplied: one that is still in flight when the
    // block ends would write into LDS that no longer belongs to it.
    if (n_steps > 0) {
        VDB_LOAD_STEP(0);
        if (n_steps > 1) VDB_LOAD_STEP(1);
        if (METRIC != VDB_DOT && tid < WM) {
            float t = row_term[(int64_t)tile0 * WM + tid];
            if (METRIC == VDB_COSINE && t < 1e-8f) t = 1.0f;        // metrics.py:476
            Rs[(tile0 & 1) * WM + tid] = t;
        }
        // the first step has landed; the second travels on
        if (n_steps > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // fragment sets: X holds the K chunk being multiplied, Y the next one
    float4 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;
    xb0 = *reinterpret_cast<const float4*>(rd_b + xo[0]); xb1 = *reinterpret_cast<const float4*>(rd_b + 32 * LWD + xo[0]);
    xa0 = *reinterpret_cast<const float4*>(rd_a + xo[0]); xa1 = *reinterpret_cast<const float4*>(rd_a + 32 * LWD + xo[0]);
    xa2 = *reinterpret_cast<const float4*>(rd_a + 64 * LWD + xo[0]); xa3 = *reinterpret_cast<const float4*>(rd_a + 96 * LWD + xo[0]);

#define VDB_MM(A, B, QD, R)                                                                                       \
    do {                                                                                                          \
        if (BF16) {                                                                                               \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A),                   \
                                                                 __builtin_bit_cast(bf16x8, B), acc[QD][R], 0, 0, 0); \
        } else {                                                                                                  \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, acc[QD][R], 0, 0, 0);                     \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, acc[QD][R], 0, 0, 0);                     \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, acc[QD][R], 0, 0, 0);                     \
            acc[QD][R] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, acc[QD][R], 0, 0, 0);                     \
        }                                                                                                         \
    } while (0)
    // fragment reads of K chunk CH from the stage at (PA, PB)
#define VDB_RA(DST, PA, CH, R) DST = *reinterpret_cast<const float4*>(PA + (R) * 32 * LWD + xo[CH])
#define VDB_RB(DST, PB, CH, QD) DST = *reinterpret_cast<const float4*>(PB + (QD) * 32 * LWD + xo[CH])

    for (int st = 0; st < n_steps; st += 2) {
#pragma unroll
      for (int B_ = 0; B_ < 2; ++B_) {          // two steps per trip: the stage in use is a literal
        if (st + B_ >= n_steps) break;
        const float* pa = rd_a + B_ * (WM * LWD);
        const float* pb = rd_b + B_ * (WQ * LWD);
        const float* na = rd_a + (1 - B_) * (WM * LWD);      // the coming step's stage
        const float* nb = rd_b + (1 - B_) * (WQ * LWD);

        // chunk 0 (set X) while chunk 1 travels into set Y
        VDB_MM(xa0, xb0, 0, 0); VDB_RB(yb0, pb, 1, 0); VDB_FENCE();
        VDB_MM(xa0, xb1, 1, 0); VDB_RB(yb1, pb, 1, 1); VDB_FENCE();
        VDB_MM(xa1, xb0, 0, 1); VDB_RA(ya0, pa, 1, 0); VDB_FENCE();
        VDB_MM(xa1, xb1, 1, 1); VDB_RA(ya1, pa, 1, 1); VDB_FENCE();
        VDB_MM(xa2, xb0, 0, 2); VDB_RA(ya2, pa, 1, 2); VDB_FENCE();
        VDB_MM(xa2, xb1, 1, 2); VDB_RA(ya3, pa, 1, 3); VDB_FENCE();
        VDB_MM(xa3, xb0, 0, 3); VDB_FENCE();
        VDB_MM(xa3, xb1, 1, 3); VDB_FENCE();
        // chunk 1 (set Y) while chunk 2 travels into set X
        VDB_MM(ya0, yb0, 0, 0); VDB_RB(xb0, pb, 2, 0); VDB_FENCE();
        VDB_MM(ya0, yb1, 1, 0); VDB_RB(xb1, pb, 2, 1); VDB_FENCE();
        VDB_MM(ya1, yb0, 0, 1); VDB_RA(xa0, pa, 2, 0); VDB_FENCE();
        VDB_MM(ya1, yb1, 1, 1); VDB_RA(xa1, pa, 2, 1); VDB_FENCE();
        VDB_MM(ya2, yb0, 0, 2); VDB_RA(xa2, pa, 2, 2); VDB_FENCE();
        VDB_MM(ya2, yb1, 1, 2); VDB_RA(xa3, pa, 2, 3); VDB_FENCE();
        VDB_MM(ya3, yb0, 0, 3); VDB_FENCE();
        VDB_MM(ya3, yb1, 1, 3); VDB_FENCE();
        // chunk 2 (set X) while chunk 3 travels into set Y
        VDB_MM(xa0, xb0, 0, 0); VDB_RB(yb0, pb, 3, 0); VDB_FENCE();
        VDB_MM(xa0, xb1, 1, 0); VDB_RB(yb1, pb, 3, 1); VDB_FENCE();
        VDB_MM(xa1, xb0, 0, 1); VDB_RA(ya0, pa, 3, 0); VDB_FENCE();
        VDB_MM(xa1, xb1, 1, 1); VDB_RA(ya1, pa, 3, 1); VDB_FENCE();
        VDB_MM(xa2, xb0, 0, 2); VDB_RA(ya2, pa, 3, 2); VDB_FENCE();
        VDB_MM(xa2, xb1, 1, 2); VDB_RA(ya3, pa, 3, 3); VDB_FENCE();
        VDB_MM(xa3, xb0, 0, 3); VDB_FENCE();
        VDB_MM(xa3, xb1, 1, 3); VDB_FENCE();
        // The step's only barrier: everyone has read this stage (chunk 3 is in registers) and the coming step's
        // transfers, issued behind the previous barrier, have landed in the other one.  Chunk 3 is multiplied
        // behind it while the step after next starts travelling into the stage just read, and the first
        // fragments of the coming step go into set X.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
            const bool fetch = st + B_ + 2 < n_steps;       // the last two steps fetch nothing (uniform)
            VDB_DMA_BASES()
            VDB_MM(ya0, yb0, 0, 0); VDB_RB(xb0, nb, 0, 0); if (fetch) VDB_DA(0, B_); VDB_FENCE();
            VDB_MM(ya0, yb1, 1, 0); VDB_RB(xb1, nb, 0, 1); if (fetch) VDB_DB(0, B_); VDB_FENCE();
            VDB_MM(ya1, yb0, 0, 1); VDB_RA(xa0, na, 0, 0); if (fetch) VDB_DA(1, B_); VDB_FENCE();
            VDB_MM(ya1, yb1, 1, 1); VDB_RA(xa1, na, 0, 1); if (fetch) VDB_DB(1, B_); VDB_FENCE();
            VDB_MM(ya2, yb0, 0, 2); VDB_RA(xa2, na, 0, 2); if (fetch) VDB_DA(2, B_); VDB_FENCE();
            VDB_MM(ya2, yb1, 1, 2); VDB_RA(xa3, na, 0, 3); if (fetch) VDB_DB(2, B_); VDB_FENCE();
            VDB_MM(ya3, yb0, 0, 3); if (fetch) VDB_DA(3, B_); VDB_FENCE();
            VDB_MM(ya3, yb1, 1, 3); if (fetch) VDB_DB(3, B_); VDB_FENCE();
            if (fetch && ++lk == ksteps) { lk = 0; ++lt; }
        }
        if (ck == ksteps - 1) {
            // row terms of the next tile travel while this tile's survivors are picked
            float nxt = 1.0f;
            const bool has_next = ct + 1 < tile1;
            if (METRIC != VDB_DOT && has_next && tid < WM) nxt = row_term[(int64_t)(ct + 1) * WM + tid];
            const float* rs = Rs + (ct & 1) * WM + wr * 128;
            const int64_t rbase = (int64_t)ct * WM + wr * 128;
            queue_epilogue<METRIC>(acc[0], rs, rbase, p.n, p.mask, half, bound[0], q_term[0], my_cnt[0], my_queue[0], p.cap);
            queue_epilogue<METRIC>(acc[1], rs, rbase, p.n, p.mask, half, bound[1], q_term[1], my_cnt[1], my_queue[1], p.cap);
            if (METRIC != VDB_DOT && has_next && tid < WM) {
                if (METRIC == VDB_COSINE && nxt < 1e-8f) nxt = 1.0f;
                Rs[((ct + 1) & 1) * WM + tid] = nxt;
            }
#pragma unroll
            for (int qd = 0; qd < 2; ++qd)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[qd][r][e] = 0.0f;
            ck = 0; ++ct;
        } else {
            ++ck;
        }
      }
    }
    __syncthreads();
#undef VDB_LOAD_STEP
#undef VDB_MM
#undef VDB_RA
#undef VDB_RB
#undef VDB_DA
#undef VDB_DB
#undef VDB_DMA_BASES

    if (tid < WQ && q_base + tid < p.nq) p.counts[(q_base + tid) * p.n_slabs + slab] = Cnt[tid];
}

// --------------------------------------------------------------------------
bool queue_use_wide(const Rows& r) {
    static const bool off = getenv("VDB_NO_QWIDE") != nullptr;
    const int64_t tiles = (r.n + WM - 1) / WM;
    return !off && r.bf16 && (r.ld16 * 2) % TKB == 0 && r.cap >= tiles * WM;
}

void plan_grid_wide(int n_tiles, int nq, int num_cus, int* n_qtiles, int* n_slabs, int* tiles_per_slab) {
    *n_qtiles = (nq + WQ - 1) / WQ;
    int want = (num_cus + *n_qtiles - 1) / *n_qtiles;       // one block per CU in total
    if (want < 1) want = 1;
    if (want > n_tiles) want = n_tiles;
    *tiles_per_slab = (n_tiles + want - 1) / want;
    *n_slabs = (n_tiles + *tiles_per_slab - 1) / *tiles_per_slab;
}

int launch_queue_wide(int metric, QueueParams& p, hipStream_t s) {
    const int groups = (p.n_slabs + 7) / 8;
    const dim3 grid(p.n_slabs >= 8 ? groups * 8 * p.n_qtiles : p.n_slabs * p.n_qtiles);
    // Staging through registers is the default.  The LDS-DMA form (VDB_WIDE_DMA=1) measures 1.68 ms against
    // 1.71 ms on dot 1.25M x 768 and 0.90 ms against 0.81 ms on cosine 1M x 384 (batch 1024): the kernel waits for
    // the bytes of the coming step (issued three quarters of a step ahead, two stages of 64 KB are all the LDS
    // holds), not for the instructions that move them.
    static const bool no_dma = getenv("VDB_WIDE_DMA") == nullptr;
    const size_t lds = (size_t)(2 * WM * (no_dma ? LW : LWD) + 2 * WQ * (no_dma ? LW : LWD) + 2 * WM + WQ) * sizeof(float);
    KernelTimer timer(s, true);
#define VDB_LAUNCH_WIDE(M)                                                                     \
    case M: {                                                                                  \
        auto kern = no_dma ? flat_queue_wide_kernel<M, true> : flat_queue_wide_dma_kernel<M>;  \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern),                       \
                                    lds));    \
        hipLaunchKernelGGL(kern, grid, dim3(WNT), lds, s, p);                                  \
        break;                                                                                 \
    }
    switch (metric) {
        VDB_LAUNCH_WIDE(VDB_EUCLIDEAN)
        VDB_LAUNCH_WIDE(VDB_EUCLIDEAN_SQ)
        VDB_LAUNCH_WIDE(VDB_COSINE)
        VDB_LAUNCH_WIDE(VDB_DOT)
        default:
            set_error("metric %d has no matrix-core path", metric);
            return VDB_ERR_INVALID;
    }
#undef VDB_LAUNCH_WIDE
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
