// Streaming distance pipeline for the bandwidth-bound searches:
//
//   stream_sums_kernel : per (row, query) the reduction  sum_j term(row_j, q_j)
//                        (or max_j) in NumPy's pairwise order, for row chunks of
//                        256 rows against groups of up to G queries.  Rows are
//                        read from HBM once per chunk with asynchronous
//                        global->LDS loads, several stages ahead of the
//                        arithmetic; the values go to a candidate buffer.
//   topk_cand_kernel   : per query, the distance finish (norm terms, clamp,
//                        sqrt), the filter bitmask, and the k best of that
//                        query's slice of the candidate buffer.
//   finish_dist_kernel : the distance finish alone (full distance rows).
//
// Served by this pipeline:
//   - Manhattan / Chebyshev, which have no GEMM form (distance/metrics.py:486-490);
//   - single queries and small batches of any metric (FlatIndex.search,
//     index/flat.py:331; Collection.search, core/collection.py:750);
//   - the per-pair distances of the IVF search: centroid ranking
//     (index/ivf.py:611-614) and the scan of the probed lists (:619-628).
//     For a batch the probed lists are inverted (list -> queries that probe
//     it), so a list chunk is read once and serves up to G queries.
//
// Work is cut into items = (row chunk of a segment, query group of that
// segment).  A flat search is one segment holding every row and every query;
// an IVF scan has one segment per inverted list.  Blocks take items
// round-robin; every block walks its items' 32-float stages as one sequence so
// loads for the next item are in flight while the current one is summed.
#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

static constexpr int S_NT = 256;                  // threads = rows per chunk
static constexpr int S_CK = 32;                   // floats of every row per stage: one 128-byte line
static constexpr int S_STAGE = S_NT * S_CK;       // floats per stage buffer (32 KiB)
static constexpr int STREAM_QS = 2;               // wave sets sharing one row chunk in the 8-query kernels

// uniform (scalar-cache) load: the address must be the same in every lane
template <class T> __device__ __forceinline__ T uload(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

struct StreamParams {
    const float* data; int ld; int dim;
    int n_leaves; int leaf_len[SumPlan::MAX_LEAVES]; int folds[SumPlan::MAX_LEAVES];   // SumPlan, dword fields (scalar loads)
    int post;                         // StreamPost applied to the sum before it is stored
    int nbuf;                         // stage buffers in LDS
    const float* q; int ldq;
    const float* zeros;               // >= 16 zero bytes, source of the columns past ld
    int n_seg;
    const int64_t* seg_row_off;       // [n_seg+1] stored row range of a segment
    const int* seg_q_off;             // [n_seg+1] range of the segment's queries in sq_*
    const int64_t* seg_item_off;      // [n_seg+1] first item of a segment
    const StreamItem* items;          // [seg_item_off[n_seg]] the decoded items
    int64_t items_cap;
    int64_t reverse_from;             // >= 0: the items are taken back to front, item i is table entry reverse_from - i
    const int* sq_qid;                // query number
    const int64_t* sq_out;            // where that query's value for the segment's row 0 goes
    float* out;
    const int* run_if;                // optional device flag: the launch does nothing while it is zero
    // Bounded form (row-resident kernel, flat row sets): instead of storing every sum, a (row, query) pair whose
    // value is within the query's bound and whose row passes the filter is appended to the query's queue.
    const float* bound;               // [queries] a distance that is not below the query's k-th best, or null
    int bound_sqrt;                   // the sums are squares of the distances the bound is given in
    const uint32_t* mask;             // filter bits by row number, or null
    int* queue_len; float* queue_d; int* queue_i; int queue_cap;   // per query: entries so far, (distance, row) pairs
    int* overflow;                    // set when a queue is full: the caller repeats the search without the bound
};

template <int OP> __device__ __forceinline__ float s_term(float x, float q) {
    if (OP == OP_SQDIFF) { const float d = q - x; return d * d; }
    if (OP == OP_ABSDIFF || OP == OP_MAXABS) return fabsf(x - q);
    if (OP == OP_SQ) return x * x;
    if (OP == OP_NEQ) return x != q ? 1.0f : 0.0f;      // counts stay exact in fp32 up to 2^24 positions
    return x * q;
}
template <int OP> __device__ __forceinline__ float s_comb(float a, float b) {
    return OP == OP_MAXABS ? fmaxf(a, b) : a + b;
}
template <int OP> __device__ __forceinline__ float s_accum(float r, float x, float q) {
    if (OP == OP_DOT) return __fmaf_rn(x, q, r);
    return s_comb<OP>(r, s_term<OP>(x, q));
}
template <int OP> __device__ __forceinline__ float s_tree(const float* r) {
    return s_comb<OP>(s_comb<OP>(s_comb<OP>(r[0], r[1]), s_comb<OP>(r[2], r[3])),
                      s_comb<OP>(s_comb<OP>(r[4], r[5]), s_comb<OP>(r[6], r[7])));
}

typedef float f2 __attribute__((ext_vector_type(2)));     // two accumulators: the packed fp32 VALU forms
template <int OP> __device__ __forceinline__ f2 s_accum2(f2 r, f2 x, f2 q) {
    if (OP == OP_DOT) return __builtin_elementwise_fma(x, q, r);
    if (OP == OP_SQDIFF) { const f2 d = q - x; return r + d * d; }
    if (OP == OP_SQ) return r + x * x;
    if (OP == OP_NEQ) return r + f2{x.x != q.x ? 1.0f : 0.0f, x.y != q.y ? 1.0f : 0.0f};
    const f2 d = __builtin_elementwise_abs(x - q);
    if (OP == OP_MAXABS) return f2{fmaxf(r.x, d.x), fmaxf(r.y, d.y)};
    return r + d;
}
// The staged form of s_accum2 for the operations whose pair costs more than one instruction: stage 1 on every
// pair of a unit, then (squared differences) the squares, then stage 3.  |d| enters the last stage as a source
// modifier of a one-element add or max; there is no packed fp32 abs or max.
template <int OP> __device__ constexpr bool s_staged() { return OP == OP_SQDIFF || OP == OP_ABSDIFF || OP == OP_MAXABS || OP == OP_SQ; }
template <int OP> __device__ __forceinline__ f2 s_stage1(f2 x, f2 q) {
    if (OP == OP_SQDIFF) return q - x;
    if (OP == OP_SQ) return x * x;
    return x - q;
}
template <int OP> __device__ __forceinline__ f2 s_stage3(f2 r, f2 d) {
    if (OP == OP_ABSDIFF) return f2{r.x + __builtin_fabsf(d.x), r.y + __builtin_fabsf(d.y)};
    if (OP == OP_MAXABS) return f2{fmaxf(r.x, __builtin_fabsf(d.x)), fmaxf(r.y, __builtin_fabsf(d.y))};
    return r + d;
}
// every value of d[] is computed before anything behind the fence starts
template <int N> __device__ __forceinline__ void s_fence(f2* d) {
    static_assert(N % 4 == 0, "pairs come in fours");
    if constexpr (N % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N; i += 8)
            asm volatile("" : "+v"(d[i]), "+v"(d[i + 1]), "+v"(d[i + 2]), "+v"(d[i + 3]), "+v"(d[i + 4]), "+v"(d[i + 5]), "+v"(d[i + 6]), "+v"(d[i + 7]));
    } else {
#pragma unroll
        for (int i = 0; i < N; i += 4) asm volatile("" : "+v"(d[i]), "+v"(d[i + 1]), "+v"(d[i + 2]), "+v"(d[i + 3]));
    }
}
template <int OP> __device__ __forceinline__ float s_tree2(const f2* a) {
    return s_comb<OP>(s_comb<OP>(s_comb<OP>(a[0].x, a[0].y), s_comb<OP>(a[1].x, a[1].y)),
                      s_comb<OP>(s_comb<OP>(a[2].x, a[2].y), s_comb<OP>(a[3].x, a[3].y)));
}

typedef int i8v __attribute__((ext_vector_type(8)));
// eight consecutive floats at any dword address through the scalar cache (one s_load_dwordx8)
__device__ __forceinline__ i8v uload8(const float* p) {
    typedef int i8v_u __attribute__((ext_vector_type(8), aligned(4)));
    return *reinterpret_cast<const __attribute__((address_space(4))) i8v_u*>(reinterpret_cast<uintptr_t>(p));
}
struct Item {
    int64_t row0;        // first stored row of the chunk
    int64_t out_rel;     // chunk's first row relative to its segment's first row
    int rows;            // rows in the chunk (1..256)
    int qbeg;            // first query entry of the group
    int qcnt;            // queries in the group (1..G)
};
static_assert(sizeof(StreamItem) == 32, "one item is one 32-byte scalar load");

__device__ __forceinline__ Item load_item(const StreamParams& p, int64_t item) {
    if (p.reverse_from >= 0) item = p.reverse_from - item;
    const i8v w = uload(reinterpret_cast<const i8v*>(p.items + item));
    Item it;
    it.row0 = ((int64_t)(uint32_t)w[1] << 32) | (uint32_t)w[0];
    it.out_rel = ((int64_t)(uint32_t)w[3] << 32) | (uint32_t)w[2];
    it.rows = w[4]; it.qbeg = w[5]; it.qcnt = w[6];
    return it;
}

template <int PER_STAGE> __device__ __forceinline__ void wait_loads(int stages_behind) {
    // the wave's LDS-DMA loads retire in order: leaving `stages_behind` later stages
    // outstanding means the stage in front of them has landed
    static_assert(PER_STAGE == 9 || PER_STAGE == 8 || PER_STAGE == 5 || PER_STAGE == 4 || PER_STAGE == 3,
                  "add the immediates for this count");
    if (stages_behind >= 2) {
        if (PER_STAGE == 9) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (PER_STAGE == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (PER_STAGE == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (PER_STAGE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (stages_behind == 1) {
        if (PER_STAGE == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (PER_STAGE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (PER_STAGE == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (PER_STAGE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// G queries per item, shared out over QS sets of waves: a thread sums R rows of the chunk (rows
// row0, row0 + 256/R, ...) against the G/QS queries of set (tid / (256/R)).  More than one set keeps
// two or more waves on every SIMD, so one wave's scalar work and waits hide behind another's
// arithmetic.  R = 2 halves what a (row, query) pair costs besides its arithmetic: a query value
// read from LDS (a broadcast to the whole wave) serves two rows, and the per-item bookkeeping,
// barriers and result stores are shared by twice as many pairs.
// WHOLE: every leaf of the plan is a whole number of stages (dimensions such as 128, 256, 384,
// 768, 1024): the group-by-group path for leaf ends inside a stage is compiled out, and with it
// the register copies where the two paths meet.
template <int OP, int G, int QS, bool WHOLE, int R>
__global__ __launch_bounds__(S_NT / R * QS, (G == 8 && R == 1 && WHOLE) ? 4 : 1) void stream_sums_kernel(const StreamParams p) {
    extern __shared__ __align__(16) float smem[];
    constexpr int TPS = S_NT / R;                 // threads per wave set
    static_assert(TPS % 64 == 0, "a wave set is whole waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tid % TPS;
    const int nbuf = p.nbuf;
    constexpr int GT = G / QS;                    // queries per thread
#ifndef VDB_STREAM_H1
#define VDB_STREAM_H1 4
#endif
    // queries per unit of the arithmetic loop (a unit's values sit in registers twice: in use and prefetched)
    constexpr int H = R > 1 ? 2 : ((G == 8 && WHOLE) ? 1 : (GT >= VDB_STREAM_H1 ? VDB_STREAM_H1 : GT));
    constexpr int U = GT / H;
    constexpr int NW = TPS * QS / 64;             // waves
    constexpr int ROW_GLDS = 32 / NW;             // row LDS-DMA instructions per wave and stage
    // SQ: the query values do not pass through LDS at all.  They are the same for every lane, so they can come
    // through the scalar cache into SGPRs and enter the packed operations as scalar operands (two s_load_dwordx8 per
    // unit, issued one unit ahead).  Measured on the 10M x 128 list scan and the 1M x 128 Manhattan batch this is no
    // faster than the LDS broadcast reads (2.61 vs 2.61 ms, 12.6 vs 12.0 ms): eight waves per CU ask the scalar
    // cache for as many bytes as they would ask LDS for.  Left off.
    constexpr bool SQ = false;
    constexpr int PER_STAGE = ROW_GLDS + (SQ ? 0 : 1);
    constexpr int SLOT = S_STAGE + G * S_CK;      // floats per stage slot: 256 row pieces, then the G query pieces
    const int qset = __builtin_amdgcn_readfirstlane(tid / TPS);
    const int g_first = qset * GT;                // this thread's slots of the pending-sum stack
    // The group's queries are dealt to the wave sets unit by unit (H at a time), so a group that
    // is only partly filled keeps every set busy: local query g is group query gq(g).
    auto gq = [&](int g) { return ((g / H) * QS + qset) * H + (g % H); };
    float* St = smem + nbuf * SLOT;               // [depth][G][256] pending partial sums (more than one leaf only)

    if (p.run_if && uload(p.run_if) == 0) return;
    const int ksteps = (p.ld + S_CK - 1) / S_CK;
    const int64_t total_items = min(uload(p.seg_item_off + p.n_seg), p.items_cap);
    // Blocks b, b + 8, b + 16, ... share an XCD and its L2.  Consecutive items are the query groups of one row chunk
    // (then the next chunk of the list), so a run of gridDim/8 consecutive items goes to the blocks of one XCD: the
    // chunk leaves HBM once for all of its groups instead of once per XCD that happens to hold one of them.
    const int per_xcd = gridDim.x >> 3;
    const int64_t first = (gridDim.x & 7) == 0 ? (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    if (first >= total_items) return;
    const int my_items = (int)((total_items - first + gridDim.x - 1) / gridDim.x);
    const int n_stages = my_items * ksteps;
    const bool one_leaf = p.n_leaves == 1;

    // ---- producer: stage -> LDS.  Row r of the chunk sits at r*32 floats; its
    // 16-byte piece c is stored at piece position c ^ ((r >> 1) & 7), which makes
    // the one-row-per-thread 16-byte reads below conflict free.
    int pn = 0, p_ks = 0, p_buf = 0;
    int64_t p_item = first;
    const int piece = lane & 7;
    // query pieces: the stage's 32 floats of each of the group's queries are G*8 pieces of
    // 16 bytes; wave w fetches pieces [w*QL, (w+1)*QL) (piece s: query s >> 3, floats (s & 7)*4..)
    constexpr int QL = (G * 8) / NW;              // lanes per wave that fetch query pieces: 2, 4, 8, 16 or 32
    static_assert(QL >= 1 && QL <= 32 && (QL & (QL - 1)) == 0, "query pieces must split evenly over the waves");
    constexpr int QPW = QL > 8 ? QL / 8 : 1;      // queries a wave's pieces belong to
    const int sidx = wave * QL + (lane & (QL - 1));
    const int ga = (wave * QL) >> 3;              // the first of them
    const float* row_src[ROW_GLDS];               // this lane's row pieces of the item, at column 0
    int row_c4[ROW_GLDS];                         // 4 * (piece of the row this lane fetches)
    // UBASE (the batch kernels, which only read row stores): a row piece is addressed as a
    // wave-uniform base (first row of the chunk, current column) plus a byte offset that is fixed per lane for
    // the whole launch, so a stage issues its loads without any vector-ALU work.  A chunk that ends before its
    // 256th row reads on into the rows behind it (a row store keeps that many rows of slack behind its last
    // row); those sums are never stored.
    constexpr bool UBASE = WHOLE && G >= 8;
    unsigned row_off[ROW_GLDS];
    const float* item_base = p.data;
#pragma unroll
    for (int i = 0; i < ROW_GLDS; ++i) {
        const int r = wave * (8 * ROW_GLDS) + i * 8 + (lane >> 3);
        row_off[i] = (unsigned)((r * p.ld + (piece ^ ((r >> 1) & 7)) * 4) * (int)sizeof(float));
    }
    const float* q_src = p.q;
    auto begin_item = [&]() {
        const Item it = load_item(p, p_item);
        if constexpr (UBASE) {
            item_base = p.data + it.row0 * p.ld;      // wave-uniform: the item comes from scalar loads
        } else {
#pragma unroll
            for (int i = 0; i < ROW_GLDS; ++i) {
                const int r = wave * (8 * ROW_GLDS) + i * 8 + (lane >> 3);
                row_src[i] = p.data + (it.row0 + min(r, it.rows - 1)) * p.ld;
                row_c4[i] = (piece ^ ((r >> 1) & 7)) * 4;
            }
        }
        if constexpr (!SQ) {
            int id = uload(p.sq_qid + it.qbeg + min(ga, it.qcnt - 1));
#pragma unroll
            for (int j = 1; j < QPW; ++j) {
                const int idj = uload(p.sq_qid + it.qbeg + min(ga + j, it.qcnt - 1));
                if ((sidx >> 3) - ga == j) id = idj;
            }
            q_src = p.q + (int64_t)id * p.ldq + (sidx & 7) * 4;
        }
    };
    begin_item();
    auto issue = [&]() {
        const unsigned slot = (unsigned)(uintptr_t)(smem + p_buf * SLOT);
        const unsigned base = __builtin_amdgcn_readfirstlane(slot + wave * (8 * ROW_GLDS) * S_CK * (unsigned)sizeof(float));
        const int col0 = p_ks * S_CK;
        if (!SQ && lane < QL) {
            const float* src = (col0 + (sidx & 7) * 4 < p.ld) ? q_src + col0 : p.zeros;
            glds16(src, __builtin_amdgcn_readfirstlane(slot + (S_STAGE + wave * QL * 4) * (unsigned)sizeof(float)));
        }
        if constexpr (UBASE) {
            const uint64_t b = (uint64_t)(uintptr_t)(item_base + col0);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
            const void* sb = reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
#pragma unroll
            for (int i = 0; i < ROW_GLDS; ++i) glds16_s(sb, row_off[i], base + i * 8 * S_CK * (unsigned)sizeof(float));
        } else {
#pragma unroll
            for (int i = 0; i < ROW_GLDS; ++i) {
                const float* src = (col0 + row_c4[i] < p.ld) ? row_src[i] + col0 + row_c4[i] : p.zeros;
                glds16(src, base + i * 8 * S_CK * (unsigned)sizeof(float));
            }
        }
        ++pn;
        if (++p_buf == nbuf) p_buf = 0;
        if (++p_ks == ksteps) {
            p_ks = 0;
            p_item += gridDim.x;
            if (pn < n_stages) begin_item();
        }
    };
    for (int i = 0; i < nbuf - 1; ++i)
        if (pn < n_stages) issue();

    // ---- consumer.  Control flow (leaf boundaries, tails) depends on the dimension
    // only, so it is uniform; inside a branch the queries are straight-line code.
    // Query slots past the group's count repeat its last query and are not stored.
    int64_t c_item = first;
    int c_ks = 0, c_buf = 0;
    Item c_it{};
    f2 acc[R][GT][4];
    float res[R][GT];
    const float* qp[GT];                          // SQ: this thread's queries of the item (wave-uniform addresses)
    int leaf = 0, in_leaf = 0, sp = 0, L = 0;
    const int sw = (row0 >> 1) & 7;               // the same for every row of the thread: TPS is a multiple of 16

    auto zero_acc = [&]() {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int g = 0; g < GT; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[rr][g][j] = f2{0.0f, 0.0f};
            }
        }
    };
    // a leaf is complete: res[] holds its sums
    auto end_leaf = [&]() {
        if (!one_leaf) {
            // push, then fold pending partial sums right into left
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int row = row0 + rr * TPS;
#pragma unroll
                for (int g = 0; g < GT; ++g) St[(sp * G + g_first + g) * S_NT + row] = res[rr][g];
            }
            ++sp;
            for (int f = 0; f < p.folds[leaf]; ++f) {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int row = row0 + rr * TPS;
#pragma unroll
                    for (int g = 0; g < GT; ++g) {
                        const float b = St[((sp - 1) * G + g_first + g) * S_NT + row];
                        const float a = St[((sp - 2) * G + g_first + g) * S_NT + row];
                        St[((sp - 2) * G + g_first + g) * S_NT + row] = s_comb<OP>(a, b);
                    }
                }
                --sp;
            }
        }
        ++leaf;
        in_leaf = 0;
        L = (leaf < p.n_leaves) ? p.leaf_len[leaf] : 0;
        zero_acc();
    };

    for (int n = 0; n < n_stages; ++n) {
        wait_loads<PER_STAGE>(pn - n - 1);
        __builtin_amdgcn_s_barrier();
        if (pn < n_stages) issue();

        if (c_ks == 0) {
            c_it = load_item(p, c_item);
            zero_acc();
            leaf = 0; in_leaf = 0; sp = 0; L = p.leaf_len[0];
            if constexpr (SQ) {
#pragma unroll
                for (int g = 0; g < GT; ++g)
                    qp[g] = p.q + (int64_t)uload(p.sq_qid + c_it.qbeg + min(gq(g), c_it.qcnt - 1)) * p.ldq;
            }
        }

        const float* xb = smem + c_buf * SLOT + row0 * S_CK;
        const float* qb = smem + c_buf * SLOT + S_STAGE;
        if (++c_buf == nbuf) c_buf = 0;
        float4 xv[R][8];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int c = 0; c < 8; ++c) xv[rr][c] = *reinterpret_cast<const float4*>(xb + rr * TPS * S_CK + ((c ^ sw) << 2));
        }

        // the stage's query values, fetched one unit (8 floats of H queries) ahead of their use
        const int col0 = c_ks * S_CK;
        float4 qv[2][H][2];
        i8v qs[2][H];
        auto load_q = [&](int buf, int unit) {
            if constexpr (SQ) {
#pragma unroll
                for (int h = 0; h < H; ++h)
                    qs[buf][h] = uload8(qp[(unit % U) * H + h] + col0 + (unit / U) * 8);
            } else {
                const float* src = qb + gq((unit % U) * H) * S_CK + (unit / U) * 8;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    qv[buf][h][0] = *reinterpret_cast<const float4*>(src + h * S_CK);
                    qv[buf][h][1] = *reinterpret_cast<const float4*>(src + h * S_CK + 4);
                }
            }
        };
        // a unit whose H queries are all padding (a group that is not full) is neither fetched nor summed
        auto unit_live = [&](int unit) { return gq((unit % U) * H) < c_it.qcnt; };
        if (SQ || unit_live(0)) load_q(0, 0);

        if (WHOLE || L - in_leaf >= S_CK) {
            // the whole stage lies inside one leaf: four full groups of 8, no control flow
#pragma unroll
            for (int unit = 0; unit < (S_CK / 8) * U; ++unit) {
                const int grp = unit / U, g0 = (unit % U) * H;
                if constexpr (SQ) {
                    // Scalar loads return in any order, so the only wait there is for them is "all of them".  Taking
                    // that wait here, before the next unit's loads go out, makes it a wait for loads that were issued
                    // a whole unit of arithmetic ago; taken at the first use below, it would wait for the loads
                    // issued just above it as well.
#pragma unroll
                    for (int h = 0; h < H; ++h) asm volatile("" :: "s"(qs[unit & 1][h]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (scalar loads are issued for padded units too, from the group's last query: branches around them
                // would only blur which loads are still in flight where the paths meet)
                if (unit + 1 < (S_CK / 8) * U && (SQ || unit_live(unit + 1))) load_q((unit + 1) & 1, unit + 1);
                if constexpr (SQ) __builtin_amdgcn_sched_barrier(0);
                if (gq(g0) < c_it.qcnt) {               // units whose queries are all padding do no arithmetic
                    f2 q2[H][4], x2[R][4];
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        if constexpr (SQ) {
                            const i8v w = qs[unit & 1][h];
#pragma unroll
                            for (int j = 0; j < 4; ++j) q2[h][j] = f2{__int_as_float(w[2 * j]), __int_as_float(w[2 * j + 1])};
                        } else {
                            const float4 qa = qv[unit & 1][h][0], qc = qv[unit & 1][h][1];
                            q2[h][0] = f2{qa.x, qa.y}; q2[h][1] = f2{qa.z, qa.w}; q2[h][2] = f2{qc.x, qc.y}; q2[h][3] = f2{qc.z, qc.w};
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < R; ++rr) {
                        x2[rr][0] = f2{xv[rr][2 * grp].x, xv[rr][2 * grp].y}; x2[rr][1] = f2{xv[rr][2 * grp].z, xv[rr][2 * grp].w};
                        x2[rr][2] = f2{xv[rr][2 * grp + 1].x, xv[rr][2 * grp + 1].y};
                        x2[rr][3] = f2{xv[rr][2 * grp + 1].z, xv[rr][2 * grp + 1].w};
                    }
                    if constexpr (s_staged<OP>()) {
                        // The unit's H*R*4 packed pairs go through the operation stage by stage (all differences, then
                        // all squares, then all additions): an instruction that consumes the result of the one before
                        // it waits out the pipeline, and a wave's neighbours are often in their LDS reads instead of
                        // filling that gap.  Same operations on the same values: the sums are unchanged.
                        constexpr int NE = H * R * 4;
                        f2 d[NE];
#pragma unroll
                        for (int h = 0; h < H; ++h)
#pragma unroll
                            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                                for (int j = 0; j < 4; ++j) d[(h * R + rr) * 4 + j] = s_stage1<OP>(x2[rr][j], q2[h][j]);
                        s_fence<NE>(d);
                        if constexpr (OP == OP_SQDIFF) {
#pragma unroll
                            for (int e = 0; e < NE; ++e) d[e] = d[e] * d[e];
                            s_fence<NE>(d);
                        }
#pragma unroll
                        for (int h = 0; h < H; ++h)
#pragma unroll
                            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[rr][g0 + h][j] = s_stage3<OP>(acc[rr][g0 + h][j], d[(h * R + rr) * 4 + j]);
                    } else {
#pragma unroll
                        for (int h = 0; h < H; ++h)
#pragma unroll
                            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[rr][g0 + h][j] = s_accum2<OP>(acc[rr][g0 + h][j], x2[rr][j], q2[h][j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the query reads one unit ahead, not all at the top
            }
            in_leaf += S_CK;
            if (in_leaf == L) {
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
#pragma unroll
                    for (int g = 0; g < GT; ++g) res[rr][g] = s_tree2<OP>(acc[rr][g]);
                }
                end_leaf();
            }
        } else if constexpr (!WHOLE) {
            // a leaf ends inside the stage, or the dimension does: group by group (rare, kept simple)
#pragma unroll
            for (int grp = 0; grp < S_CK / 8; ++grp) {
                if (col0 + grp * 8 < p.dim) {
                    const int cnt = min(8, L - in_leaf);
#pragma unroll
                    for (int g = 0; g < GT; ++g) {
                        const float4 qa = *reinterpret_cast<const float4*>(qb + gq(g) * S_CK + grp * 8);
                        const float4 qc = *reinterpret_cast<const float4*>(qb + gq(g) * S_CK + grp * 8 + 4);
#pragma unroll
                        for (int rr = 0; rr < R; ++rr) {
                            const f2 x2[4] = {f2{xv[rr][2 * grp].x, xv[rr][2 * grp].y}, f2{xv[rr][2 * grp].z, xv[rr][2 * grp].w},
                                              f2{xv[rr][2 * grp + 1].x, xv[rr][2 * grp + 1].y},
                                              f2{xv[rr][2 * grp + 1].z, xv[rr][2 * grp + 1].w}};
                            if (cnt == 8) {
                                const f2 q2[4] = {f2{qa.x, qa.y}, f2{qa.z, qa.w}, f2{qc.x, qc.y}, f2{qc.z, qc.w}};
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[rr][g][j] = s_accum2<OP>(acc[rr][g][j], x2[j], q2[j]);
                                res[rr][g] = s_tree2<OP>(acc[rr][g]);
                            } else {
                                // sequential tail of the leaf (or a leaf shorter than 8)
                                const float xs[8] = {x2[0].x, x2[0].y, x2[1].x, x2[1].y, x2[2].x, x2[2].y, x2[3].x, x2[3].y};
                                const float qs[8] = {qa.x, qa.y, qa.z, qa.w, qc.x, qc.y, qc.z, qc.w};
                                float r = in_leaf == 0 ? 0.0f : s_tree2<OP>(acc[rr][g]);
#pragma unroll
                                for (int j = 0; j < 8; ++j)
                                    if (j < cnt) r = s_comb<OP>(r, s_term<OP>(xs[j], qs[j]));
                                res[rr][g] = r;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    in_leaf += cnt;
                    if (in_leaf == L) end_leaf();
                }
            }
        }

        if (++c_ks == ksteps) {
            c_ks = 0;
            c_item += gridDim.x;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int row = row0 + rr * TPS;
                const int64_t chunk_off = c_it.out_rel + row;
                const bool live = row < c_it.rows;
#pragma unroll
                for (int g = 0; g < GT; ++g) {
                    if (gq(g) < c_it.qcnt) {
                        float v = one_leaf ? res[rr][g] : St[(g_first + g) * S_NT + row];
                        if (p.post == POST_SQRT) v = sqrtf(v);
                        else if (p.post == POST_NEG) v = -v;
                        const int64_t ob = uload(p.sq_out + c_it.qbeg + gq(g));
                        if (live) p.out[ob + chunk_off] = v;
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------
// Row-resident form of the batch sums, for rows of D = 32, 64 or 128 floats that are one leaf of the summation
// plan: a thread holds its row of the chunk in registers and sums it against every query of the item, up to
// RR_GROUP of them, instead of re-reading staged row pieces from LDS for every group of 8 or 16 queries and meeting
// the block at a barrier after every 32 columns.
//   Rows: a wave brings its 64 rows in through LDS in slabs of 16 columns (LDS-DMA, 64 consecutive bytes per row, so
// the loads are whole half lines; a thread that fetched its own row 16 bytes at a time would touch 64 lines per
// instruction and the kernel would be bound by the address path of the vector cache).  Piece c of row r sits at
// position c ^ ((r >> 2) & 3) of the row's 64-byte slot (swizzled on the source side), which makes the
// one-row-per-lane 16-byte reads conflict free.  The ring of RR_NB slabs is private to the wave: no barrier.  The
// first RR_NB slabs of the NEXT item are fetched while this item is summed; the rest of a row (D = 128) follows
// when the first have been moved to registers.  (Measured per item and wave at 10M x 128, 13 queries per item:
// 29 k cycles summing, 3.3 k issuing the second half's sixteen DMA instructions, 2 k waiting for it, 3 k issuing
// the next item's first half, 0.6 k moving slabs to registers, 1.3 k at the step ends.  Starting the first query
// pair before the second half has arrived, staging through registers instead of DMA, and handing the row registers
// over to the next item's loads during the last pair were each no faster or slower.)
//   Queries: through LDS, RR_TQ at a time, double buffered, one barrier per RR_TQ queries; every lane reads the same
// values (broadcast reads), one unit of 8 columns ahead of their use.
//   Items are the (row chunk, query group) records of stream_sums_kernel with groups of up to RR_GROUP queries,
// walked as one sequence of (item, query tile) steps.  Sums are taken exactly as there: eight strided accumulators as
// four packed pairs, then the tree.
static constexpr int RR_TQ = 8;
static constexpr int RR_GROUP = 128;
static constexpr int RR_NB = 4;                     // ring buffers per wave, one slab of 64 rows x 16 floats each
static constexpr size_t rowreg_lds_bytes(int D) { return (size_t)(S_NT / 64) * RR_NB * 64 * 16 * 4 + 2 * RR_TQ * (size_t)D * 4; }

template <int OP, int D, bool BOUNDED>
__global__ __launch_bounds__(S_NT, 2) void rowreg_sums_kernel(const StreamParams p) {
    static_assert(D % 16 == 0 && D <= 128, "one leaf of whole slabs");
    constexpr int H = 2;                            // queries summed side by side
    extern __shared__ __align__(16) float smem[];
    float* ring_all = smem;                                         // [waves][RR_NB][64][16]
    float (*qs)[RR_TQ][D] = reinterpret_cast<float (*)[RR_TQ][D]>(smem + (S_NT / 64) * RR_NB * 64 * 16);   // [2][RR_TQ][D]
    constexpr int C4 = D / 4;                       // 16-byte pieces of a row
    constexpr int SL = D / 16;                      // slabs of a row
    constexpr int QF4 = RR_TQ * C4;                 // 16-byte pieces of a query tile
    constexpr int NP = (QF4 + S_NT - 1) / S_NT;     // of which a thread stages this many
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (p.run_if && uload(p.run_if) == 0) return;
    const int64_t total_items = min(uload(p.seg_item_off + p.n_seg), p.items_cap);
    const int per_xcd = gridDim.x >> 3;             // runs of consecutive items per XCD, as in stream_sums_kernel
    const int64_t first = (gridDim.x & 7) == 0 ? (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    if (first >= total_items) return;

    float* ring = ring_all + wave * (RR_NB * 64 * 16);
    const unsigned ring_addr = (unsigned)(uintptr_t)ring;
    // slab sl of item jt's rows into ring buffer b: four instructions of 16 rows x 64 bytes.  The address is a
    // wave-uniform base (the item's first row of this wave, the slab's first column) plus a byte offset per lane that
    // is fixed for the launch, so issuing a slab costs no vector-ALU work.
    unsigned row_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        row_off[i] = (unsigned)((r * p.ld + c * 4) * (int)sizeof(float));
    }
    auto issue_slab = [&](const Item& jt, int sl, int b) {
        const uint64_t base = (uint64_t)(uintptr_t)(p.data + (jt.row0 + wave * 64) * p.ld + sl * 16);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        const void* sb = reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16_s(sb, row_off[i], __builtin_amdgcn_readfirstlane(ring_addr + (unsigned)(b * 4096 + i * 1024)));
    };
    float4 x[C4];
    auto take_slab = [&](int sl, int b) {
        const float* base = ring + b * 1024 + lane * 16;
        const int sw = (lane >> 2) & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) x[4 * sl + c] = *reinterpret_cast<const float4*>(base + ((c ^ sw) << 2));
    };

    int64_t item = first;
    Item it = load_item(p, item);
    int tile = 0;
    int ids[NP];
    // the query numbers of the pieces this thread stages for (item jt, tile t) ...
    auto load_ids = [&](const Item& jt, int t) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = tid + S_NT * i;
            const int jq = min(t * RR_TQ + f / C4, jt.qcnt - 1);       // slots past the group repeat its last query
            ids[i] = p.sq_qid[jt.qbeg + jq];
        }
    };
    // ... and the pieces themselves, global memory -> LDS without passing through registers (a wave's 64 pieces are
    // 1 KiB of the tile in lane order); awaited with vmcnt(0) before the barrier that publishes the tile
    auto dma_q = [&](int b) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = tid + S_NT * i;
            if (NP * S_NT == QF4 || f < QF4) {          // whole waves: QF4 is a multiple of 64
                const float* src = p.q + (int64_t)ids[i] * p.ldq + (f % C4) * 4;
                const unsigned dst = (unsigned)(uintptr_t)(&qs[b][0][0]) + (unsigned)(f - lane) * 16u;
                glds16(src, __builtin_amdgcn_readfirstlane(dst));
            }
        }
    };
    load_ids(it, 0);
#pragma unroll
    for (int sl = 0; sl < (SL < RR_NB ? SL : RR_NB); ++sl) issue_slab(it, sl, sl);
    dma_q(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (;;) {
        // the step after this one
        Item nit = it;
        int ntile = tile + 1;
        int64_t nitem = item;
        bool has_next = true;
        if (ntile * RR_TQ >= it.qcnt) {
            nitem = item + gridDim.x;
            ntile = 0;
            has_next = nitem < total_items;
            if (has_next) nit = load_item(p, nitem);
        }
        if (tile == 0) {
            // this item's rows: the first slabs are in the ring (fetched while the item before was summed; rows
            // past the chunk's end come from the rows behind it, a row store keeps 256 rows of slack, and their
            // sums are not stored)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int sl = 0; sl < (SL < RR_NB ? SL : RR_NB); ++sl) take_slab(sl, sl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the ring is free again
            if constexpr (SL > RR_NB) {
#pragma unroll
                for (int sl = RR_NB; sl < SL; ++sl) issue_slab(it, sl, sl - RR_NB);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int sl = RR_NB; sl < SL; ++sl) take_slab(sl, sl - RR_NB);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (item + gridDim.x < total_items) {
                const Item pit = load_item(p, item + gridDim.x);           // the block's next item, whatever its tiles
#pragma unroll
                for (int sl = 0; sl < (SL < RR_NB ? SL : RR_NB); ++sl) issue_slab(pit, sl, sl);
            }
        }
        if (has_next) load_ids(nit, ntile);
        const int live = min(RR_TQ, it.qcnt - tile * RR_TQ);
        const int pairs = (live + H - 1) / H;
        // the H queries' values for columns 8c .. 8c+7 of the row, read one unit ahead of their use (every lane reads
        // the same address); a pair's first unit is read while the pair before it is finished and stored
        float4 qv[2][H][2];
        auto load_qv = [&](const float* qa, int b, int c) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                qv[b][h][0] = *reinterpret_cast<const float4*>(qa + h * D + c * 8);
                qv[b][h][1] = *reinterpret_cast<const float4*>(qa + h * D + c * 8 + 4);
            }
        };
        load_qv(&qs[buf][0][0], 0, 0);
        for (int pr = 0; pr < pairs; ++pr) {
            const float* qa = &qs[buf][H * pr][0];
            // where the pair's results go (or, bounded form, the queries' numbers and bounds): asked for now, needed
            // when the row has been summed
            int64_t ob[H];
            int qn[H];
            float thr[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int e = it.qbeg + min(tile * RR_TQ + H * pr + h, it.qcnt - 1);
                ob[h] = 0; qn[h] = 0; thr[h] = 0.0f;
                if constexpr (BOUNDED) {
                    qn[h] = uload(p.sq_qid + e);
                    thr[h] = uload(p.bound + qn[h]);
                } else {
                    ob[h] = uload(p.sq_out + e);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f2 acc[H][4];
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][j] = f2{0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < D / 8; ++c) {
                if (c + 1 < D / 8) load_qv(qa, (c + 1) & 1, c + 1);
                else if (pr + 1 < pairs) load_qv(qa + H * D, 0, 0);        // D / 8 is even: the next pair starts in buffer 0
                f2 q2[H][4], x2[4];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float4 a = qv[c & 1][h][0], b = qv[c & 1][h][1];
                    q2[h][0] = f2{a.x, a.y}; q2[h][1] = f2{a.z, a.w}; q2[h][2] = f2{b.x, b.y}; q2[h][3] = f2{b.z, b.w};
                }
                x2[0] = f2{x[2 * c].x, x[2 * c].y}; x2[1] = f2{x[2 * c].z, x[2 * c].w};
                x2[2] = f2{x[2 * c + 1].x, x[2 * c + 1].y}; x2[3] = f2{x[2 * c + 1].z, x[2 * c + 1].w};
                if constexpr (s_staged<OP>()) {
                    f2 d[H * 4];
#pragma unroll
                    for (int h = 0; h < H; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) d[h * 4 + j] = s_stage1<OP>(x2[j], q2[h][j]);
                    s_fence<H * 4>(d);
                    if constexpr (OP == OP_SQDIFF) {
#pragma unroll
                        for (int e = 0; e < H * 4; ++e) d[e] = d[e] * d[e];
                        s_fence<H * 4>(d);
                    }
#pragma unroll
                    for (int h = 0; h < H; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[h][j] = s_stage3<OP>(acc[h][j], d[h * 4 + j]);
                } else {
#pragma unroll
                    for (int h = 0; h < H; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[h][j] = s_accum2<OP>(acc[h][j], x2[j], q2[h][j]);
                }
                // the running sums are taken here, not collected at the end of the row (the compiler would otherwise
                // keep every unit's terms alive until then, in registers the row needs)
                s_fence<H * 4>(&acc[0][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int gq = tile * RR_TQ + H * pr + h;              // the query's place in the group
                if (gq < it.qcnt) {
                    float v = s_tree2<OP>(acc[h]);
                    if (p.post == POST_SQRT) v = sqrtf(v);
                    else if (p.post == POST_NEG) v = -v;
                    if constexpr (BOUNDED) {
                        // in reach of the query's k best?  (squares: the bound squared, a few ulps up, as no row within
                        // the bound exceeds; rows equal to the bound stay in: ties are settled by row number later)
                        float lim = thr[h];
                        if (p.bound_sqrt) lim = fmaxf(__fmaf_rn(lim * lim, 4e-7f, lim * lim), 1e-36f);
                        const int64_t row = it.row0 + tid;
                        bool in = tid < it.rows && v <= lim;
                        if (in && p.mask) in = (p.mask[row >> 5] >> (row & 31)) & 1u;
                        const unsigned long long m = __ballot(in);
                        if (m != 0ull) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(p.queue_len + qn[h], __popcll(m));
                            base = __builtin_amdgcn_readfirstlane(base);
                            const int at = base + __popcll(m & ((1ull << lane) - 1ull));
                            if (in) {
                                if (at < p.queue_cap) {
                                    p.queue_d[(int64_t)qn[h] * p.queue_cap + at] = p.bound_sqrt ? sqrtf(v) : v;
                                    p.queue_i[(int64_t)qn[h] * p.queue_cap + at] = (int)row;
                                } else {
                                    *p.overflow = 1;
                                }
                            }
                        }
                    } else if (tid < it.rows) {
                        p.out[ob[h] + it.out_rel + tid] = v;
                    }
                }
            }
            // the coming step's query values into the other buffer: their numbers were asked for a pair of queries ago
            if (pr == 0 && has_next) dma_q(buf ^ 1);
        }
        if (!has_next) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        it = nit; tile = ntile; item = nitem;
    }
}

static bool rowreg_fits(const StreamJob& j) {
    const bool op_ok = j.op == OP_SQDIFF || j.op == OP_ABSDIFF || j.op == OP_MAXABS || j.op == OP_DOT;
    return op_ok && j.dim == j.ld && j.ld == j.ldq && j.plan.n_leaves == 1 && (j.dim == 128 || j.dim == 64 || j.dim == 32);
}
// the query group width a batch job over these rows should be planned with.  n_rows < 0: many row chunks (an IVF list
// scan).  A small row set (centroids) keeps the 16-query items: with groups of 128 there would be fewer items than
// blocks.
int stream_batch_group(int dim, int ld, const SumPlan& plan, int op, int nq, int64_t n_rows) {
    static const int env_group = getenv("VDB_STREAM_GROUP") ? atoi(getenv("VDB_STREAM_GROUP")) : 0;
    if (env_group == 8 || env_group == 16) return env_group;
    StreamJob j{};
    j.dim = dim; j.ld = ld; j.ldq = ld; j.plan = plan; j.op = op;
    const int64_t items = n_rows < 0 ? INT64_MAX : ((n_rows + S_NT - 1) / S_NT) * ((nq + RR_GROUP - 1) / RR_GROUP);
    // VDB_RR_MIN_ITEMS: the item count from which the row-resident kernel is taken (tests run it on small row sets)
    static const int64_t min_items = getenv("VDB_RR_MIN_ITEMS") ? atoll(getenv("VDB_RR_MIN_ITEMS")) : 8 * (int64_t)ctx().num_cus;
    if (rowreg_fits(j) && items >= min_items) return RR_GROUP;
    return nq >= 16 ? 16 : 8;
}

static int launch_rowreg_sums(const StreamJob& j, const StreamParams& p, hipStream_t s) {
    const int grid = ctx().num_cus * 2;
    const size_t lds = rowreg_lds_bytes(j.dim);
    KernelTimer timer(s, j.timed);
#define VDB_LAUNCH_RR_K(KERN)                                                                     \
    {                                                                                            \
        auto kern = KERN;                                                                        \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(S_NT), lds, s, p);                             \
    }
#define VDB_LAUNCH_RR_D(O, DD)                                                                   \
    {                                                                                            \
        if (p.bound) VDB_LAUNCH_RR_K((rowreg_sums_kernel<O, DD, true>))                          \
        else VDB_LAUNCH_RR_K((rowreg_sums_kernel<O, DD, false>))                                 \
    }
#define VDB_LAUNCH_RR(O)                                                                         \
    case O:                                                                                      \
        if (j.dim == 128) VDB_LAUNCH_RR_D(O, 128) else if (j.dim == 64) VDB_LAUNCH_RR_D(O, 64) else VDB_LAUNCH_RR_D(O, 32) \
        break;
    switch (j.op) {
        VDB_LAUNCH_RR(OP_DOT)
        VDB_LAUNCH_RR(OP_SQDIFF)
        VDB_LAUNCH_RR(OP_ABSDIFF)
        VDB_LAUNCH_RR(OP_MAXABS)
        default: VDB_REQUIRE(false, "row-resident sums: unsupported op %d", j.op);
    }
#undef VDB_LAUNCH_RR
#undef VDB_LAUNCH_RR_D
#undef VDB_LAUNCH_RR_K
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
static int stream_depth(const SumPlan& plan) {
    int sp = 0, depth = 0;
    for (int i = 0; i < plan.n_leaves; ++i) {
        ++sp;
        depth = std::max(depth, sp);
        sp -= plan.folds[i];
    }
    return plan.n_leaves > 1 ? depth : 0;
}

int launch_stream_sums(const StreamJob& j, hipStream_t s) {
    VDB_REQUIRE(j.group == 1 || j.group == 8 || j.group == 16 || (j.group == RR_GROUP && rowreg_fits(j)),
                "query group must be 1, 8 or 16 (or %d for rows that fit the row-resident kernel)", RR_GROUP);
    VDB_REQUIRE(j.ld % 4 == 0 && j.ld >= j.dim, "stored rows must be padded to a multiple of 4 floats");
    StreamParams p{};
    p.data = j.data; p.ld = j.ld; p.dim = j.dim; p.post = j.post;
    p.n_leaves = j.plan.n_leaves;
    for (int i = 0; i < j.plan.n_leaves; ++i) { p.leaf_len[i] = j.plan.leaf_len[i]; p.folds[i] = j.plan.folds[i]; }
    p.q = j.q; p.ldq = j.ldq; p.zeros = ctx().zeros;
    p.n_seg = j.n_seg; p.seg_row_off = j.seg_row_off; p.seg_q_off = j.seg_q_off; p.seg_item_off = j.seg_item_off;
    p.sq_qid = j.sq_qid; p.sq_out = j.sq_out; p.out = j.out; p.items = j.items; p.items_cap = j.items_cap;
    p.reverse_from = j.reverse_items > 0 ? j.reverse_items - 1 : -1;
    p.run_if = j.run_if;
    VDB_REQUIRE(j.bound == nullptr || (j.group == RR_GROUP && j.queue_len && j.queue_d && j.queue_i && j.queue_cap > 0 && j.overflow),
                "the bounded form belongs to the row-resident kernel and needs its queues");
    p.bound = j.bound; p.bound_sqrt = j.bound_sqrt ? 1 : 0; p.mask = j.mask; p.queue_len = j.queue_len; p.queue_d = j.queue_d;
    p.queue_i = j.queue_i; p.queue_cap = j.queue_cap; p.overflow = j.overflow;
    if (j.group == RR_GROUP) return launch_rowreg_sums(j, p, s);
    const size_t stack = (size_t)stream_depth(j.plan) * j.group * S_NT * sizeof(float);
    const size_t budget = 160 * 1024;
    const size_t slot = (size_t)(S_STAGE + j.group * S_CK) * sizeof(float);
    VDB_REQUIRE(stack + 2 * slot <= budget, "dimension %d is too large for the streaming scan", j.dim);
    // one query: bandwidth-bound, one block per CU with up to four stages in flight.
    // eight queries: arithmetic-bound, two independent blocks per CU (two stage buffers each),
    // so that one block's loads, waits and barriers hide behind the other's arithmetic.
    static const int env_nbuf = getenv("VDB_STREAM_NBUF") ? atoi(getenv("VDB_STREAM_NBUF")) : 0;
    static const int env_rows = getenv("VDB_STREAM_ROWS") ? atoi(getenv("VDB_STREAM_ROWS")) : 0;
    const int rows_per_thread = env_rows == 1 ? 1 : 2;
    static const bool no_whole = getenv("VDB_NO_WHOLE") != nullptr;
    bool whole = !no_whole && j.dim == j.ld;
    for (int i = 0; i < j.plan.n_leaves; ++i) whole = whole && j.plan.leaf_len[i] % S_CK == 0;
    // 16 queries per row chunk, two rows per thread: one block of eight waves per CU with four stages in flight
    // (half the global->LDS traffic per (row, query) pair of the 8-query form)
    const bool g16r2 = j.group == 16 && rows_per_thread == 2 && whole;
    const int want_nbuf = env_nbuf > 0 ? env_nbuf : ((j.group == 1 || g16r2) ? 4 : 2);
    p.nbuf = (int)std::max<size_t>(2, std::min<size_t>(want_nbuf, (budget - stack) / slot));
    const size_t lds = (size_t)p.nbuf * slot + stack;
    const int blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, budget / lds));
    const int grid = ctx().num_cus * blocks_per_cu;   // blocks past the item count leave at once
    KernelTimer timer(s, j.timed);
#define VDB_LAUNCH_STREAM(O, GG, QQ, RR)                                                        \
    {                                                                                            \
        /* two rows per thread only where every leaf is whole stages: the general path needs too many registers */ \
        auto kern = whole ? stream_sums_kernel<O, GG, QQ, true, RR> : stream_sums_kernel<O, GG, QQ, false, 1>; \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(S_NT / (whole ? RR : 1) * QQ), lds, s, p);     \
    }
#define VDB_LAUNCH_STREAM_OP(O)                                                                  \
    case O:                                                                                      \
        if (j.group == 1) VDB_LAUNCH_STREAM(O, 1, 1, 1)                                           \
        else if (j.group == 8 && rows_per_thread == 2) VDB_LAUNCH_STREAM(O, 8, STREAM_QS, 2)      \
        else if (j.group == 8) VDB_LAUNCH_STREAM(O, 8, STREAM_QS, 1)                              \
        else if (g16r2) VDB_LAUNCH_STREAM(O, 16, 4, 2)                                            \
        else VDB_LAUNCH_STREAM(O, 16, STREAM_QS, 1)                                               \
        break;
    switch (j.op) {
        VDB_LAUNCH_STREAM_OP(OP_DOT)
        VDB_LAUNCH_STREAM_OP(OP_SQDIFF)
        VDB_LAUNCH_STREAM_OP(OP_ABSDIFF)
        VDB_LAUNCH_STREAM_OP(OP_MAXABS)
        VDB_LAUNCH_STREAM_OP(OP_NEQ)
        case OP_SQ:
            VDB_REQUIRE(j.group == 1, "squared norms run one row set at a time");
            VDB_LAUNCH_STREAM(OP_SQ, 1, 1, 1)
            break;
        default: VDB_REQUIRE(false, "unknown streaming op %d", j.op);
    }
#undef VDB_LAUNCH_STREAM_OP
#undef VDB_LAUNCH_STREAM
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
static int launch_build_items(const StreamPlanBufs& b, const int64_t* seg_row_beg, const int64_t* seg_row_end, int n_seg, int group,
                              hipStream_t s);

// plan of a flat scan: one segment with every row and every query
__global__ void flat_plan_kernel(int64_t n, int nq, int group, int64_t out_stride,
                                 int64_t* seg_row_off, int* seg_q_off, int64_t* seg_item_off,
                                 int* sq_qid, int64_t* sq_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        seg_row_off[0] = 0; seg_row_off[1] = n;
        seg_q_off[0] = 0; seg_q_off[1] = nq;
        seg_item_off[0] = 0;
        seg_item_off[1] = ((n + S_NT - 1) / S_NT) * ((nq + group - 1) / group);
    }
    if (t < nq) { sq_qid[t] = t; sq_out[t] = (int64_t)t * out_stride; }
}

int launch_flat_plan(int64_t n, int nq, int group, int64_t out_stride, const StreamPlanBufs& b, hipStream_t s) {
    hipLaunchKernelGGL(flat_plan_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, n, nq, group, out_stride,
                       b.seg_row_off, b.seg_q_off, b.seg_item_off, b.sq_qid, b.sq_out);
    VDB_HIP(hipGetLastError());
    return launch_build_items(b, b.seg_row_off, b.seg_row_off + 1, 1, group, s);
}


// expands the per-segment item counts into the item table (one thread per item).  Segment i holds the stored rows
// [seg_row_beg[i], seg_row_end[i]); segments need not be adjacent or in order (lists that were moved to the end of
// the row buffer when they outgrew their place).
__global__ void build_items_kernel(int n_seg, int group, const int64_t* __restrict__ seg_row_beg, const int64_t* __restrict__ seg_row_end,
                                   const int* __restrict__ seg_q_off, const int64_t* __restrict__ seg_item_off,
                                   int64_t cap, StreamItem* __restrict__ items) {
    const int64_t total = min(seg_item_off[n_seg], cap);
    for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_seg;                      // seg_item_off[lo] <= item < seg_item_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_item_off[mid] <= item) lo = mid; else hi = mid;
        }
        const int64_t local = item - seg_item_off[lo];
        const int q0 = seg_q_off[lo];
        const int nqs = seg_q_off[lo + 1] - q0;
        const int groups = (nqs + group - 1) / group;
        const int64_t chunk = local / groups;
        const int grp = (int)(local - chunk * groups);
        const int64_t r0 = seg_row_beg[lo], r1 = seg_row_end[lo];
        StreamItem it;
        it.out_rel = chunk * S_NT;
        it.row0 = r0 + it.out_rel;
        it.rows = (int)min<int64_t>(S_NT, r1 - it.row0);
        it.qbeg = q0 + grp * group;
        it.qcnt = min(group, nqs - grp * group);
        it.pad = 0;
        items[item] = it;
    }
}

static int launch_build_items(const StreamPlanBufs& b, const int64_t* seg_row_beg, const int64_t* seg_row_end, int n_seg, int group,
                              hipStream_t s) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((b.items_cap + 255) / 256, 4096));
    hipLaunchKernelGGL(build_items_kernel, dim3(grid), dim3(256), 0, s, n_seg, group, seg_row_beg, seg_row_end, b.seg_q_off,
                       b.seg_item_off, b.items_cap, b.items);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// plan of an IVF list scan.  probes[nq][n_probe] lists the probed inverted lists
// of every query in ranking order.  Candidate slot of (query q, probe p, list row
// r) = q * cand_stride + cand_off[q][p] + r, so a query's slots follow the order
// in which the reference visits its candidates (index/ivf.py:619-628).
// A probe entry outside [0, n_lists) (a query whose centroid distances are all NaN has
// no ranking) stands for an empty list.
__global__ __launch_bounds__(256) void ivf_plan_count_kernel(const int* __restrict__ probes, int nq, int n_probe, int n_lists,
                                                             const int64_t* __restrict__ list_beg, const int64_t* __restrict__ list_end,
                                                             int* __restrict__ list_cnt, int64_t* __restrict__ cand_off,
                                                             int* __restrict__ tile_off) {
    // one wave per query: the lanes take 64 probes at a time and a prefix sum gives every probe its offsets
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    int64_t run = 0;
    int tiles = 0;
    for (int base = 0; base < n_probe; base += 64) {
        const int pr = base + lane;
        const int l = pr < n_probe ? probes[(int64_t)q * n_probe + pr] : -1;
        const bool ok = l >= 0 && l < n_lists;
        const int64_t len = ok ? list_end[l] - list_beg[l] : 0;
        const int t = (int)((len + LIST_NT - 1) / LIST_NT);
        int64_t sum_len = len;
        int sum_t = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t a = __shfl_up(sum_len, off);
            const int b = __shfl_up(sum_t, off);
            if (lane >= off) { sum_len += a; sum_t += b; }
        }
        if (pr < n_probe) {
            cand_off[(int64_t)q * (n_probe + 1) + pr] = run + sum_len - len;
            tile_off[(int64_t)q * (n_probe + 1) + pr] = tiles + sum_t - t;
        }
        if (ok && list_cnt) atomicAdd(&list_cnt[l], 1);
        run += __shfl(sum_len, 63);
        tiles += __shfl(sum_t, 63);
    }
    if (lane == 0) {
        cand_off[(int64_t)q * (n_probe + 1) + n_probe] = run;
        tile_off[(int64_t)q * (n_probe + 1) + n_probe] = tiles;
    }
}

// exclusive scans over the lists: query entries and items (single block)
__global__ __launch_bounds__(1024) void ivf_plan_scan_kernel(const int* __restrict__ list_cnt, const int64_t* __restrict__ list_beg,
                                                             const int64_t* __restrict__ list_end,
                                                             int n_lists, int group, int* __restrict__ seg_q_off,
                                                             int64_t* __restrict__ seg_item_off, int* __restrict__ cursor) {
    __shared__ int64_t sh_items[16];             // the waves' totals
    __shared__ int sh_cnt[16];
    __shared__ int64_t carry_items;
    __shared__ int carry_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_items = 0; carry_cnt = 0; }
    __syncthreads();
    for (int base = 0; base < n_lists; base += 1024) {
        const int l = base + tid;
        int c = 0;
        int64_t items = 0;
        if (l < n_lists) {
            c = list_cnt[l];
            const int64_t len = list_end[l] - list_beg[l];
            items = ((len + S_NT - 1) / S_NT) * ((c + group - 1) / group);
        }
        // inclusive sums over the wave, then over the sixteen waves' totals
        int sum_c = c;
        int64_t sum_i = items;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int a = __shfl_up(sum_c, off);
            const int64_t b = __shfl_up(sum_i, off);
            if (lane >= off) { sum_c += a; sum_i += b; }
        }
        if (lane == 63) { sh_cnt[wave] = sum_c; sh_items[wave] = sum_i; }
        __syncthreads();
        int before_c = carry_cnt;
        int64_t before_i = carry_items;
        for (int w = 0; w < wave; ++w) { before_c += sh_cnt[w]; before_i += sh_items[w]; }
        if (l < n_lists) {
            seg_q_off[l] = before_c + sum_c - c;
            seg_item_off[l] = before_i + sum_i - items;
            cursor[l] = before_c + sum_c - c;
        }
        __syncthreads();
        if (tid == 1023) { carry_cnt = before_c + sum_c; carry_items = before_i + sum_i; }
        __syncthreads();
    }
    if (tid == 0) { seg_q_off[n_lists] = carry_cnt; seg_item_off[n_lists] = carry_items; }
}

__global__ void ivf_plan_fill_kernel(const int* __restrict__ probes, int nq, int n_probe, int n_lists, int64_t cand_stride,
                                     const int64_t* __restrict__ cand_off, int* __restrict__ cursor,
                                     int* __restrict__ sq_qid, int64_t* __restrict__ sq_out, int* __restrict__ sq_list) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)nq * n_probe) return;
    const int q = (int)(t / n_probe), pr = (int)(t - (int64_t)q * n_probe);
    const int l = probes[t];
    if (l < 0 || l >= n_lists) return;
    const int slot = atomicAdd(&cursor[l], 1);
    sq_qid[slot] = q;
    sq_out[slot] = (int64_t)q * cand_stride + cand_off[(int64_t)q * (n_probe + 1) + pr];
    if (sq_list) sq_list[slot] = l;
}

int launch_ivf_plan(const int* probes, int nq, int n_probe, const int64_t* list_beg, const int64_t* list_end, int n_lists, int group,
                    int64_t cand_stride, int* list_cnt, int* cursor, int64_t* cand_off, int* tile_off,
                    const StreamPlanBufs& b, hipStream_t s) {
    VDB_HIP(hipMemsetAsync(list_cnt, 0, (size_t)n_lists * sizeof(int), s));
    hipLaunchKernelGGL(ivf_plan_count_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, probes, nq, n_probe, n_lists,
                       list_beg, list_end, list_cnt, cand_off, tile_off);
    hipLaunchKernelGGL(ivf_plan_scan_kernel, dim3(1), dim3(1024), 0, s, list_cnt, list_beg, list_end, n_lists, group, b.seg_q_off,
                       b.seg_item_off, cursor);
    const int64_t total = (int64_t)nq * n_probe;
    hipLaunchKernelGGL(ivf_plan_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, probes, nq, n_probe,
                       n_lists, cand_stride, cand_off, cursor, b.sq_qid, b.sq_out, (int*)nullptr);
    VDB_HIP(hipGetLastError());
    return launch_build_items(b, list_beg, list_end, n_lists, group, s);
}

// Where each of `blocks` blocks starts in the units in list order so that all get about the same number of
// (256-row tile, query) items: seg_item_off is the running item count over the lists (group = 1), seg_q_off the running
// unit count.  first[b] = block b's first unit, first[blocks] = the number of units.
__global__ void ivf_unit_ranges_kernel(const int* __restrict__ seg_q_off, const int64_t* __restrict__ seg_item_off,
                                       const int64_t* __restrict__ list_beg, const int64_t* __restrict__ list_end,
                                       int n_lists, int blocks, int* __restrict__ first) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > blocks) return;
    const int64_t all = seg_item_off[n_lists];
    if (b == blocks) { first[b] = seg_q_off[n_lists]; return; }
    const int64_t want = all * b / blocks;       // items in front of block b
    int lo = 0, hi = n_lists;                    // the list the want-th item falls into: seg_item_off[lo] <= want < seg_item_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_item_off[mid] <= want) lo = mid; else hi = mid;
    }
    const int64_t tiles = (list_end[lo] - list_beg[lo] + S_NT - 1) / S_NT;
    const int cnt = seg_q_off[lo + 1] - seg_q_off[lo];
    const int64_t into = tiles > 0 ? (want - seg_item_off[lo] + tiles - 1) / tiles : 0;      // whole units of the list before the block
    first[b] = b == 0 ? 0 : seg_q_off[lo] + (int)(into < cnt ? into : cnt);
}

// The (query, probe) units in list order, for a scan that walks them itself: slot offsets as above, then per unit its
// query, the start of its candidate slots and its list (b.sq_qid, b.sq_out, sq_list), the units of one list side by
// side (probes of no list are left out), and first[0..blocks]: the unit each of `blocks` blocks starts at.  No item table.
int launch_ivf_units_by_list(const int* probes, int nq, int n_probe, const int64_t* list_beg, const int64_t* list_end, int n_lists,
                             int64_t cand_stride, int* list_cnt, int* cursor, int64_t* cand_off, int* tile_off,
                             const StreamPlanBufs& b, int* sq_list, int blocks, int* first, hipStream_t s) {
    VDB_HIP(hipMemsetAsync(list_cnt, 0, (size_t)n_lists * sizeof(int), s));
    hipLaunchKernelGGL(ivf_plan_count_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, probes, nq, n_probe, n_lists,
                       list_beg, list_end, list_cnt, cand_off, tile_off);
    hipLaunchKernelGGL(ivf_plan_scan_kernel, dim3(1), dim3(1024), 0, s, list_cnt, list_beg, list_end, n_lists, 1, b.seg_q_off,
                       b.seg_item_off, cursor);
    const int64_t total = (int64_t)nq * n_probe;
    hipLaunchKernelGGL(ivf_plan_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, probes, nq, n_probe,
                       n_lists, cand_stride, cand_off, cursor, b.sq_qid, b.sq_out, sq_list);
    hipLaunchKernelGGL(ivf_unit_ranges_kernel, dim3((blocks + 256) / 256), dim3(256), 0, s, b.seg_q_off, b.seg_item_off, list_beg, list_end,
                       n_lists, blocks, first);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// only the offsets of every (query, probe) in the query's run of candidate slots and tiles (what a scan that walks
// the units itself needs of the plan)
int launch_ivf_slot_offsets(const int* probes, int nq, int n_probe, const int64_t* list_beg, const int64_t* list_end, int n_lists,
                            int64_t* cand_off, int* tile_off, hipStream_t s) {
    if (nq <= 0) return VDB_OK;
    hipLaunchKernelGGL(ivf_plan_count_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, probes, nq, n_probe, n_lists,
                       list_beg, list_end, nullptr, cand_off, tile_off);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// distance finish: the streamed sum -> the metric's value
struct Finish {
    int metric;
    const float* q_sq; const float* q_nrm;       // per query
    const float* row_sq; const float* row_nrm;   // per stored row
    float cos_eps;                               // cosine matrix form: norms below it count as 1
};

// arccos of a similarity the way the reference takes it (metrics.py:535-538): the fp32 quotient goes to
// double, is clipped, and np.arccos runs in double; the result is rounded to fp32 for the caller
__device__ __forceinline__ float angular_of(float sim) {
    const double s = fmin(fmax((double)sim, -1.0), 1.0);
    return (float)acos(s);
}

__device__ __forceinline__ float finish_value(const Finish& f, float sum, float q_sq, float q_nrm, int64_t pos) {
    switch (f.metric) {
        case VDB_EUCLIDEAN: case VDB_EUCLIDEAN_SQ: {
            const float sq = fmaxf(__fmaf_rn(-2.0f, sum, q_sq + f.row_sq[pos]), 0.0f);     // metrics.py:456-462
            return f.metric == VDB_EUCLIDEAN ? sqrtf(sq) : sq;
        }
        case VDB_COSINE: {
            if (q_nrm < 1e-8f && f.cos_eps <= 1e-8f) return 1.0f;    // metrics.py:472-473
            if (q_nrm < f.cos_eps) q_nrm = 1.0f;                     // metrics.py:360-361 with a caller's eps
            float cn = f.row_nrm[pos];
            if (cn < f.cos_eps) cn = 1.0f;                           // metrics.py:476, :362-363
            const float sim = fminf(fmaxf(sum / (q_nrm * cn), -1.0f), 1.0f);
            return 1.0f - sim;
        }
        case VDB_COSINE_PAIR: {
            const float den = q_nrm * f.row_nrm[pos];                // metrics.py:96-100
            const float sim = den < 1e-8f ? 0.0f : sum / den;
            return 1.0f - sim;
        }
        case VDB_ANGULAR: {
            const float den = q_nrm * f.row_nrm[pos];                // metrics.py:96-100 via :535
            return angular_of(den < 1e-8f ? 0.0f : sum / den);
        }
        default: return sum;      // already final (direct forms, Manhattan, Chebyshev, negated dot)
    }
}

__global__ void finish_dist_kernel(Finish f, float* __restrict__ v, int nq, int64_t n) {
    const int64_t total = (int64_t)nq * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i / n);
        const int64_t pos = i - (int64_t)q * n;
        v[i] = finish_value(f, v[i], f.q_sq ? f.q_sq[q] : 0.0f, f.q_nrm ? f.q_nrm[q] : 0.0f, pos);
    }
}

int launch_finish_dist(int metric, const float* q_sq, const float* q_nrm, const float* row_sq, const float* row_nrm,
                       float cos_eps, float* v, int nq, int64_t n, hipStream_t s) {
    if (metric != VDB_EUCLIDEAN && metric != VDB_EUCLIDEAN_SQ && metric != VDB_COSINE && metric != VDB_COSINE_PAIR &&
        metric != VDB_ANGULAR)
        return VDB_OK;
    Finish f{metric, q_sq, q_nrm, row_sq, row_nrm, cos_eps};
    const int64_t total = (int64_t)nq * n;
    hipLaunchKernelGGL(finish_dist_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 16384)), dim3(256), 0, s,
                       f, v, nq, n);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// per query: finish, filter and keep the k best candidates of its slice
struct CandParams {
    const float* cand; int64_t cand_stride;      // query q owns cand[q*cand_stride ...]
    int64_t n;                                   // flat: candidates per query (slot = stored row)
    const int* probes; int n_probe; int n_lists; // lists: probed lists of every query, or null
    const int64_t* list_beg; const int64_t* list_end;
    const int64_t* cand_off; const int* tile_off; // lists: [nq][n_probe+1] slots / 256-slot tiles before every probe
    const int* row_ids;                          // stored position -> row number (null = identity)
    const uint32_t* mask;                        // by row number
    Finish fin;
    const float* lb_d; const int64_t* lb_k;      // exclusive lower bound (distance, slot) per query
    float radius; int use_radius;
    int k; int splits;
    float* part_d; int* part_i;                  // [nq][splits][k], idx = candidate slot
    // One launch for selection, merge and row lookup (lists only): the block that finishes last for a query
    // (ticket[q] counts them and is left at zero again) merges every block's lists and writes the k_out results.
    int* ticket; int k_out;
    float* fin_d; int64_t* fin_rows; int64_t* fin_order;
    const int* run_if;                           // optional device flag: the launch does nothing while it is zero
    const int64_t* n_of_query; int n_stride;     // flat: candidates of query q = n_of_query[q * n_stride] instead of n
    int block_list;                              // the block's wave lists leave as one list: part_* is [nq][splits][k]
};

static constexpr int CAND_WAVES = LIST_NT / 64;   // candidate lists one block writes
static constexpr int CAND_UNROLL = 4;       // tiles whose loads are in flight together
static constexpr int RUN_CHUNK = 256;      // probed lists whose ranges are staged in LDS at a time

// The block's CAND_WAVES wave lists as one list of k: an entry's place is the number of entries that precede it, its own
// list's (it is sorted) and the other lists' (counted).  Wd / Wi: CAND_WAVES * 64 values each, no longer read by anyone.
__device__ __forceinline__ void store_block_list(const WaveTopK& best, int k, float* Wd, int* Wi, float* bd, int* bi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Wd[wave * 64 + lane] = best.e; Wi[wave * 64 + lane] = best.eid;
    if (tid < k) { bd[tid] = INFINITY; bi[tid] = EMPTY_IDX; }
    __syncthreads();
    if (lane < k && best.eid != EMPTY_IDX) {
        int rank = lane;
        for (int w = 0; w < CAND_WAVES; ++w) {
            if (w == wave) continue;
            for (int e = 0; e < k; ++e) rank += pair_less(Wd[w * 64 + e], Wi[w * 64 + e], best.e, best.eid) ? 1 : 0;
        }
        if (rank < k) { bd[rank] = best.e; bi[rank] = best.eid; }
    }
}

// FIN: how a candidate value becomes a distance, fixed per launch so that the row terms of the
// CAND_UNROLL tiles are fetched together instead of one dependent load per tile:
// 0 = the value is the distance (direct forms, Manhattan, Chebyshev, Hamming, negated dot), 1 = the Euclidean
// identity form, 2 = cosine of the matrix form, 3 = cosine of the pair form, 4 = the angle of the pair form,
// 5 = the square root of the streamed sum (direct-form Euclidean whose scan left the root to this kernel: one
// correctly rounded sqrtf per candidate costs the scan, which is bound by its vector ALUs, more than it costs here).
// ONE: the one-launch mode (p.ticket set): few blocks that run nearly alone, register-only merge network
template <int FIN, bool ONE>
__global__ __launch_bounds__(LIST_NT) void topk_cand_kernel(CandParams p) {
    extern __shared__ __align__(16) float smem[];
    int64_t* R_pos0 = reinterpret_cast<int64_t*>(smem);         // first stored row of the run
    int64_t* R_len = R_pos0 + RUN_CHUNK;                        // rows in the run
    int* R_slot0 = reinterpret_cast<int*>(R_len + RUN_CHUNK);   // first candidate slot of the run
    int* R_tile = R_slot0 + RUN_CHUNK;                          // [RUN_CHUNK + 1] tiles before the run
    const int tid = threadIdx.x;
    const int qi = blockIdx.x, split = blockIdx.y;
    const int k = p.k;
    const int lane = tid & 63, wave = tid >> 6;
    if (p.run_if && *p.run_if == 0) return;
    WaveTopK best;
    best.init(k);
    const float q_sq = p.fin.q_sq ? p.fin.q_sq[qi] : 0.0f;
    const float q_nrm = p.fin.q_nrm ? p.fin.q_nrm[qi] : 0.0f;
    float lbd = -INFINITY;
    int64_t lbk = -1;
    if (p.lb_d) { lbd = p.lb_d[qi]; lbk = p.lb_k[qi]; }
    const float* cq = p.cand + (int64_t)qi * p.cand_stride;

    // A query's candidates are runs of consecutive slots: one run per probed list (or
    // one run holding every row).  Tiles of 256 slots are numbered through the runs and
    // dealt to the splits round-robin.
    const int n_runs = p.probes ? p.n_probe : 1;
    for (int c0 = 0; c0 < n_runs; c0 += RUN_CHUNK) {
        const int nr = min(RUN_CHUNK, n_runs - c0);
        __syncthreads();
        if (p.probes) {
            const int64_t qo = (int64_t)qi * (p.n_probe + 1) + c0;
            for (int i = tid; i < nr; i += LIST_NT) {
                const int l = p.probes[(int64_t)qi * p.n_probe + c0 + i];
                const bool ok = l >= 0 && l < p.n_lists;
                const int64_t pos0 = ok ? p.list_beg[l] : 0;
                R_pos0[i] = pos0;
                R_len[i] = ok ? p.list_end[l] - pos0 : 0;
                R_slot0[i] = (int)p.cand_off[qo + i];
            }
            for (int i = tid; i <= nr; i += LIST_NT) R_tile[i] = p.tile_off[qo + i];
        } else if (tid == 0) {
            const int64_t nq_ = p.n_of_query ? p.n_of_query[(int64_t)qi * p.n_stride] : p.n;
            R_pos0[0] = 0; R_len[0] = nq_; R_slot0[0] = 0;
            R_tile[0] = 0; R_tile[1] = (int)((nq_ + LIST_NT - 1) / LIST_NT);
        }
        __syncthreads();
        const int t_begin = R_tile[0], t_end = R_tile[nr];
        int run = 0;
        // CAND_UNROLL tiles per step: their loads are issued together, then the inserts run in tile order
        for (int t = t_begin + (split - t_begin % p.splits + p.splits) % p.splits; t < t_end; t += p.splits * CAND_UNROLL) {
            int64_t pos[CAND_UNROLL];
            int slot[CAND_UNROLL];
            bool ok[CAND_UNROLL];
#pragma unroll
            for (int u = 0; u < CAND_UNROLL; ++u) {
                const int tu = t + u * p.splits;
                ok[u] = tu < t_end;
                pos[u] = 0; slot[u] = 0;
                if (ok[u]) {
                    while (R_tile[run + 1] <= tu) ++run;
                    const int64_t r = (int64_t)(tu - R_tile[run]) * LIST_NT + tid;
                    ok[u] = r < R_len[run];
                    pos[u] = R_pos0[run] + r;
                    slot[u] = R_slot0[run] + (int)r;
                }
            }
            float raw[CAND_UNROLL];
            int rown[CAND_UNROLL];
#pragma unroll
            for (int u = 0; u < CAND_UNROLL; ++u) {
                raw[u] = ok[u] ? cq[slot[u]] : 0.0f;
                rown[u] = (ok[u] && p.row_ids && p.mask) ? p.row_ids[pos[u]] : (int)pos[u];   // row numbers serve the filter only
            }
            float rt[CAND_UNROLL];
#pragma unroll
            for (int u = 0; u < CAND_UNROLL; ++u) {
                rt[u] = 1.0f;
                if (FIN == 1 && ok[u]) rt[u] = p.fin.row_sq[pos[u]];
                if (FIN >= 2 && FIN <= 4 && ok[u]) rt[u] = p.fin.row_nrm[pos[u]];
            }
            float d[CAND_UNROLL];
#pragma unroll
            for (int u = 0; u < CAND_UNROLL; ++u) {
                float v = raw[u];
                if (FIN == 1) {
                    const float sq = fmaxf(__fmaf_rn(-2.0f, v, q_sq + rt[u]), 0.0f);            // metrics.py:456-462
                    v = p.fin.metric == VDB_EUCLIDEAN ? sqrtf(sq) : sq;
                } else if (FIN == 2) {
                    const float cn = rt[u] < 1e-8f ? 1.0f : rt[u];                              // metrics.py:476
                    const float sim = fminf(fmaxf(v / (q_nrm * cn), -1.0f), 1.0f);
                    v = q_nrm < 1e-8f ? 1.0f : 1.0f - sim;                                      // metrics.py:472-473
                } else if (FIN == 3) {
                    const float den = q_nrm * rt[u];                                            // metrics.py:96-100
                    v = 1.0f - (den < 1e-8f ? 0.0f : v / den);
                } else if (FIN == 4) {
                    const float den = q_nrm * rt[u];                                            // metrics.py:535-538
                    v = angular_of(den < 1e-8f ? 0.0f : v / den);
                } else if (FIN == 5) {
                    v = sqrtf(v);                                                               // metrics.py:43
                }
                d[u] = ok[u] ? v : INFINITY;
            }
            if (p.mask) {
#pragma unroll
                for (int u = 0; u < CAND_UNROLL; ++u)
                    if (ok[u]) ok[u] = (p.mask[rown[u] >> 5] >> (rown[u] & 31)) & 1u;
            }

#pragma unroll
            for (int u = 0; u < CAND_UNROLL; ++u) {
                bool want = ok[u];
                if (p.use_radius && !(d[u] <= p.radius)) want = false;
                if (!((d[u] > lbd) || (d[u] == lbd && slot[u] > lbk))) want = false;
                best.offer<true, ONE>(want, d[u], slot[u], lane);      // tiles in order, lanes in order: ascending slots
            }

        }
    }
    if (!ONE) {
        if (p.block_list) {
            __syncthreads();                    // the run tables in smem are no longer read
            const int64_t list = (int64_t)qi * p.splits + split;
            store_block_list(best, k, smem, reinterpret_cast<int*>(smem + CAND_WAVES * 64), p.part_d + list * k, p.part_i + list * k);
            return;
        }
        // one candidate list per wave
        const int64_t list = ((int64_t)qi * p.splits + split) * CAND_WAVES + wave;
        best.store(p.part_d + list * k, p.part_i + list * k, lane);
        return;
    }
    // ---- one launch: the block's four wave lists become one (a pair's place = the pairs that precede it), and
    // the block that finishes last for its query merges every block's list
    __syncthreads();                            // the run tables in smem are no longer read
    float* Wd = smem;                           // [CAND_WAVES][64]
    int* Wi = reinterpret_cast<int*>(smem + CAND_WAVES * 64);
    Wd[wave * 64 + lane] = best.e; Wi[wave * 64 + lane] = best.eid;
    __syncthreads();
    float* bd = p.part_d + ((int64_t)qi * p.splits + split) * k;
    int* bi = p.part_i + ((int64_t)qi * p.splits + split) * k;
    if (tid < k) { bd[tid] = INFINITY; bi[tid] = EMPTY_IDX; }
    __syncthreads();
    if (lane < k && best.eid != EMPTY_IDX) {
        int rank = lane;
        for (int w = 0; w < CAND_WAVES; ++w) {
            if (w == wave) continue;
            for (int e = 0; e < k; ++e) rank += pair_less(Wd[w * 64 + e], Wi[w * 64 + e], best.e, best.eid) ? 1 : 0;
        }
        if (rank < k) { bd[rank] = best.e; bi[rank] = best.eid; }
    }
    __shared__ int s_last;
    __threadfence();                            // this block's list is visible device-wide before its ticket
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(p.ticket + qi, 1) == p.splits - 1;
    __syncthreads();
    if (!s_last || wave != 0) return;
    __threadfence();                            // and the other blocks' lists are visible to this one
    WaveTopK fin;
    fin.init(k);
    const int total = p.splits * k;
    const float* pd = p.part_d + (int64_t)qi * total;
    const int* pi = p.part_i + (int64_t)qi * total;
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + lane;
        const bool have = c < total;
        const float d = have ? pd[c] : INFINITY;
        const int id = have ? pi[c] : EMPTY_IDX;
        fin.offer<true, true>(have && id != EMPTY_IDX, d, id, lane);
    }
    // slot -> (probe, position) -> row: the query's slot offsets and probes go to LDS in one coalesced read each,
    // so the search per result is a few LDS reads instead of a chain of global ones
    int* s_co = reinterpret_cast<int*>(smem);   // [n_probe + 1] (a query's slots are below 2^31), then its probes
    const bool small = 2 * p.n_probe + 1 <= 2 * CAND_WAVES * 64;
    if (small) {
        for (int i = lane; i <= p.n_probe; i += 64) s_co[i] = (int)p.cand_off[(int64_t)qi * (p.n_probe + 1) + i];
        for (int i = lane; i < p.n_probe; i += 64) s_co[p.n_probe + 1 + i] = p.probes[(int64_t)qi * p.n_probe + i];
    }
    if (lane < p.k_out) {
        const int64_t at = (int64_t)qi * p.k_out + lane;
        float d = INFINITY;
        int64_t row = -1, order = -1;
        if (lane < k && fin.eid != EMPTY_IDX) {
            const int slot = fin.eid;
            const int64_t* co = p.cand_off + (int64_t)qi * (p.n_probe + 1);
            int lo = 0, hi = p.n_probe;         // co[lo] <= slot < co[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((small ? (int64_t)s_co[mid] : co[mid]) <= slot) lo = mid; else hi = mid;
            }
            const int l = small ? s_co[p.n_probe + 1 + lo] : p.probes[(int64_t)qi * p.n_probe + lo];
            d = fin.e;
            const int64_t in_list = slot - (small ? (int64_t)s_co[lo] : co[lo]);
            row = p.row_ids[p.list_beg[l] + in_list];
            order = ((int64_t)lo << 32) | in_list;                    // the reference's visiting order (ivf.py:619-628)
        }
        p.fin_d[at] = d;
        p.fin_rows[at] = row;
        if (p.fin_order) p.fin_order[at] = order;
    }
    if (lane == 0) p.ticket[qi] = 0;
}


int topk_cand_lists_per_split() { return CAND_WAVES; }

// The plain case of the selection above, as its own kernel: one run of candidates per query (slot = index), the value
// is the distance or its square (SQRT), no filter, no radius, no lower bound.  Nothing but the loads, one compare and
// one ballot per 64 candidates is left on the common path; the square root is taken only in tiles where some value
// is within the square of the list's k-th distance (a few ulps up, as in the PQ scan).  Same lists out.
template <bool SQRT>
__global__ __launch_bounds__(LIST_NT) void topk_plain_kernel(CandParams p) {
    if (p.run_if && *p.run_if == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = blockIdx.x, split = blockIdx.y;
    const int k = p.k;
    const int64_t n = p.n_of_query ? p.n_of_query[(int64_t)qi * p.n_stride] : p.n;
    const float* cq = p.cand + (int64_t)qi * p.cand_stride;
    WaveTopK best;
    best.init(k);
    float thr2 = INFINITY;
    const int ntiles = (int)((n + LIST_NT - 1) / LIST_NT);
    for (int t = split; t < ntiles; t += p.splits * CAND_UNROLL) {
        float v[CAND_UNROLL];
        int slot[CAND_UNROLL];
        bool ok[CAND_UNROLL];
#pragma unroll
        for (int u = 0; u < CAND_UNROLL; ++u) {
            const int tu = t + u * p.splits;
            slot[u] = tu * LIST_NT + tid;
            ok[u] = tu < ntiles && slot[u] < n;
            v[u] = ok[u] ? cq[slot[u]] : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < CAND_UNROLL; ++u) {
            if (SQRT) {
                if (__ballot(ok[u] && v[u] <= thr2) != 0ull) {
                    best.offer<true, false>(ok[u], sqrtf(v[u]), slot[u], lane);                  // metrics.py:43
                    const float th = best.thr;
                    thr2 = fmaxf(__fmaf_rn(th * th, 4e-7f, th * th), 1e-36f);
                }
            } else {
                best.offer<true, false>(ok[u], v[u], slot[u], lane);
            }
        }
    }
    if (p.block_list) {
        __shared__ float Wd[CAND_WAVES * 64];
        __shared__ int Wi[CAND_WAVES * 64];
        const int64_t list = (int64_t)qi * p.splits + split;
        store_block_list(best, k, Wd, Wi, p.part_d + list * k, p.part_i + list * k);
        return;
    }
    const int64_t list = ((int64_t)qi * p.splits + split) * CAND_WAVES + wave;
    best.store(p.part_d + list * k, p.part_i + list * k, lane);
}

int launch_topk_cand(const CandJob& j, hipStream_t s) {
    VDB_REQUIRE(j.k >= 1 && j.k <= 64, "candidate selection supports k <= 64, got %d", j.k);
    CandParams p{};
    p.cand = j.cand; p.cand_stride = j.cand_stride; p.n = j.n; p.probes = j.probes; p.n_probe = j.n_probe; p.n_lists = j.n_lists;
    p.list_beg = j.list_beg; p.list_end = j.list_end; p.row_ids = j.row_ids; p.mask = j.mask;
    p.fin = Finish{j.metric, j.q_sq, j.q_nrm, j.row_sq, j.row_nrm, 1e-8f};
    p.lb_d = j.lb_d; p.lb_k = j.lb_k; p.radius = j.radius; p.use_radius = j.use_radius;
    p.k = j.k; p.splits = j.splits; p.part_d = j.part_d; p.part_i = j.part_i;
    p.ticket = j.ticket; p.k_out = j.k_out; p.fin_d = j.fin_d; p.fin_rows = j.fin_rows; p.fin_order = j.fin_order;
    p.run_if = j.run_if; p.n_of_query = j.n_of_query; p.n_stride = j.n_stride;
    p.block_list = j.block_list ? 1 : 0;
    VDB_REQUIRE(!(j.block_list && j.ticket), "the one-launch selection writes its own lists");
    VDB_REQUIRE(j.ticket == nullptr || (j.probes && j.row_ids && j.fin_d && j.fin_rows && j.k_out >= j.k && j.k_out <= 64),
                "the one-launch selection serves list candidates with k <= 64");
    VDB_REQUIRE(j.probes == nullptr || (j.cand_off && j.tile_off), "list candidates need their slot and tile offsets");
    p.cand_off = j.cand_off; p.tile_off = j.tile_off;
    const size_t lds = (size_t)RUN_CHUNK * 24 + 8;
    const int fin = (j.metric == VDB_EUCLIDEAN || j.metric == VDB_EUCLIDEAN_SQ) ? 1
                    : j.metric == VDB_COSINE ? 2 : j.metric == VDB_COSINE_PAIR ? 3 : j.metric == VDB_ANGULAR ? 4
                    : (j.metric == VDB_EUCLIDEAN_DIRECT && j.sqrt_here) ? 5 : 0;
    const bool one = j.ticket != nullptr;
    if ((fin == 0 || fin == 5) && !one && j.probes == nullptr && j.mask == nullptr && !j.use_radius && j.lb_d == nullptr &&
        (int64_t)((j.n + LIST_NT - 1) / LIST_NT) * LIST_NT < (int64_t)0x7fffffff) {
        if (fin == 5) hipLaunchKernelGGL(topk_plain_kernel<true>, dim3(j.nq, j.splits), dim3(LIST_NT), 0, s, p);
        else hipLaunchKernelGGL(topk_plain_kernel<false>, dim3(j.nq, j.splits), dim3(LIST_NT), 0, s, p);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
#define VDB_CAND_KERN(F) (one ? topk_cand_kernel<F, true> : topk_cand_kernel<F, false>)
    auto kern = fin == 1 ? VDB_CAND_KERN(1) : fin == 2 ? VDB_CAND_KERN(2) : fin == 3 ? VDB_CAND_KERN(3)
                : fin == 4 ? VDB_CAND_KERN(4) : fin == 5 ? VDB_CAND_KERN(5) : VDB_CAND_KERN(0);
#undef VDB_CAND_KERN
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(j.nq, j.splits), dim3(LIST_NT), lds, s, p);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// Range search (index/flat.py:501-572): the rows of one query whose distance is <= radius and that
// pass the filter, the FIRST `cap` of them in row order (the reference walks the rows in insertion
// order and stops at max_results, :545-567), then sorted by distance with ties in row order (the
// stable sort of :570).  Three passes over the candidate buffer: count per block of 1024 rows, scan
// of the block counts, ordered fill of (distance, row) keys; the keys are sorted by a bitonic network
// and the distances are recomputed from the candidate values on the way out, so a caller sees the same
// bits a search would return.
static constexpr int RANGE_BLOCK = 1024;          // rows per block: four passes of 256

__device__ __forceinline__ uint64_t range_key(float d, int64_t row) {
    if (d == 0.0f) d = 0.0f;                       // -0 and +0 are one distance
    uint32_t u = __float_as_uint(d);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;    // unsigned order = float order
    return ((uint64_t)u << 32) | (uint32_t)row;
}

struct RangeParams {
    const float* cand; int64_t n; const uint32_t* mask; Finish fin; float radius;
    int* block_counts; const int64_t* block_off; int64_t cap; uint64_t* keys;
};

template <bool FILL>
__global__ __launch_bounds__(256) void range_pass_kernel(RangeParams p) {
    __shared__ int wave_cnt[4][4];                 // [pass][wave]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float q_sq = p.fin.q_sq ? p.fin.q_sq[0] : 0.0f, q_nrm = p.fin.q_nrm ? p.fin.q_nrm[0] : 0.0f;
    const int64_t base = (int64_t)blockIdx.x * RANGE_BLOCK;
    float d[4];
    bool hit[4];
    int rank[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t row = base + ps * 256 + tid;
        hit[ps] = false;
        d[ps] = 0.0f;
        if (row < p.n) {
            d[ps] = finish_value(p.fin, p.cand[row], q_sq, q_nrm, row);
            hit[ps] = d[ps] <= p.radius;           // index/flat.py:546 (a NaN distance never passes)
            if (hit[ps] && p.mask) hit[ps] = (p.mask[row >> 5] >> (row & 31)) & 1u;
        }
        const unsigned long long b = __ballot(hit[ps]);
        rank[ps] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[ps][wave] = __popcll(b);
    }
    __syncthreads();
    if (!FILL) {
        if (tid == 0) {
            int t = 0;
            for (int i = 0; i < 16; ++i) t += (&wave_cnt[0][0])[i];
            p.block_counts[blockIdx.x] = t;
        }
        return;
    }
    int64_t before = p.block_off[blockIdx.x];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w == wave && hit[ps]) {
                const int64_t at = before + rank[ps];
                if (at < p.cap) p.keys[at] = range_key(d[ps], base + ps * 256 + tid);
            }
            before += wave_cnt[ps][w];
        }
    }
}

// exclusive scan of the block counts by one block; total[0] = number of rows inside the radius
__global__ __launch_bounds__(1024) void range_scan_kernel(const int* __restrict__ counts, int64_t nb,
                                                          int64_t* __restrict__ off, int64_t* __restrict__ total) {
    __shared__ int64_t sh[1024];
    __shared__ int64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        const int64_t b = b0 + tid;
        const int64_t c = b < nb ? counts[b] : 0;
        sh[tid] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int64_t a = tid >= o ? sh[tid - o] : 0;
            __syncthreads();
            sh[tid] += a;
            __syncthreads();
        }
        if (b < nb) off[b] = carry + sh[tid] - c;
        __syncthreads();
        if (tid == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (tid == 0) total[0] = carry;
}

// bitonic network over P = 2^x >= 1024 keys.  Element i is compared with i ^ j at stage (k, j), ascending
// where (i & k) == 0.  Strides below 1024 run inside LDS on chunks of 1024 keys.
__device__ __forceinline__ void bitonic_lds_steps(uint64_t* sh, int tid, int64_t chunk0, int64_t k, int j_first) {
    for (int j = j_first; j > 0; j >>= 1) {
        const int lo = 2 * tid - (tid & (j - 1));
        const int hi = lo + j;
        const bool up = ((chunk0 + lo) & k) == 0;
        const uint64_t a = sh[lo], b = sh[hi];
        if ((a > b) == up) { sh[lo] = b; sh[hi] = a; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512) void bitonic_chunk_sort_kernel(uint64_t* __restrict__ keys) {
    __shared__ uint64_t sh[1024];
    const int tid = threadIdx.x;
    const int64_t chunk0 = (int64_t)blockIdx.x * 1024;
    sh[tid] = keys[chunk0 + tid];
    sh[tid + 512] = keys[chunk0 + tid + 512];
    __syncthreads();
    for (int64_t k = 2; k <= 1024; k <<= 1) bitonic_lds_steps(sh, tid, chunk0, k, (int)(k >> 1));
    keys[chunk0 + tid] = sh[tid];
    keys[chunk0 + tid + 512] = sh[tid + 512];
}

__global__ __launch_bounds__(256) void bitonic_global_step_kernel(uint64_t* __restrict__ keys, int64_t half, int64_t j, int64_t k) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < half; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = 2 * t - (t & (j - 1));
        const int64_t hi = lo + j;
        const bool up = (lo & k) == 0;
        const uint64_t a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
    }
}

__global__ __launch_bounds__(512) void bitonic_chunk_merge_kernel(uint64_t* __restrict__ keys, int64_t k) {
    __shared__ uint64_t sh[1024];
    const int tid = threadIdx.x;
    const int64_t chunk0 = (int64_t)blockIdx.x * 1024;
    sh[tid] = keys[chunk0 + tid];
    sh[tid + 512] = keys[chunk0 + tid + 512];
    __syncthreads();
    bitonic_lds_steps(sh, tid, chunk0, k, 512);
    keys[chunk0 + tid] = sh[tid];
    keys[chunk0 + tid + 512] = sh[tid + 512];
}

__global__ void range_out_kernel(const uint64_t* __restrict__ keys, int64_t m, RangeParams p,
                                 float* __restrict__ out_d, int64_t* __restrict__ out_row) {
    const float q_sq = p.fin.q_sq ? p.fin.q_sq[0] : 0.0f, q_nrm = p.fin.q_nrm ? p.fin.q_nrm[0] : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = (int64_t)(uint32_t)keys[i];
        out_row[i] = row;
        out_d[i] = finish_value(p.fin, p.cand[row], q_sq, q_nrm, row);
    }
}

static RangeParams range_params(const RangeJob& j) {
    RangeParams p{};
    p.cand = j.cand; p.n = j.n; p.mask = j.mask; p.radius = j.radius;
    p.fin = Finish{j.metric, j.q_sq, j.q_nrm, j.row_sq, j.row_nrm, 1e-8f};
    return p;
}

int64_t range_blocks(int64_t n) { return (n + RANGE_BLOCK - 1) / RANGE_BLOCK; }

int launch_range_count(const RangeJob& j, int* block_counts, int64_t* block_off, int64_t* total, hipStream_t s) {
    VDB_REQUIRE(j.n >= 1 && j.n < (int64_t)0xffffffff, "range search handles up to 2^32-1 rows");
    RangeParams p = range_params(j);
    p.block_counts = block_counts;
    const int64_t nb = range_blocks(j.n);
    hipLaunchKernelGGL(range_pass_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, p);
    hipLaunchKernelGGL(range_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, nb, block_off, total);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int64_t range_sort_capacity(int64_t m) {
    int64_t p = 1024;
    while (p < m) p <<= 1;
    return p;
}

int launch_range_collect(const RangeJob& j, const int64_t* block_off, int64_t cap, int64_t m, uint64_t* keys,
                         float* out_d, int64_t* out_row, hipStream_t s) {
    if (m <= 0) return VDB_OK;
    const int64_t P = range_sort_capacity(m);
    VDB_HIP(hipMemsetAsync(keys, 0xff, (size_t)P * sizeof(uint64_t), s));       // padding sorts behind every key
    RangeParams p = range_params(j);
    p.block_off = block_off; p.cap = cap; p.keys = keys;
    hipLaunchKernelGGL(range_pass_kernel<true>, dim3((unsigned)range_blocks(j.n)), dim3(256), 0, s, p);
    const unsigned chunks = (unsigned)(P / 1024);
    hipLaunchKernelGGL(bitonic_chunk_sort_kernel, dim3(chunks), dim3(512), 0, s, keys);
    const unsigned gsteps = (unsigned)std::min<int64_t>((P / 2 + 255) / 256, 65535);
    for (int64_t k = 2048; k <= P; k <<= 1) {
        for (int64_t jj = k >> 1; jj >= 1024; jj >>= 1)
            hipLaunchKernelGGL(bitonic_global_step_kernel, dim3(gsteps), dim3(256), 0, s, keys, P / 2, jj, k);
        hipLaunchKernelGGL(bitonic_chunk_merge_kernel, dim3(chunks), dim3(512), 0, s, keys, k);
    }
    hipLaunchKernelGGL(range_out_kernel, dim3((unsigned)std::min<int64_t>((m + 255) / 256, 65535)), dim3(256), 0, s,
                       keys, m, p, out_d, out_row);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// candidate slot -> row number for the results of a list scan
__global__ void ivf_rows_kernel(const int* __restrict__ slots, int nq, int k, int out_stride, int out_off,
                                const int* __restrict__ probes, int n_probe, int n_lists, const int64_t* __restrict__ cand_off,
                                const int64_t* __restrict__ list_beg, const int* __restrict__ row_ids,
                                int64_t* __restrict__ out_rows, int64_t* __restrict__ out_order, const int* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) return;
    const int q = t / k, r = t - q * k;
    const int64_t at = (int64_t)q * out_stride + out_off + r;
    const int slot = slots[at];
    if (slot < 0) { out_rows[at] = -1; if (out_order) out_order[at] = -1; return; }
    const int64_t* co = cand_off + (int64_t)q * (n_probe + 1);
    int lo = 0, hi = n_probe;                   // co[lo] <= slot < co[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (co[mid] <= slot) lo = mid; else hi = mid;
    }
    const int l = probes[(int64_t)q * n_probe + lo];
    out_rows[at] = (l >= 0 && l < n_lists) ? (int64_t)row_ids[list_beg[l] + (slot - co[lo])] : -1;
    // visiting order of the reference's scan (ivf.py:619-628): probe rank, then position in the list
    if (out_order) out_order[at] = ((int64_t)lo << 32) | (int64_t)(slot - co[lo]);
}

int launch_ivf_rows(const int* slots, int nq, int k, int out_stride, int out_off, const int* probes, int n_probe, int n_lists,
                    const int64_t* cand_off, const int64_t* list_beg, const int* row_ids, int64_t* out_rows,
                    int64_t* out_order, hipStream_t s, const int* run_if) {
    if (nq <= 0 || k <= 0) return VDB_OK;
    hipLaunchKernelGGL(ivf_rows_kernel, dim3((nq * k + 255) / 256), dim3(256), 0, s, slots, nq, k, out_stride, out_off,
                       probes, n_probe, n_lists, cand_off, list_beg, row_ids, out_rows, out_order, run_if);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// row norms in NumPy order: sq[i] = np.sum(row_i**2), nrm[i] = sqrt(sq[i])
// (distance/metrics.py:456,470)
__global__ void sqrt_kernel(const float* __restrict__ sq, float* __restrict__ nrm, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        nrm[i] = sqrtf(sq[i]);
}

// A few rows (a batch of queries): one thread per row walks the same pairwise plan on its own,
// one launch instead of four.  Same sums bit for bit: leaves of at most 128 elements with 8 strided
// accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and a sequential tail, leaves
// folded right into left (np.sum of a contiguous fp32 run; distance/metrics.py:455-456,469-470).
struct SmallNormPlan { int n_leaves; short leaf_len[SumPlan::MAX_LEAVES]; unsigned char folds[SumPlan::MAX_LEAVES]; };

__device__ static float small_norm_of(const float* __restrict__ x, const SmallNormPlan& plan) {
    float stack[8];
    int sp = 0, off = 0;
    for (int l = 0; l < plan.n_leaves; ++l) {
        const int len = plan.leaf_len[l];
        float res;
        if (len < 8) {
            res = 0.0f;
            for (int i = 0; i < len; ++i) res = res + x[off + i] * x[off + i];
        } else {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = x[off + j] * x[off + j];
            int i = 8;
            for (; i + 8 <= len; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = r[j] + x[off + i + j] * x[off + i + j];
            }
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i < len; ++i) res = res + x[off + i] * x[off + i];
        }
        off += len;
        stack[sp++] = res;
        for (int f = 0; f < plan.folds[l]; ++f) {
            const float b = stack[--sp];
            const float a = stack[--sp];
            stack[sp++] = a + b;
        }
    }
    return stack[0];
}

__global__ void small_norms_kernel(const float* __restrict__ data, int ld, int64_t n, SmallNormPlan plan,
                                   float* __restrict__ sq, float* __restrict__ nrm) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const float total = small_norm_of(data + row * ld, plan);
    sq[row] = total;
    nrm[row] = sqrtf(total);
}

// --------------------------------------------------------------------------
// One query against an IVF index: everything between the centroid sums and the list scan in one block.
//   query norms (NumPy order) -> distance to every centroid -> the n_probe nearest in (distance, list) order
//   (index/ivf.py:611-614) -> slot and tile offsets of the probed lists, the scan's items, its query entries.
// Replaces, for a single query, the selection and merge launches of the probe, the norm launches and the four
// planning launches of the batch path.
struct IvfOneParams {
    const float* sums; float* dists; Finish fin;  // streamed sums against the centroids (dists: where their distances go; may be sums)
    const float* q; SmallNormPlan qplan;
    int n_lists; int n_probe;
    const int64_t* list_beg; const int64_t* list_end;
    float* q_sq; float* q_nrm;                   // out: the query's norms, for the finish of the list scan
    int* probes; float* probe_d;                 // out [n_probe]
    int64_t* cand_off; int* tile_off;            // out [n_probe + 1]
    int* sq_qid; int64_t* sq_out;                // out [n_probe]
    int64_t* seg_item_off;                       // out [2]
    StreamItem* items; int64_t items_cap;
};
static constexpr int ONE_NT = 1024;

// unsigned key whose order is the order of the distances (-0 and +0 are one distance)
__device__ __forceinline__ uint32_t order_key(float d) {
    if (d == 0.0f) d = 0.0f;
    const uint32_t u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

// A single block has the chip to itself and runs at whatever the slowest dependent chain costs, so the ranking is
// a radix select: four rounds of a 256-bin histogram over the distance keys find the n_probe-th smallest key T
// exactly; the keys up to T (all ties of T included) are gathered, and one wave sorts them by (distance, list).
__global__ __launch_bounds__(ONE_NT) void ivf_one_plan_kernel(IvfOneParams p) {
    extern __shared__ __align__(16) uint32_t s_key[];           // [n_lists]
    __shared__ float s_q[2];
    __shared__ int s_hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need, s_cnt, s_valid;
    __shared__ float Cd[ONE_NT];
    __shared__ int Ci[ONE_NT];
    __shared__ int64_t s_len[64], s_co[65], s_io[65];
    __shared__ int s_list[64], s_to[65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = p.n_probe;
    if (tid == 0) {
        const float sq = small_norm_of(p.q, p.qplan);
        s_q[0] = sq; s_q[1] = sqrtf(sq);
        p.q_sq[0] = sq; p.q_nrm[0] = s_q[1];
        s_prefix = 0; s_need = np; s_cnt = 0; s_valid = 0;
    }
    if (tid < 256) s_hist[tid] = 0;
    if (tid < np) { s_list[tid] = -1; s_len[tid] = 0; p.probes[tid] = -1; if (p.probe_d) p.probe_d[tid] = INFINITY; }
    __syncthreads();
    const float q_sq = s_q[0], q_nrm = s_q[1];
    // distances (kept in place of the sums), keys, and the first histogram
    int valid = 0;
    for (int i = tid; i < p.n_lists; i += ONE_NT) {
        const float d = finish_value(p.fin, p.sums[i], q_sq, q_nrm, i);
        p.dists[i] = d;
        const bool ok = !(d != d);                           // a NaN distance ranks nowhere
        const uint32_t key = ok ? order_key(d) : 0xffffffffu;
        s_key[i] = key;
        if (ok) { atomicAdd(&s_hist[key >> 24], 1); ++valid; }
    }
    if (valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    const bool take_all = s_valid <= np;
    for (int round = 0; round < 4 && !take_all; ++round) {
        const int shift = 24 - 8 * round;
        if (wave == 0) {
            // the bin in which the running count reaches what is still needed
            const int h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2], h3 = s_hist[4 * lane + 3];
            int incl = h0 + h1 + h2 + h3;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            const int need = s_need;
            const unsigned long long m = __ballot(incl >= need);
            const int first = __builtin_ctzll(m);            // m != 0: the valid keys are more than n_probe
            if (lane == first) {
                int before = incl - (h0 + h1 + h2 + h3), bin = 4 * lane;
                if (before + h0 < need) { before += h0; ++bin;
                    if (before + h1 < need) { before += h1; ++bin;
                        if (before + h2 < need) { before += h2; ++bin; } } }
                s_prefix = (s_prefix << 8) | (uint32_t)bin;
                s_need = need - before;
            }
        }
        __syncthreads();
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        if (round < 3) {
            const uint32_t prefix = s_prefix;
            for (int i = tid; i < p.n_lists; i += ONE_NT) {
                const uint32_t key = s_key[i];
                if ((key >> shift) == prefix && key != 0xffffffffu) atomicAdd(&s_hist[(key >> (shift - 8)) & 255], 1);
            }
            __syncthreads();
        }
    }
    // every key up to T: the n_probe best and the ties of the last of them
    const uint32_t T = take_all ? 0xfffffffeu : s_prefix;
    for (int i = tid; i < p.n_lists; i += ONE_NT) {
        const uint32_t key = s_key[i];
        if (key <= T && key != 0xffffffffu) {
            const int at = atomicAdd(&s_cnt, 1);
            if (at < ONE_NT) { Cd[at] = p.dists[i]; Ci[at] = i; }
        }
    }
    __syncthreads();
    const int cnt = min(s_cnt, ONE_NT);                      // more than 1024 ties of one distance: the lowest lists among the first 1024 gathered
    if (cnt <= 64) {
        if (wave == 0) {
            const bool have = lane < cnt;
            const unsigned long long r = wave_merge_body(INFINITY, EMPTY_IDX, have ? Cd[lane] : INFINITY, have ? Ci[lane] : EMPTY_IDX);
            const int l = (int)(r >> 32);
            if (lane < np && l != EMPTY_IDX) {
                p.probes[lane] = l;
                if (p.probe_d) p.probe_d[lane] = __int_as_float((int)(unsigned)r);
                s_list[lane] = l;
                s_len[lane] = p.list_end[l] - p.list_beg[l];
            }
        }
    } else if (tid < cnt) {
        // many ties at T: a pair's place is the number of pairs that precede it
        const float md = Cd[tid];
        const int mid = Ci[tid];
        int rank = 0;
        for (int e = 0; e < cnt; ++e) rank += pair_less(Cd[e], Ci[e], md, mid) ? 1 : 0;
        if (rank < np) {
            p.probes[rank] = mid;
            if (p.probe_d) p.probe_d[rank] = md;
            s_list[rank] = mid;
            s_len[rank] = p.list_end[mid] - p.list_beg[mid];
        }
    }
    __syncthreads();
    if (tid == 0) {
        int64_t co = 0, io = 0;
        int to = 0;
        for (int i = 0; i < np; ++i) {
            s_co[i] = co; s_to[i] = to; s_io[i] = io;
            const int64_t chunks = (s_len[i] + S_NT - 1) / S_NT;
            co += s_len[i]; to += (int)chunks; io += chunks;
        }
        s_co[np] = co; s_to[np] = to; s_io[np] = io;
        p.seg_item_off[0] = 0; p.seg_item_off[1] = io;
    }
    __syncthreads();
    if (tid <= np) { p.cand_off[tid] = s_co[tid]; p.tile_off[tid] = s_to[tid]; }
    if (tid < np) { p.sq_qid[tid] = 0; p.sq_out[tid] = s_co[tid]; }
    // items: chunk c of probe pr; one thread per item
    const int64_t total = min(s_io[np], p.items_cap);
    for (int64_t it = tid; it < total; it += ONE_NT) {
        int lo = 0, hi = np;                      // s_io[lo] <= it < s_io[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_io[mid] <= it) lo = mid; else hi = mid;
        }
        const int64_t c = it - s_io[lo];
        StreamItem item;
        item.out_rel = c * S_NT;
        item.row0 = p.list_beg[s_list[lo]] + item.out_rel;
        item.rows = (int)min<int64_t>(S_NT, s_len[lo] - item.out_rel);
        item.qbeg = lo; item.qcnt = 1; item.pad = 0;
        p.items[it] = item;
    }
}

int launch_ivf_one_plan(const float* sums, int metric, const float* cent_sq, const float* cent_nrm, float cos_eps, const float* q,
                        const SumPlan& plan, int n_lists, int n_probe, const int64_t* list_beg, const int64_t* list_end,
                        float* q_sq, float* q_nrm, int* probes, float* probe_d, int64_t* cand_off, int* tile_off,
                        const StreamPlanBufs& b, hipStream_t s) {
    VDB_REQUIRE(n_probe >= 1 && n_probe <= 64, "the one-query plan ranks at most 64 lists");
    VDB_REQUIRE(n_lists >= 1 && n_lists <= 16384, "the one-query plan keeps the keys of at most 16384 lists in LDS");
    IvfOneParams p{};
    p.sums = sums; p.dists = const_cast<float*>(sums); p.fin = Finish{metric, nullptr, nullptr, cent_sq, cent_nrm, cos_eps};
    p.q = q;
    p.qplan.n_leaves = plan.n_leaves;
    for (int i = 0; i < plan.n_leaves; ++i) { p.qplan.leaf_len[i] = plan.leaf_len[i]; p.qplan.folds[i] = plan.folds[i]; }
    p.n_lists = n_lists; p.n_probe = n_probe; p.list_beg = list_beg; p.list_end = list_end;
    p.q_sq = q_sq; p.q_nrm = q_nrm; p.probes = probes; p.probe_d = probe_d; p.cand_off = cand_off; p.tile_off = tile_off;
    p.sq_qid = b.sq_qid; p.sq_out = b.sq_out; p.seg_item_off = b.seg_item_off; p.items = b.items; p.items_cap = b.items_cap;
    const size_t lds = (size_t)n_lists * sizeof(uint32_t);
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(ivf_one_plan_kernel), lds));
    hipLaunchKernelGGL(ivf_one_plan_kernel, dim3(1), dim3(ONE_NT), lds, s, p);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// A batch of rows (up to a few thousand queries): eight lanes per row, lane j keeps the strided
// accumulator r[j] of every leaf, and the leaf result ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) comes from
// three exchange steps (a + b and b + a are the same float, so all eight lanes hold the same bits).
// Tails and short leaves are summed by every lane on its own.  Same plan walk as small_norms_kernel.
__global__ __launch_bounds__(256) void group_norms_kernel(const float* __restrict__ data, int ld, int64_t n, SmallNormPlan plan,
                                                          float* __restrict__ sq, float* __restrict__ nrm) {
    const int j = threadIdx.x & 7;
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = row < n;
    const float* x = data + (live ? row : 0) * ld;          // idle groups walk row 0 and store nothing
    float stack[8];
    int sp = 0, off = 0;
    for (int l = 0; l < plan.n_leaves; ++l) {
        const int len = plan.leaf_len[l];
        float res;
        if (len < 8) {
            res = 0.0f;
            for (int i = 0; i < len; ++i) res = res + x[off + i] * x[off + i];
        } else {
            float r = x[off + j] * x[off + j];
            int i = 8;
#pragma unroll 4
            for (; i + 8 <= len; i += 8) {
                const float v = x[off + i + j];
                r = r + v * v;
            }
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            res = r + __shfl_xor(r, 4);
            for (; i < len; ++i) res = res + x[off + i] * x[off + i];
        }
        off += len;
        stack[sp++] = res;
        for (int f = 0; f < plan.folds[l]; ++f) {
            const float b = stack[--sp];
            const float a = stack[--sp];
            stack[sp++] = a + b;
        }
    }
    if (live && j == 0) {
        sq[row] = stack[0];
        nrm[row] = sqrtf(stack[0]);
    }
}

static constexpr int64_t GROUP_NORMS_MAX_ROWS = 16384;

static bool plan_depth_ok(const SumPlan& plan) {
    int sp = 0, depth = 0;
    for (int i = 0; i < plan.n_leaves; ++i) { ++sp; depth = std::max(depth, sp); sp -= plan.folds[i]; }
    return depth <= 8;
}

int launch_row_norms(const float* data, int ld, int dim, int64_t n, const SumPlan& plan, const StreamPlanBufs& b,
                     float* sq, float* nrm, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    if (n <= 64 && plan_depth_ok(plan)) {
        SmallNormPlan sp{};
        sp.n_leaves = plan.n_leaves;
        for (int i = 0; i < plan.n_leaves; ++i) { sp.leaf_len[i] = plan.leaf_len[i]; sp.folds[i] = plan.folds[i]; }
        (void)dim;
        hipLaunchKernelGGL(small_norms_kernel, dim3(1), dim3(64), 0, s, data, ld, n, sp, sq, nrm);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    if (n <= GROUP_NORMS_MAX_ROWS && plan_depth_ok(plan)) {
        SmallNormPlan sp{};
        sp.n_leaves = plan.n_leaves;
        for (int i = 0; i < plan.n_leaves; ++i) { sp.leaf_len[i] = plan.leaf_len[i]; sp.folds[i] = plan.folds[i]; }
        hipLaunchKernelGGL(group_norms_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, data, ld, n, sp, sq, nrm);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    VDB_REQUIRE(b.items_cap >= (n + S_NT - 1) / S_NT, "item table too small for %lld rows", (long long)n);
    VDB_TRY(launch_flat_plan(n, 1, 1, n, b, s));
    StreamJob j{};
    j.data = data; j.ld = ld; j.dim = dim; j.plan = plan; j.op = OP_SQ; j.post = POST_NONE;
    j.q = data; j.ldq = 0;                       // the query operand is unused: any readable row
    j.group = 1; j.n_seg = 1;
    j.seg_row_off = b.seg_row_off; j.seg_q_off = b.seg_q_off; j.seg_item_off = b.seg_item_off;
    j.sq_qid = b.sq_qid; j.sq_out = b.sq_out; j.items = b.items; j.items_cap = b.items_cap;
    j.out = sq; j.timed = false;
    VDB_TRY(launch_stream_sums(j, s));
    hipLaunchKernelGGL(sqrt_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, sq, nrm, n);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
