// Batched distance + top-k for the Flat index on the fp32 matrix cores.
//
// Computes, for a tile of 128 queries against a slab of database rows, the dot
// products c.q with v_mfma_f32_32x32x2_f32 (exact fp32 fma chains), turns each
// into the final distance value exactly as the reference's query_distances
// does (distance/metrics.py:453-484: |q|^2+|c|^2-2c.q -> clamp -> sqrt;
// c.q/(|q||c|) -> clip -> 1-s; -(c.q)) and keeps the k best per query without
// ever writing the nq x n matrix (index/flat.py:331,354-362 materialise it).
//
// Orientation: the MFMA "A" operand is the database tile, "B" the query tile,
// so in the 32x32 accumulator a lane owns ONE query (column = lane & 31) and
// 16 database rows.  Selection is therefore lane-local: every thread keeps a
// sorted list of its k best (distance, row) in LDS and filters new values
// against a threshold shared with the one other lane that owns the same query.
//
// The per-value work in the epilogue is a cheap conservative filter in the
// dot-product domain (one or two VALU operations and a compare); the exact
// distance (division, sqrt) and the list insertion run only for the values
// that passed.  The filter threshold can be
// warm-started from the k-th best distance of a strided sample of the rows
// (any subset's k-th best bounds the k-th best of the whole set from above).
//
// Block = 256 threads = 4 waves; wave w owns queries [32w, 32w+32) of the
// tile and all 128 database rows of the current tile (4 accumulators of
// 32x32).  K is walked in steps of 32 staged through LDS (row stride 36
// floats: conflict-free ds_read_b128 fragment reads).
#include "flat_math.h"

namespace vdb {

struct GemmParams {
    const void* rows; int64_t n; int ld_bytes;     // row pitch in bytes (multiple of 16); fp32 or bf16 elements
    const float* row_sq; const float* row_nrm;
    const void* q; int nq;                         // same element type and pitch as the rows
    const float* q_sq; const float* q_nrm;
    const uint32_t* mask;
    const float* lb_d; const int64_t* lb_k;
    const float* thr_init;          // optional per-query upper bound of the k-th best distance
    int k;
    int n_tiles;                    // tiles visited by this launch
    int tile_stride;                // visited tile t is database tile t * tile_stride
    int n_slabs, tiles_per_slab, n_qtiles;
    float* part_d; int* part_i; int n_cand;
    float* dist_out;
    const int* run_if;              // optional device flag: the launch does nothing while it is zero
    const int* q_count;             // optional device count: only the first *q_count queries are there
    float cos_eps;                  // cosine: norms below it count as 1 (1e-8 for searches)
};

// SAMPLE only gives the strided warm-up pass a kernel name of its own, so that
// profiles list it separately from the full pass.
// BF16: rows and queries are bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulation); the LDS image,
// the staging and the fragment addresses are the same bytes as for fp32.
template <int METRIC, bool TOPK, bool SAMPLE, bool BF16>
__global__ __launch_bounds__(NT, 2) void flat_gemm_kernel(GemmParams p) {
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                   // [TM][LW]  database rows
    float* Bs = As + TM * LW;           // [TQ][LW]  queries
    float* Rs = Bs + TQ * LW;           // [TM]      per-row term (|c|^2 or |c|)
    float* Ld = Rs + TM;                // [k][NT]   per-thread best distances
    int* Li = reinterpret_cast<int*>(Ld + (TOPK ? p.k : 0) * NT);

    if (p.run_if && *p.run_if == 0) return;
    if (p.q_count) {
        p.nq = min(p.nq, *p.q_count);
        if (p.nq <= 0) return;
    }
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
        // few slabs (a small row set, e.g. centroids): every block is live and the
        // query tiles spread over all XCDs; the rows fit every L2 anyway
        qt = blockIdx.x / p.n_slabs;
        slab = blockIdx.x - qt * p.n_slabs;
    }
    const int tile0 = slab * p.tiles_per_slab;
    const int tile1 = min(tile0 + p.tiles_per_slab, p.n_tiles);
    if (qt * TQ >= p.nq) return;                 // a query tile beyond the queries that are there (q_count)

    const int my_q = qt * TQ + wave * 32 + col;
    const bool q_ok = my_q < p.nq;
    const int my_q_ld = min(my_q, p.nq - 1);
    float q_term = 0.0f;
    if (METRIC == VDB_COSINE) {
        q_term = p.q_nrm[my_q_ld];
        // metrics.py:472-473: all ones; with a caller's eps (distance matrix only) the norm counts as 1, :360-361
        if (q_term < p.cos_eps) q_term = (TOPK || p.cos_eps <= 1e-8f) ? INFINITY : 1.0f;
    } else if (METRIC != VDB_DOT) {
        q_term = p.q_sq[my_q_ld];
    }
    float lbd = -INFINITY;
    int64_t lbk = -1;
    if (TOPK && p.lb_d) { lbd = p.lb_d[my_q_ld]; lbk = p.lb_k[my_q_ld]; }

    float own_kth = INFINITY, thr = INFINITY;
    if (TOPK) {
        list_init(Ld, Li, p.k, tid);
        if (p.thr_init) thr = p.thr_init[my_q_ld];
    }
    float bound = filter_bound<METRIC>(thr, q_term);

    const int ksteps = (p.ld_bytes + TKB - 1) / TKB;
    const int srow = tid >> 3;          // staging: row within a 32-row pass
    const int skc = (tid & 7) * 4;      // staging: dword offset inside the K step (16-byte pieces)
    const int64_t q_base = (int64_t)qt * TQ;

    // staging addresses: a wave-uniform base (tile, K step) plus a 32-bit per-thread
    // offset (row inside the tile, float4 inside the K step), so a K step costs
    // eight loads and no per-thread address arithmetic
    const bool k_tail = (p.ld_bytes % TKB) != 0;
    const int ldw = p.ld_bytes >> 2;    // row pitch in dwords
    uint32_t offb[4], offa[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int64_t gq = min(q_base + ps * 32 + srow, (int64_t)p.nq - 1) - q_base;
        offb[ps] = (uint32_t)(gq * ldw + skc);
        offa[ps] = (uint32_t)((ps * 32 + srow) * ldw + skc);
    }
    const float* rows_w = reinterpret_cast<const float*>(p.rows);      // dword views of the operands
    const float* qbase_ptr = reinterpret_cast<const float*>(p.q) + q_base * ldw;
    float4 ra[4], rb[4];
    auto issue_loads = [&](int tile, int ks) {
        const int64_t rbase = (int64_t)tile * p.tile_stride * TM;
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
    };

    if (tile0 < tile1) issue_loads(tile0, 0);

    for (int tile = tile0; tile < tile1; ++tile) {
        const int64_t rbase = (int64_t)tile * p.tile_stride * TM;
        f32x16 acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][e] = 0.0f;

        for (int ks = 0; ks < ksteps; ++ks) {
            __syncthreads();            // everyone is done reading the previous step (and Rs)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int r = ps * 32 + srow;
                *reinterpret_cast<float4*>(As + r * LW + skc) = ra[ps];
                *reinterpret_cast<float4*>(Bs + r * LW + skc) = rb[ps];
            }
            if (ks == 0 && METRIC != VDB_DOT && tid < TM) {
                const int64_t gr = min(rbase + tid, p.n - 1);
                float t = (METRIC == VDB_COSINE) ? p.row_nrm[gr] : p.row_sq[gr];
                if (METRIC == VDB_COSINE && t < p.cos_eps) t = 1.0f;    // metrics.py:476
                Rs[tid] = t;
            }
            __syncthreads();
            // prefetch the next step (possibly of the next tile) while computing
            if (ks + 1 < ksteps) issue_loads(tile, ks + 1);
            else if (tile + 1 < tile1) issue_loads(tile + 1, 0);

            // all fragment reads of this K step go out first; the MFMAs then start as
            // soon as their own operands have landed while the rest are still in flight
            const float* bq = Bs + (wave * 32 + col) * LW + half * 4;
            const float* ar = As + col * LW + half * 4;
            float4 bf[4], af[4][4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                bf[ch] = *reinterpret_cast<const float4*>(bq + ch * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s) af[ch][s] = *reinterpret_cast<const float4*>(ar + s * 32 * LW + ch * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (BF16) {
                        // the 16 bytes are 8 consecutive bf16 of the K step: k = 16*ch + 8*half + 0..7
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ch][s]),
                                                                         __builtin_bit_cast(bf16x8, bf[ch]), acc[s], 0, 0, 0);
                    } else {
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ch][s].x, bf[ch].x, acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ch][s].y, bf[ch].y, acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ch][s].z, bf[ch].z, acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ch][s].w, bf[ch].w, acc[s], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue: lane owns query `my_q`, rows rbase + 32s + 8g + 4*half + j
        if (TOPK) {
            const bool tail = rbase + TM > p.n;
            const bool masked = tail || p.mask != nullptr;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint32_t bits = 0xffffffffu;
                if (masked) {
                    const int64_t r0 = rbase + s * 32;
                    if (p.mask) bits = (r0 < p.n) ? p.mask[r0 >> 5] : 0u;
                    const int64_t left = p.n - r0;
                    if (left < 32) bits = left <= 0 ? 0u : (bits & ((1u << left) - 1u));
                    bits >>= (4 * half);
                }
                float4 t[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    t[g] = make_float4(0, 0, 0, 0);
                    if (METRIC != VDB_DOT) t[g] = *reinterpret_cast<const float4*>(Rs + s * 32 + 8 * g + 4 * half);
                }
                // cheap filter over the 16 values of this accumulator -> per-lane bit set
                uint32_t hits = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float rts[4] = {t[g].x, t[g].y, t[g].z, t[g].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool pass = filter_pass<METRIC>(acc[s][g * 4 + j], rts[j], q_term, bound);
                        if (masked) pass = pass && ((bits >> (8 * g + j)) & 1u);
                        hits |= pass ? (1u << (g * 4 + j)) : 0u;
                    }
                }
                // exact distance and insertion, only for the values that passed
                if (hits) {
                    float* ld = Ld + tid;
                    int* li = Li + tid;
                    do {
                        const int e = __ffs(hits) - 1;
                        hits &= hits - 1;
                        const int off = 8 * (e >> 2) + (e & 3) + 4 * half;     // row inside the 32-row sub-tile
                        const float rt = (METRIC != VDB_DOT) ? Rs[s * 32 + off] : 0.0f;
                        const float d = finish<METRIC>(pick16(acc[s], e), rt, q_term);
                        const int row = (int)(rbase + s * 32 + off);
                        const bool above = (d > lbd) || (d == lbd && (int64_t)row > lbk);   // exclusive bound of a follow-up pass
                        if (d <= thr && above && d < own_kth) {
                            // rows reach a lane in ascending order: an equal distance stays behind
                            int pos = p.k - 1;
                            while (pos > 0) {
                                const float prev = ld[(pos - 1) * NT];
                                if (!(prev > d)) break;
                                ld[pos * NT] = prev;
                                li[pos * NT] = li[(pos - 1) * NT];
                                --pos;
                            }
                            ld[pos * NT] = d;
                            li[pos * NT] = row;
                            own_kth = ld[(p.k - 1) * NT];
                            thr = fminf(thr, own_kth);
                        }
                    } while (hits);
                    bound = filter_bound<METRIC>(thr, q_term);
                }
            }
            // the other half-wave lane with the same query may hold a tighter bound
            const float shared = fminf(thr, __shfl_xor(thr, 32));
            if (shared < thr) { thr = shared; bound = filter_bound<METRIC>(thr, q_term); }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 rt = make_float4(0, 0, 0, 0);
                    if (METRIC != VDB_DOT) rt = *reinterpret_cast<const float4*>(Rs + s * 32 + 8 * g + 4 * half);
                    const float rts[4] = {rt.x, rt.y, rt.z, rt.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t row = rbase + s * 32 + 8 * g + 4 * half + j;
                        if (q_ok && row < p.n)
                            p.dist_out[(int64_t)my_q * p.n + row] = finish<METRIC>(acc[s][g * 4 + j], rts[j], q_term);
                    }
                }
        }
    }

    if (TOPK && q_ok) {
        float* od = p.part_d + ((int64_t)my_q * p.n_cand + slab * 2 + half) * p.k;
        int* oi = p.part_i + ((int64_t)my_q * p.n_cand + slab * 2 + half) * p.k;
        for (int i = 0; i < p.k; ++i) { od[i] = Ld[i * NT + tid]; oi[i] = Li[i * NT + tid]; }
    }
}

// --------------------------------------------------------------------------

int flat_gemm_num_cand(int64_t n, int nq, int k, int* n_cand) {
    int nqt, ns, tps;
    plan_grid((int)((n + TM - 1) / TM), nq, ctx().num_cus, &nqt, &ns, &tps);
    *n_cand = ns * 2;
    return VDB_OK;
}

int flat_gemm_query_tile() { return TQ; }

int flat_gemm_num_cand_tiles(int n_tiles, int nq, int* n_cand) {
    int nqt, ns, tps;
    plan_grid(n_tiles, nq, ctx().num_cus, &nqt, &ns, &tps);
    *n_cand = ns * 2;
    return VDB_OK;
}

int flat_gemm_sample_plan(int64_t n, int* sample_tiles, int* tile_stride) {
    // a strided sample of ~1/64 of the tiles (at least 32 of them) once the set has 512 tiles:
    // the sample pass then costs a few percent of the full pass, and the bound it yields lets
    // the full pass run as the candidate-queue kernel
    const int n_tiles = (int)((n + TM - 1) / TM);
    static const int min_tiles = getenv("VDB_SAMPLE_MIN_TILES") ? atoi(getenv("VDB_SAMPLE_MIN_TILES")) : 512;
    if (n_tiles < min_tiles) { *sample_tiles = 0; *tile_stride = 1; return VDB_OK; }
    int want = std::max(32, n_tiles / 64);
    *tile_stride = n_tiles / want;
    *sample_tiles = want;
    return VDB_OK;
}

template <bool TOPK, bool BF16>
static int launch_gemm(int metric, GemmParams& p, hipStream_t s) {
    plan_grid(p.n_tiles, p.nq, ctx().num_cus, &p.n_qtiles, &p.n_slabs, &p.tiles_per_slab);
    const int groups = (p.n_slabs + 7) / 8;
    const dim3 grid(p.n_slabs >= 8 ? groups * 8 * p.n_qtiles : p.n_slabs * p.n_qtiles);
    const size_t lds = (size_t)(2 * TM * LW + TM) * sizeof(float) + (TOPK ? (size_t)p.k * NT * 8 : 0);
    KernelTimer timer(s, p.tile_stride == 1 && p.run_if == nullptr && p.q_count == nullptr);
#define VDB_LAUNCH_GEMM(M)                                                                     \
    case M: {                                                                                  \
        auto kern = (TOPK && p.tile_stride != 1) ? flat_gemm_kernel<M, TOPK, TOPK, BF16> : flat_gemm_kernel<M, TOPK, false, BF16>; \
        VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(kern),                       \
                                    lds));    \
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);                                   \
        break;                                                                                 \
    }
    switch (metric) {
        VDB_LAUNCH_GEMM(VDB_EUCLIDEAN)
        VDB_LAUNCH_GEMM(VDB_EUCLIDEAN_SQ)
        VDB_LAUNCH_GEMM(VDB_COSINE)
        VDB_LAUNCH_GEMM(VDB_DOT)
        default:
            set_error("metric %d has no matrix-core path", metric);
            return VDB_ERR_INVALID;
    }
#undef VDB_LAUNCH_GEMM
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_flat_gemm_topk(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm,
                          int nq, int k, const uint32_t* mask, const float* lb_d, const int64_t* lb_k,
                          const float* thr_init, int sample_tiles, int tile_stride,
                          float* part_d, int* part_i, int n_cand, hipStream_t s, const int* run_if, const int* q_count) {
    VDB_REQUIRE(k >= 1 && k <= KMAX_FUSED, "fused selection supports 1 <= k <= %d, got %d", KMAX_FUSED, k);
    VDB_REQUIRE(r.n >= 1 && nq >= 1, "empty search");
    GemmParams p{};
    p.rows = r.bf16 ? (const void*)r.data16 : (const void*)r.data; p.n = r.n;
    p.ld_bytes = r.bf16 ? r.ld16 * 2 : r.ld * 4; p.row_sq = r.sq; p.row_nrm = r.nrm;
    p.q = q; p.nq = nq; p.q_sq = q_sq; p.q_nrm = q_nrm;
    p.mask = mask; p.lb_d = lb_d; p.lb_k = lb_k; p.thr_init = thr_init; p.k = k;
    p.n_tiles = sample_tiles > 0 ? sample_tiles : (int)((r.n + TM - 1) / TM);
    p.tile_stride = sample_tiles > 0 ? tile_stride : 1;
    p.part_d = part_d; p.part_i = part_i; p.n_cand = n_cand; p.dist_out = nullptr; p.run_if = run_if; p.q_count = q_count;
    p.cos_eps = 1e-8f;
    return r.bf16 ? launch_gemm<true, true>(metric, p, s) : launch_gemm<true, false>(metric, p, s);
}

int launch_flat_gemm_dist(const Rows& r, int metric, const void* q, const float* q_sq, const float* q_nrm,
                          int nq, float* out, hipStream_t s) {
    VDB_REQUIRE(r.n >= 1 && nq >= 1, "empty distance computation");
    GemmParams p{};
    p.rows = r.bf16 ? (const void*)r.data16 : (const void*)r.data; p.n = r.n;
    p.ld_bytes = r.bf16 ? r.ld16 * 2 : r.ld * 4; p.row_sq = r.sq; p.row_nrm = r.nrm;
    p.q = q; p.nq = nq; p.q_sq = q_sq; p.q_nrm = q_nrm;
    p.k = 0; p.dist_out = out; p.cos_eps = r.cos_eps;
    p.n_tiles = (int)((r.n + TM - 1) / TM);
    p.tile_stride = 1;
    return r.bf16 ? launch_gemm<false, true>(metric, p, s) : launch_gemm<false, false>(metric, p, s);
}

// ---------------------------------------------------------------- searching overflowed queries again, alone
__global__ __launch_bounds__(1024) void redo_gather_kernel(const int* __restrict__ flag, const int* __restrict__ mark, int nq,
                                                           const float* __restrict__ q, int ldq, const float* __restrict__ q_sq,
                                                           const float* __restrict__ q_nrm, const float* __restrict__ thr,
                                                           int* __restrict__ list, int* __restrict__ count,
                                                           float* __restrict__ q_out, float* __restrict__ q_sq_out,
                                                           float* __restrict__ q_nrm_out, float* __restrict__ thr_out,
                                                           int limit, int* __restrict__ all_flag) {
    __shared__ int wave_sum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (*flag == 0) {
        if (tid == 0) { *count = 0; *all_flag = 0; }
        return;
    }
    if (tid == 0) base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {                 // ascending query order
        const int qi = q0 + tid;
        const bool hit = qi < nq && mark[qi] != 0;
        const unsigned long long votes = __ballot(hit);
        if (lane == 0) wave_sum[wave] = __popcll(votes);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (hit) list[before + __popcll(votes & ((1ull << lane) - 1ull))] = qi;
        __syncthreads();
        if (tid == 0) {
            int t = base;
            for (int w = 0; w < 16; ++w) t += wave_sum[w];
            base = t;
        }
        __syncthreads();
    }
    int n = base;
    if (n > limit || n == 0) {                                // too many for one query tile (or a flag without a mark): the whole batch again
        if (tid == 0) { *count = 0; *all_flag = 1; }
        return;
    }
    if (tid == 0) { *count = n; *all_flag = 0; }
    __syncthreads();                                          // the list is written (same block: LDS barrier + global writes)
    __threadfence_block();
    for (int j = tid; j < n; j += 1024) {
        const int src = list[j];
        q_sq_out[j] = q_sq[src];
        q_nrm_out[j] = q_nrm[src];
        thr_out[j] = thr ? thr[src] : INFINITY;
    }
    for (int64_t e = tid; e < (int64_t)n * ldq; e += 1024) {
        const int j = (int)(e / ldq);
        q_out[e] = q[(int64_t)list[j] * ldq + (e - (int64_t)j * ldq)];
    }
}

__global__ void redo_scatter_kernel(const int* __restrict__ list, const int* __restrict__ count, int k,
                                    const float* __restrict__ d, const int64_t* __restrict__ i64, const int* __restrict__ i32,
                                    float* __restrict__ out_d, int64_t* __restrict__ out_i, int* __restrict__ out_i32) {
    const int n = *count;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * k; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e / k);
        const int64_t to = (int64_t)list[j] * k + (e - (int64_t)j * k);
        if (out_d) out_d[to] = d[e];
        if (out_i) out_i[to] = i64[e];
        if (out_i32) out_i32[to] = i32[e];
    }
}

int launch_redo_gather(const int* flag, const int* ovf_mark, int nq, const float* q, int ldq, const float* q_sq, const float* q_nrm,
                       const float* thr, int* list, int* count, float* q_out, float* q_sq_out, float* q_nrm_out, float* thr_out,
                       int limit, int* all_flag, hipStream_t s) {
    hipLaunchKernelGGL(redo_gather_kernel, dim3(1), dim3(1024), 0, s, flag, ovf_mark, nq, q, ldq, q_sq, q_nrm, thr, list, count,
                       q_out, q_sq_out, q_nrm_out, thr_out, limit, all_flag);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

__global__ __launch_bounds__(256) void redo_gather_list_kernel(const int* __restrict__ list, const int* __restrict__ count, int lim,
                                                               const float* __restrict__ q, int ldq, const float* __restrict__ q_sq,
                                                               const float* __restrict__ q_nrm, float* __restrict__ q_out,
                                                               float* __restrict__ q_sq_out, float* __restrict__ q_nrm_out,
                                                               int* __restrict__ q_count, int* __restrict__ all_flag) {
    const int wanted = *count;
    const bool fits = wanted <= lim;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *q_count = fits ? wanted : 0; *all_flag = fits ? 0 : 1; }
    if (!fits) return;
    for (int j = blockIdx.x; j < wanted; j += gridDim.x) {
        const int src = list[j];
        if (threadIdx.x == 0) {
            q_sq_out[j] = q_sq[src];
            if (q_nrm) q_nrm_out[j] = q_nrm[src];
        }
        for (int e = threadIdx.x; e < ldq; e += 256) q_out[(int64_t)j * ldq + e] = q[(int64_t)src * ldq + e];
    }
}

int launch_redo_gather_list(const int* list, const int* count, int lim, const float* q, int ldq, const float* q_sq,
                            const float* q_nrm, float* q_out, float* q_sq_out, float* q_nrm_out, int* q_count, int* all_flag,
                            hipStream_t s) {
    hipLaunchKernelGGL(redo_gather_list_kernel, dim3(std::min(lim, 512)), dim3(256), 0, s, list, count, lim, q, ldq, q_sq, q_nrm,
                       q_out, q_sq_out, q_nrm_out, q_count, all_flag);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_redo_scatter(const int* list, const int* count, int nq, int k, const float* d, const int64_t* i64, const int* i32,
                        float* out_d, int64_t* out_i, int* out_i32, hipStream_t s) {
    const int blocks = (int)std::min<int64_t>(((int64_t)nq * k + 255) / 256, 256);
    hipLaunchKernelGGL(redo_scatter_kernel, dim3(blocks), dim3(256), 0, s, list, count, k, d, i64, i32, out_d, out_i, out_i32);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
