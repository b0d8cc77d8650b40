// Batched IVF list scan on the matrix cores, behind an exact recomputation.
//
// The reference scans the probed lists with the per-pair scalar distance (index/ivf.py:619-628: for Euclidean
// sqrt(sum((q - c)^2)), distance/metrics.py:43) and sorts the candidates (:631).  What a batch needs from the scan
// is each query's k best; everything else of the ~n_probe * n / n_lists distances per query is thrown away.  So:
//
//   1. bound pass  (ivf_bound_kernel): over the first 2048 rows a query visits (its nearest probed lists), the
//      approximate key  |q|^2 + |c|^2 - 2 bf16(q).bf16(c)  of every row on v_mfma_f32_32x32x16_bf16, reduced to 128
//      minima over disjoint row subsets; the k-th smallest of them is not below the k-th smallest key of those rows.
//   2. ivf_threshold_kernel: that value, widened by what the approximate key can differ from the reference's
//      float32 sum (bf16 rounding of both operands, the float32 roundings of both forms), and widened again for
//      the test itself: every row that can be among the reference's k best has a key within the threshold.
//   3. scan pass   (ivf_approx_kernel): the same keys for every (probed list row, query) pair, list-major
//      (a list chunk is read once for up to 32 queries that probe it); pairs within the threshold (and passing the
//      filter bit) are kept with their keys, per wave; ivf_scatter_kernel hands them to the queries' queues.
//   4. ivf_finish_kernel, one block per query: the k-th smallest key of the queue, A_k, bounds the reference's k-th
//      best sum as the bound pass's value did (k rows have keys within it), only much tighter, so the same widening
//      gives a second threshold, and only the pairs within that one (a couple of dozen of several hundred) are
//      recomputed in the reference's own arithmetic (element-wise float32 q - c, square, NumPy's pairwise summation
//      order, sqrtf); the k best of those under (distance, candidate slot) = (distance, visiting order of
//      index/ivf.py:619-628) are the result.
//
// Rows, order and distance bits equal those of the scalar scan, because every returned value is computed by the same
// float32 operations on the same float32 rows; the matrix cores only decide which pairs get computed.  A queue that
// overflows raises a device flag, and the streaming scan repeats the batch behind that flag.
#include "vdb_internal.h"
#include "vdb_device.h"
#include "flat_math.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace vdb {

static constexpr int IA_NT = 256;          // threads per block: four independent waves
static constexpr int IA_SETS = 4;          // scan pass: register sets of one 32-row tile per wave (IA_SETS - 1 tiles in flight)
static constexpr int IA_LEN_STRIDE = 32;   // ints between the queue lengths of two queries: one 128-byte line each (they are hit by atomics)
static constexpr int IA_RUN = 4;           // scan pass: consecutive items (256-row chunks) one wave takes at a time
static constexpr int IB_UNITS = 8;         // bound pass: 256-row units of a query's visiting order (2048 rows)
static constexpr int IA_SUBSETS = IB_UNITS * 16;   // disjoint row subsets (32-row tile x half) whose key minima bound the k-th best

struct IvfApproxParams {
    const uint16_t* rows16; int ld16;      // bf16 copies of the stored rows in the tiled layout (tiled_at), ld16 columns each
    int n_chunks;                          // column chunks of KS * 16 elements per row
    const float* row_sq;                   // |c|^2 of the float32 rows, by position
    const uint16_t* q16;                   // bf16 copies of the queries, [nq][ld16]
    const float* q_sq;                     // |q|^2 of the float32 queries
    const StreamItem* items; const int64_t* seg_item_off; int n_seg; int64_t items_cap;
    const int* sq_qid; const int64_t* sq_out;                         // per (list, query): query, first candidate slot
    int64_t cand_stride;                   // sq_out = query * cand_stride + slot of the list's first row
    const float* thr;                      // per query: pairs with key <= thr are kept
    int4* seg; int* seg_len; int seg_cap;  // per wave of the launch: its kept pairs (query, slot, position, key), and how many
    int* overflow;
    const int* row_ids; const uint32_t* mask;
};

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef int i8v_u __attribute__((ext_vector_type(8), aligned(4)));       // eight dwords at any dword address

template <class T> __device__ __forceinline__ T uniform_load(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

// The bf16 copy is laid out for the matrix cores, not row by row: positions go in tiles of 32, and inside a tile the
// 16-byte pieces (8 columns) of the 32 rows are interleaved, piece c of row r at element ((tile * pieces + c) * 32 + r) * 8.
// A wave loads piece c of 32 consecutive rows as 512 contiguous bytes.
__device__ __forceinline__ int64_t tiled_at(int64_t pos, int piece, int pieces) {
    return (((pos >> 5) * pieces + piece) * 32 + (pos & 31)) * 8;
}

// One wave per run of consecutive items = (rows of one list, up to 32 of the queries that probe the list); a run is up to
// IA_RUN 256-row chunks that follow each other in memory and share their query group.
// A operand: 32 rows, lane (row = lane % 32, half = lane / 32) loads pieces 2 j + half (j = 0 .. KS - 1) of its row's
// current column chunk straight from global memory: one load instruction reads 1 KiB in two (or, for a run that does
// not start on a tile boundary, four) contiguous stretches.  k step j multiplies piece 2 j of one half with piece 2 j + 1
// of the other, i.e. the columns are visited in a permuted order, the same for rows and queries (the sum does not care).
// B operand: the queries (row-major) in the same piece order.  The 32 x 32 accumulator gives lane (query = lane % 32)
// sixteen rows: 8 g + 4 half + i.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // four floats at any dword address

struct ItemHeader { int64_t row0, out_rel; int rows, qbeg, qcnt; };
__device__ __forceinline__ ItemHeader load_header(const StreamItem* items, int64_t item) {
    const i8v w = uniform_load(reinterpret_cast<const i8v*>(items + item));
    ItemHeader it;
    it.row0 = ((int64_t)(uint32_t)w[1] << 32) | (uint32_t)w[0];
    it.out_rel = ((int64_t)(uint32_t)w[3] << 32) | (uint32_t)w[2];
    it.rows = w[4]; it.qbeg = w[5]; it.qcnt = w[6];
    return it;
}

template <int KS, bool MULTI>
__global__ __launch_bounds__(IA_NT, 2) void ivf_approx_kernel(IvfApproxParams p) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, h = lane >> 5;
    const int64_t n_waves = (int64_t)gridDim.x * (IA_NT / 64);
    // (the wave number through readfirstlane: what derives from it, items and row addresses, stays in scalar registers)
    const int64_t wave0 = (int64_t)blockIdx.x * (IA_NT / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t total = min(uniform_load(p.seg_item_off + p.n_seg), p.items_cap);
    constexpr int CE = KS * 16;                    // elements per column chunk
    const int pieces = p.ld16 >> 3;
    int4* my_seg = p.seg + wave0 * p.seg_cap;
    int seg_n = 0;
    for (int64_t unit = wave0 * IA_RUN; unit < total; unit += n_waves * IA_RUN) {
        const int64_t unit_end = min(unit + IA_RUN, total);
        int64_t item = unit;
        ItemHeader next = load_header(p.items, item);
        while (item < unit_end) {
            // the run: this item and the ones behind it that continue its rows for the same queries
            const ItemHeader first = next;
            int rows = first.rows;
            ++item;
            while (item < unit_end) {
                next = load_header(p.items, item);
                if (next.qbeg != first.qbeg || next.row0 != first.row0 + rows || (rows & 255) != 0) break;
                rows += next.rows;
                ++item;
            }
            const int64_t row0 = first.row0;
            const bool live = n < first.qcnt;
            const int qslot = first.qbeg + (live ? n : 0);
            const int qid = p.sq_qid[qslot];
            const int slot0 = (int)(p.sq_out[qslot] - (int64_t)qid * p.cand_stride + first.out_rel);
            const float q_sq = p.q_sq[qid];
            const float thr = live ? p.thr[qid] : -INFINITY;
            const int n_tiles = (rows + 31) >> 5;
            const uint16_t* arow = p.rows16 + tiled_at(row0 + n, h, pieces);          // piece h of this lane's first row
            const uint16_t* brow = p.q16 + (int64_t)qid * p.ld16 + h * 8;
            const int64_t tile_step = (int64_t)32 * p.ld16;                            // elements from a tile to the next
            i4v b[KS];
            // |c|^2 of 256 rows at a time: lane l holds rows 4 l .. 4 l + 3 of the current 256, the next 256 are on their way
            f4u cs4 = *reinterpret_cast<const f4u*>(p.row_sq + row0 + 4 * lane);
            f4u cs4_next = cs4;
            // keys of one tile's rows from its accumulator; pairs within the threshold go to this wave's segment
            auto finish_tile = [&](int t, const f32x16& acc) {
                uint32_t hits = 0;
                float keys[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = t * 32 + 8 * g + 4 * h + i;
                        // |c|^2 of row r sits in lane (r mod 256) / 4, component i
                        const float csq = __shfl(cs4[i], (t & 7) * 8 + 2 * g + h);
                        const float key = __fmaf_rn(-2.0f, acc[g * 4 + i], q_sq + csq);
                        keys[g * 4 + i] = key;
                        if (key <= thr && r < rows) hits |= 1u << (g * 4 + i);
                    }
                }
                // kept pairs go to this wave's own segment: places are handed out by ballot, no atomic and nothing to wait for
                unsigned long long pending = __ballot(hits != 0);
                while (pending != 0ull) {
                    const bool has = hits != 0;
                    const int e = has ? __builtin_ctz(hits) : 0;
                    hits &= hits - 1;
                    const int r = t * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
                    const int64_t pos = row0 + r;
                    bool keep = has;
                    if (keep && p.mask) {
                        const int id = p.row_ids[pos];
                        keep = (p.mask[id >> 5] >> (id & 31)) & 1u;
                    }
                    const unsigned long long km = __ballot(keep);
                    const int at = seg_n + __popcll(km & ((1ull << lane) - 1ull));
                    if (keep) {
                        float key = keys[0];
#pragma unroll
                        for (int x = 1; x < 16; ++x) key = e == x ? keys[x] : key;
                        if (at < p.seg_cap) my_seg[at] = make_int4(qid, slot0 + r, (int)pos, __float_as_int(key));
                        else *p.overflow = 1;
                    }
                    seg_n += __popcll(km);
                    pending = __ballot(hits != 0);
                }
            };
            auto next_row_terms = [&](int t) {           // at the first tile of every 256 rows
                cs4 = cs4_next;
                if ((t + 8) * 32 < rows) cs4_next = *reinterpret_cast<const f4u*>(p.row_sq + row0 + (int64_t)(t + 8) * 32 + 4 * lane);
            };
            if (MULTI) {
                for (int t = 0; t < n_tiles; ++t) {
                    if ((t & 7) == 0) next_row_terms(t);
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
                    for (int c = 0; c < p.n_chunks; ++c) {
                        i4v a[KS];
#pragma unroll
                        for (int j = 0; j < KS; ++j)
                            a[j] = *reinterpret_cast<const i4v*>(arow + t * tile_step + (int64_t)(c * (CE / 8) + 2 * j) * 256);
#pragma unroll
                        for (int j = 0; j < KS; ++j) b[j] = *reinterpret_cast<const i4v*>(brow + c * CE + j * 16);
#pragma unroll
                        for (int j = 0; j < KS; ++j)
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[j]),
                                                                          acc, 0, 0, 0);
                    }
                    finish_tile(t, acc);
                }
            } else {
                // IA_SETS register sets of one tile each: while a tile is multiplied the next IA_SETS - 1 are on their way.
                // A tile past the end refetches the last one: no branch around the loads.
                i4v a[IA_SETS][KS];
                auto load_rows = [&](int t, i4v* a) {
                    const uint16_t* src = arow + min(t, n_tiles - 1) * tile_step;
#pragma unroll
                    for (int j = 0; j < KS; ++j) a[j] = *reinterpret_cast<const i4v*>(src + j * 512);
                };
                auto multiply = [&](int t, const i4v* a) {
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
                    for (int j = 0; j < KS; ++j)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[j]),
                                                                      acc, 0, 0, 0);
                    finish_tile(t, acc);
                };
#pragma unroll
                for (int j = 0; j < KS; ++j) b[j] = *reinterpret_cast<const i4v*>(brow + j * 16);
#pragma unroll
                for (int u = 0; u < IA_SETS - 1; ++u) load_rows(u, a[u]);
                for (int t = 0; t < n_tiles; t += IA_SETS) {
                    if ((t & 7) == 0) next_row_terms(t);
#pragma unroll
                    for (int u = 0; u < IA_SETS; ++u) {
                        if (t + u >= n_tiles) break;
                        load_rows(t + u + IA_SETS - 1, a[(u + IA_SETS - 1) % IA_SETS]);
                        multiply(t + u, a[u]);
                    }
                }
            }
        }
    }
    if (lane == 0) p.seg_len[wave0] = min(seg_n, p.seg_cap);
}

// --------------------------------------------------------------------------
// Bound pass.  One wave per (query, unit, tile): unit u covers candidate slots [256 u, 256 u + 256) of the query's
// visiting order, i.e. rows of its nearest probed lists, in eight tiles of 32; every lane finds the stored position of
// its own slot (slots of one tile may straddle two lists).  The query sits in every column of B, so lanes 0 and 32 hold
// the keys of the tile's rows 8 g + i and 8 g + 4 + i; each writes the smallest key of its sixteen rows that exist and
// pass the filter (a key below zero counts as zero: the k-th smallest of larger values is still a bound) to the subset
// (unit, tile, half).
//   A wave is a chain of dependent reads (list offsets -> probed list -> its first position -> rows), so the tiles of a
// unit are separate waves, and the offsets and numbers of the first eight probed lists (which hold the 2048 slots
// unless lists are shorter than 256 rows) come in one round of scalar loads.  What is left is the reading itself:
// every query reads its own 2048 rows (512 KiB in bf16 at 128 columns), half a gigabyte for 1024 queries.
struct IvfBoundParams {
    const uint16_t* rows16; int ld16; int n_chunks;
    const float* row_sq; const uint16_t* q16; const float* q_sq;
    const int* probes; int n_probe; int n_lists; const int64_t* cand_off; const int64_t* list_beg;
    const int* row_ids; const uint32_t* mask;
    float* mins;                           // [nq][IA_SUBSETS]
    int nq;
    int* queue_len; int* overflow;         // cleared here for the passes that follow: the queues' lengths, the flag
};

static constexpr int IB_AHEAD = 8;         // probed lists whose offsets are read ahead

template <int KS, bool MULTI>
__global__ __launch_bounds__(IA_NT) void ivf_bound_kernel(IvfBoundParams p) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, h = lane >> 5;
    const int w = blockIdx.x * (IA_NT / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = w / (IB_UNITS * 8), ut = w - q * (IB_UNITS * 8);
    if (q >= p.nq) return;
    if (ut == 0 && lane == 0) {
        p.queue_len[q * IA_LEN_STRIDE] = 0;
        if (q == 0) *p.overflow = 0;
    }
    const int u = ut >> 3, t = ut & 7;
    constexpr int CE = KS * 16;
    const int64_t* co = p.cand_off + (int64_t)q * (p.n_probe + 1);
    const int* probes = p.probes + (int64_t)q * p.n_probe;
    const int64_t total = uniform_load(co + p.n_probe);
    float* out = p.mins + (int64_t)q * IA_SUBSETS + (u * 8 + t) * 2 + h;
    if ((int64_t)u * 256 + t * 32 >= total) {                  // the same for the whole wave: nothing left to visit
        if (n == 0) *out = INFINITY;
        return;
    }
    const int ahead = min(p.n_probe, IB_AHEAD);
    int64_t c_at[IB_AHEAD + 1];
    int l_at[IB_AHEAD];
#pragma unroll
    for (int i = 0; i <= IB_AHEAD; ++i) c_at[i] = uniform_load(co + min(i, ahead));
#pragma unroll
    for (int i = 0; i < IB_AHEAD; ++i) l_at[i] = uniform_load(probes + min(i, ahead - 1));
    const float q_sq = p.q_sq[q];
    const uint16_t* brow = p.q16 + (int64_t)q * p.ld16 + h * 8;
    i4v b[KS];
    if (!MULTI) {
#pragma unroll
        for (int j = 0; j < KS; ++j) b[j] = *reinterpret_cast<const i4v*>(brow + j * 16);
    }
    const int64_t slot = (int64_t)u * 256 + t * 32 + n;
    bool ok = slot < total;
    int64_t pos = 0;
    if (ok) {
        // the probe this lane's slot lies in: co[pr] <= slot < co[pr + 1] (empty lists have equal offsets and are stepped over)
        int pr = 0;
        int64_t first = c_at[0];
        int l = l_at[0];
#pragma unroll
        for (int i = 1; i < IB_AHEAD; ++i)
            if (i < ahead && c_at[i] <= slot) { pr = i; first = c_at[i]; l = l_at[i]; }
        if (c_at[IB_AHEAD] <= slot && ahead == IB_AHEAD) {     // beyond the lists read ahead (rare)
            pr = IB_AHEAD;
            while (co[pr + 1] <= slot) ++pr;
            first = co[pr];
            l = probes[pr];
        }
        pos = p.list_beg[l] + (slot - first);
        if (p.mask) {
            const int id = p.row_ids[pos];
            ok = (p.mask[id >> 5] >> (id & 31)) & 1u;
        }
    }
    const unsigned long long okm = __ballot(ok);
    const float csq_own = p.row_sq[pos];
    const uint16_t* arow = p.rows16 + tiled_at(pos, h, p.ld16 >> 3);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int chunks = MULTI ? p.n_chunks : 1;
    for (int c = 0; c < chunks; ++c) {
        i4v a[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) a[j] = *reinterpret_cast<const i4v*>(arow + (int64_t)(c * (CE / 8) + 2 * j) * 256);
        if (MULTI) {
#pragma unroll
            for (int j = 0; j < KS; ++j) b[j] = *reinterpret_cast<const i4v*>(brow + c * CE + j * 16);
        }
#pragma unroll
        for (int j = 0; j < KS; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[j]),
                                                          acc, 0, 0, 0);
    }
    float best = INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * g + 4 * h + i;               // the tile's row: lane r (and r + 32) looked it up
            const float csq = __shfl(csq_own, r);
            const float key = __fmaf_rn(-2.0f, acc[g * 4 + i], q_sq + csq);
            if ((okm >> r) & 1ull) best = fminf(best, fmaxf(key, 0.0f));
        }
    }
    if (n == 0) *out = best;
}

// --------------------------------------------------------------------------
// Per query: B = the k-th smallest of its subset minima (an upper bound of the k-th smallest approximate key over the
// rows of its nearest lists), then the threshold of the scan pass.
//
// Let d be the squared distance in real arithmetic, R the reference's float32 sum of squared differences and A the
// approximate key, u = 2^-24.
//   |A - d| <= 2 E + (D + 4) u (|q|^2 + |c|^2): E bounds the error of the bf16 dot product as the matrix cores form it,
//              (2^-7 + 2^-16) |q||c| for rounding both operands to 8 significant bits plus D 2^-22 |q||c| for the
//              float32 accumulation; the rest are the roundings of the two norm sums, their sum and the final fma;
//   |R - d| <= (D + 2) u d (1 + ...) <= 2 (D + 2) u (|q|^2 + |c|^2): two roundings per term, D - 1 in the sum.
// With S = 2 E + (4 D + 16) u (|q|^2 + max|c|^2) every row has |A - R| <= S.  k rows have A <= B, hence R <= B + S, so
// the reference's k-th smallest sum is at most B + S; rows that tie it after the square root lie within one part in
// 2^22 of it; and any row with R <= (B + S)(1 + 2^-21) has A <= (B + S)(1 + 2^-21) + S, which is the threshold.
// Norms whose squares or products could leave float32 (or are not numbers) raise the overflow flag instead.
__global__ __launch_bounds__(256) void ivf_threshold_kernel(const float* __restrict__ mins, int nq, int k, int dim,
                                                            const float* __restrict__ q_sq, const float* __restrict__ q_nrm,
                                                            const float* __restrict__ max_row_nrm, float* __restrict__ thr,
                                                            float* __restrict__ slack, int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    WaveTopK top;
    top.init(k);
    for (int s0 = 0; s0 < IA_SUBSETS; s0 += 64) {
        const float v = mins[(int64_t)q * IA_SUBSETS + s0 + lane];
        top.offer(v < INFINITY, v, s0 + lane, lane);
    }
    if (lane != 0) return;
    const float qn = q_nrm[q], big = *max_row_nrm;
    if (!(qn < 1e18f) || !(big < 1e18f)) { *overflow = 1; thr[q] = INFINITY; slack[q] = INFINITY; return; }
    const float s = key_slack(dim, qn, big, q_sq[q]);
    const float b = top.thr;                     // +inf while fewer than k subsets saw a row
    thr[q] = b < INFINITY ? ((b + s) * 1.000001f + s) : INFINITY;
    slack[q] = s;                                // the second threshold (ivf_finish_kernel) is widened by the same amount
}

// --------------------------------------------------------------------------
// The reference's sum of squared differences of one (query, row) pair: float32 q - c, its square, np.sum's pairwise
// order (leaves of at most 128 terms with eight strided partial sums, folded pairwise; distance/metrics.py:43).
// Two neighbouring lanes (m = 0, 1) share one pair: lane m keeps partial sums 4m .. 4m + 3 of the eight and reads its
// rows sixteen bytes at a time; the halves ((r0+r1)+(r2+r3)) and ((r4+r5)+(r6+r7)) meet through a lane swap.  Both
// lanes return the sum.
__device__ __forceinline__ f4u sqdiff4(const float* __restrict__ x, const float* __restrict__ q) {
    const f4u xv = *reinterpret_cast<const f4u*>(x);
    const f4u qv = *reinterpret_cast<const f4u*>(q);
    const f4u d = qv - xv;
    return d * d;
}

__device__ __forceinline__ float leaf_sqdiff(const float* __restrict__ x, const float* __restrict__ q, int len, int m) {
    if (len < 8) {
        float res = 0.0f;
        for (int i = 0; i < len; ++i) { const float d = q[i] - x[i]; res = res + d * d; }
        return res;
    }
    f4u r = sqdiff4(x + 4 * m, q + 4 * m);
    int i = 8;
    // eight, then four groups of eight at a time, then the last few together: their loads are in flight together, the
    // sums stay in order
    for (; i + 64 <= len; i += 64) {
        f4u xv[8], qv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            xv[u] = *reinterpret_cast<const f4u*>(x + i + 8 * u + 4 * m);
            qv[u] = *reinterpret_cast<const f4u*>(q + i + 8 * u + 4 * m);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const f4u d = qv[u] - xv[u]; r = r + d * d; }
    }
    for (; i + 32 <= len; i += 32) {
        f4u xv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = *reinterpret_cast<const f4u*>(x + i + 8 * u + 4 * m);
            qv[u] = *reinterpret_cast<const f4u*>(q + i + 8 * u + 4 * m);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const f4u d = qv[u] - xv[u]; r = r + d * d; }
    }
    {
        const int left = (len - i) >> 3;                     // 0 .. 3 whole groups
        f4u xv[3], qv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int at = u < left ? i + 8 * u : 0;         // (a group that is not there reads the first one again: unused)
            xv[u] = *reinterpret_cast<const f4u*>(x + at + 4 * m);
            qv[u] = *reinterpret_cast<const f4u*>(q + at + 4 * m);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u < left) { const f4u d = qv[u] - xv[u]; r = r + d * d; }
        i += 8 * left;
    }
    const float half = (r[0] + r[1]) + (r[2] + r[3]);
    float res = half + __shfl_xor(half, 1);                // the same value in both lanes: addition commutes
    for (; i < len; ++i) { const float d = q[i] - x[i]; res = res + d * d; }
    return res;
}

__device__ static float exact_sqdiff(const float* __restrict__ x, const float* __restrict__ q, const SumPlan& plan, int m) {
    if (plan.n_leaves == 1) return leaf_sqdiff(x, q, plan.leaf_len[0], m);
    float stack[8];
    int sp = 0, off = 0;
    for (int l = 0; l < plan.n_leaves; ++l) {
        const int len = plan.leaf_len[l];
        stack[sp++] = leaf_sqdiff(x + off, q + off, len, m);
        off += len;
        for (int f = 0; f < plan.folds[l]; ++f) {
            const float b = stack[--sp];
            const float a = stack[--sp];
            stack[sp++] = a + b;
        }
    }
    return stack[0];
}

// Blocks (segment of the scan pass, slice): every kept pair takes a place in its query's queue.  Neighbouring pairs
// of a segment come from one item, i.e. from a handful of queries: every lane finds the lanes of its wave that hold
// its query (64 broadcasts), the first of them reserves places for all with one atomic, and the others take theirs
// behind it: an eighth of the atomics of one per pair, which are what this kernel waits for.
struct IvfScatterParams {
    const int4* seg; const int* seg_len; int seg_cap;
    float* queue_key; int* queue_slot; int* queue_pos; int* queue_len; int queue_cap; int* overflow;   // per query: [nq][queue_cap]
};

__global__ __launch_bounds__(256) void ivf_scatter_kernel(IvfScatterParams p) {
    const int len = min(p.seg_len[blockIdx.x], p.seg_cap);
    const int lane = threadIdx.x & 63;
    const int4* seg = p.seg + (int64_t)blockIdx.x * p.seg_cap;
    for (int e0 = blockIdx.y * 256 + (threadIdx.x & ~63); e0 < len; e0 += 256 * gridDim.y) {      // (the same for a wave)
        const int e = e0 + lane;
        const bool have = e < len;
        const int4 ent = have ? seg[e] : make_int4(-1, 0, 0, 0);
        unsigned long long same = 0ull;                      // the lanes that hold this lane's query
#pragma unroll 8
        for (int l = 0; l < 64; ++l) {
            const int other = __builtin_amdgcn_readlane(ent.x, l);
            const unsigned long long m = __ballot(ent.x == other);
            if (ent.x == other) same = m;
        }
        const int leader = __builtin_ctzll(same | (1ull << 63));
        int base = 0;
        if (have && lane == leader) base = atomicAdd(&p.queue_len[ent.x * IA_LEN_STRIDE], __popcll(same));
        base = __shfl(base, leader);
        const int at = base + __popcll(same & ((1ull << lane) - 1ull));
        if (have) {
            if (at < p.queue_cap) {
                const int64_t o = (int64_t)ent.x * p.queue_cap + at;
                p.queue_key[o] = __int_as_float(ent.w);
                p.queue_slot[o] = ent.y;
                p.queue_pos[o] = ent.z;
            } else {
                *p.overflow = 1;
            }
        }
    }
}

// One block of four waves per query, over its queue of (key, slot, position) in any order.
//   U: every thread's smallest key, then the k-th smallest of those (found by counting) over the block or, for the
//      short queues of the scan, per wave and the smallest over the waves: k different pairs have keys within U.
//   The pairs within U widened (as the threshold kernel widens its bound) are gathered in LDS; among them the k-th
//   smallest key A_k is found by counting, and the pairs within A_k widened are the finalists: with S the query's
//   slack, k rows have reference sums R <= A_k + S, so the reference's k-th smallest sum is at most that; rows tying
//   it after the square root lie within one part in 2^22; and a row with R <= (A_k + S)(1 + 2^-21) has a key of at
//   most (A_k + S)(1 + 2^-21) + S.
//   The finalists' distances in the reference's arithmetic (a pair of lanes per finalist), then each finds its place
//   under (distance, slot) by counting.
// More than IF_HELD pairs within the first bound (thousands of equal rows): the overflow flag, as for a full queue.
//
// PROBE: the same over the keys of a query against every centroid (ranking the lists to probe, index/ivf.py:611-616):
// the "queue" is the query's row of the key matrix, slot = position = the centroid's number, the slack comes from the
// norms here.  More than IF_HELD centroids within the first bound (norms beyond float32's squares, or over a thousand
// equal centroids): the block computes every centroid's distance and takes the k smallest one after the other.
struct IvfFinishParams {
    const float* rows; int ld; SumPlan plan;     // the float32 rows, by position
    const float* q; int ldq;                     // the float32 queries (padded rows)
    int take_sqrt;                               // VDB_EUCLIDEAN_DIRECT: the distance is the root of the sum
    float* queue_key; const int* queue_slot; const int* queue_pos; const int* queue_len; int queue_cap;
    const float* slack;
    int dim; const float* q_sq; const float* q_nrm; const float* max_row_nrm;      // PROBE: what the slack is made of
    int exact_keys;                              // PROBE: the keys are the distances themselves (no slack, nothing recomputed)
    int k, k_out;
    float* out_d; int* out_slots;                // [nq][k_out]
    int* overflow;
};

static constexpr int IF_HELD = 1024;

// (value, id) as one 64-bit number whose unsigned order is the order of the pairs (value ascending, then id; values
// that are not numbers behind +inf; -0 counts as +0); the counting loops compare these, two per LDS read
__device__ __forceinline__ unsigned long long packed(float v, int id) {
    const uint32_t b = (uint32_t)__float_as_int(v + 0.0f);
    const uint32_t ordered = v == v ? b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u) : 0xffffffffu;
    return ((unsigned long long)ordered << 32) | (uint32_t)id;
}
__device__ __forceinline__ int count_before(const unsigned long long* pairs, int n, unsigned long long mine) {
    int c = 0, e = 0;
    for (; e + 2 <= n; e += 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(pairs + e);
        c += (v.x < mine ? 1 : 0) + (v.y < mine ? 1 : 0);
    }
    if (e < n) c += pairs[e] < mine ? 1 : 0;
    return c;
}

template <bool PROBE>
__global__ __launch_bounds__(256) void ivf_finish_kernel(IvfFinishParams p) {
    __shared__ __align__(16) float sh_key[IF_HELD];
    __shared__ int sh_slot[IF_HELD];
    __shared__ int sh_pos[IF_HELD];
    __shared__ __align__(16) float sh_d[IF_HELD];            // the finalists' distances
    __shared__ __align__(16) int sh_fs[IF_HELD];             // and slots
    __shared__ __align__(16) unsigned long long sh_pair[IF_HELD];        // what is being ranked, as packed pairs
    __shared__ int sh_fin[IF_HELD];              // the finalists (places in sh_key, sh_slot, sh_pos)
    __shared__ float sh_u, sh_kth;
    __shared__ int sh_cnt, sh_nfin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int k = p.k;
    const int total = PROBE ? p.queue_cap : min(p.queue_len[q * IA_LEN_STRIDE], p.queue_cap);
    float* keys = p.queue_key + (int64_t)q * p.queue_cap;
    float s;
    if (PROBE && p.exact_keys) {
        s = 0.0f;
    } else if (PROBE) {
        const float qn = p.q_nrm[q], big = *p.max_row_nrm;
        s = (qn < 1e18f && big < 1e18f) ? key_slack(p.dim, qn, big, p.q_sq[q]) : INFINITY;     // +inf: no key can be trusted
    } else {
        s = p.slack[q];
    }
    if (tid == 0) { sh_cnt = 0; sh_nfin = 0; sh_kth = INFINITY; sh_u = INFINITY; }
    // every place starts empty (the slots are mapped to rows whatever happens below: none may stay unwritten); the
    // finalists write theirs behind the barriers that follow
    for (int f = tid; f < k; f += 256) {
        p.out_d[(int64_t)q * p.k_out + f] = INFINITY;
        p.out_slots[(int64_t)q * p.k_out + f] = -1;
    }
    // U: every thread's smallest key (eight loads in flight at a time), then the k-th smallest of the 256 by counting
    constexpr int AH = 8;
    float hv = INFINITY;                         // threads without a pair (or with NaN alone) stand at +inf
    for (int i0 = 0; i0 < total; i0 += 256 * AH) {
        float v[AH];
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            const int i = i0 + a * 256 + tid;
            v[a] = i < total ? keys[i] : INFINITY;
        }
#pragma unroll
        for (int a = 0; a < AH; ++a)
            if (v[a] < hv) hv = v[a];
    }
    if (PROBE) {
        // thousands of keys, k up to 64: the k-th smallest of all 256 threads' minima (little above the k-th smallest key)
        const unsigned long long mine = packed(hv, tid);
        sh_pair[tid] = mine;
        __syncthreads();
        if (count_before(sh_pair, 256, mine) == k - 1) sh_u = hv;       // exactly one thread: the counts are a permutation
    } else {
        // hundreds of keys: a wave's k-th smallest of its lanes' minima, the smallest over the waves (no pass over LDS)
        int before = 0;
#pragma unroll
        for (int l = 0; l < 64; ++l) {
            const float other = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), l));
            before += (other < hv || (other == hv && l < lane)) ? 1 : 0;
        }
        const unsigned long long kth_lane = __ballot(before == k - 1);    // exactly one lane
        const float wave_thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), __builtin_ctzll(kth_lane)));
        if (lane == 0) sh_d[wave] = wave_thr;
        __syncthreads();
        if (tid == 0) sh_u = fminf(fminf(sh_d[0], sh_d[1]), fminf(sh_d[2], sh_d[3]));
    }
    __syncthreads();
    const float u = sh_u;
    const float first = u < INFINITY ? ((u + s) * 1.000001f + s) : INFINITY;
    // the pairs within the first bound, in LDS
    for (int i0 = 0; i0 < total; i0 += 256 * AH) {
        float v[AH];
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            const int i = i0 + a * 256 + tid;
            v[a] = i < total ? keys[i] : INFINITY;
        }
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            const int i = i0 + a * 256 + tid;
            const bool in = i < total && v[a] <= first;
            const unsigned long long m = __ballot(in);
            if (m != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh_cnt, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int at = base + __popcll(m & ((1ull << lane) - 1ull));
                if (in && at < IF_HELD) {
                    sh_key[at] = v[a];
                    sh_pair[at] = packed(v[a], at);
                    sh_slot[at] = PROBE ? i : p.queue_slot[(int64_t)q * p.queue_cap + i];
                    sh_pos[at] = PROBE ? i : p.queue_pos[(int64_t)q * p.queue_cap + i];
                }
            }
        }
    }
    __syncthreads();
    const int held = sh_cnt;
    const int half = tid & 1;
    const float* qrow = p.q + (int64_t)q * p.ldq;
    if (held > IF_HELD || (PROBE && !(s < INFINITY))) {      // the same for every thread of the block
        if (!PROBE) {
            if (tid == 0) *p.overflow = 1;
            return;
        }
        // every centroid's distance over its key, then k times the smallest (distance, number) behind the last one taken
        for (int i0 = 0; i0 < total && !p.exact_keys; i0 += 128) {
            const int i = i0 + (tid >> 1);
            if (i < total) {
                const float sum = exact_sqdiff(p.rows + (int64_t)i * p.ld, qrow, p.plan, half);
                if (half == 0) keys[i] = p.take_sqrt ? sqrtf(sum) : sum;
            }
        }
        __syncthreads();
        float last_d = -INFINITY;
        int last_i = -1;
        for (int r = 0; r < k; ++r) {
            float best_d = INFINITY;
            int best_i = EMPTY_IDX;
            for (int i = tid; i < total; i += 256) {
                const float d = keys[i];
                if (d == d && pair_less(last_d, last_i, d, i) && pair_less(d, i, best_d, best_i)) { best_d = d; best_i = i; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(best_d, off);
                const int oi = __shfl_xor(best_i, off);
                if (pair_less(od, oi, best_d, best_i)) { best_d = od; best_i = oi; }
            }
            if (lane == 0) { sh_d[wave] = best_d; sh_fin[wave] = best_i; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (pair_less(sh_d[w], sh_fin[w], best_d, best_i)) { best_d = sh_d[w]; best_i = sh_fin[w]; }
            __syncthreads();
            if (best_i == EMPTY_IDX) break;                  // fewer distances that are numbers than k: the rest stays empty
            if (tid == 0) { p.out_d[(int64_t)q * p.k_out + r] = best_d; p.out_slots[(int64_t)q * p.k_out + r] = best_i; }
            last_d = best_d;
            last_i = best_i;
        }
        return;
    }
    // A_k: the held key with k - 1 held pairs before it under (key, place)
    for (int t = tid; t < held; t += 256) {
        if (count_before(sh_pair, held, sh_pair[t]) == k - 1) sh_kth = sh_key[t];
    }
    __syncthreads();
    const float a_k = sh_kth;                    // +inf with fewer than k pairs: every pair is a finalist
    const float second = a_k < INFINITY ? ((a_k + s) * 1.000001f + s) : INFINITY;
    for (int t0 = 0; t0 < held; t0 += 256) {
        const int t = t0 + tid;
        const bool in = t < held && sh_key[t] <= second;
        const unsigned long long m = __ballot(in);
        if (m != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_nfin, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (in) sh_fin[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
        }
    }
    __syncthreads();
    const int nfin = sh_nfin;
    // the reference's distances of the finalists
    for (int f0 = 0; f0 < nfin; f0 += 128) {                 // (the pair of lanes of a finalist stays together)
        const int f = f0 + (tid >> 1);
        if (f < nfin) {
            float sum;
            if (PROBE && p.exact_keys) sum = sh_key[sh_fin[f]];
            else sum = exact_sqdiff(p.rows + (int64_t)sh_pos[sh_fin[f]] * p.ld, qrow, p.plan, half);
            if (half == 0) {
                const float d = p.take_sqrt ? sqrtf(sum) : sum;
                const int slot = sh_slot[sh_fin[f]];
                sh_d[f] = d; sh_fs[f] = slot; sh_pair[f] = packed(d, slot);
            }
        }
    }
    __syncthreads();
    for (int f = tid; f < nfin; f += 256) {
        const float my = sh_d[f];
        const int my_slot = sh_fs[f];
        if (!(my == my)) continue;                           // not a number: in no result (it precedes nothing and nothing precedes it)
        const int place = count_before(sh_pair, nfin, sh_pair[f]);
        if (place < k) { p.out_d[(int64_t)q * p.k_out + place] = my; p.out_slots[(int64_t)q * p.k_out + place] = my_slot; }
    }
}

__global__ void ivf_round_rows_kernel(const float* __restrict__ src, int64_t src_ld, int dim, int64_t n,
                                      uint16_t* __restrict__ dst16, int ld16) {
    const int64_t total = n * ld16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld16;
        const int c = (int)(i - r * ld16);
        uint32_t b = 0;
        if (c < dim) {
            const uint32_t x = __float_as_uint(src[r * src_ld + c]);
            b = ((x & 0x7fffffffu) > 0x7f800000u) ? ((x >> 16) | 0x40u) : ((x + 0x7fffu + ((x >> 16) & 1u)) >> 16);
        }
        dst16[i] = (uint16_t)b;
    }
}

// fp32 rows [pos0, pos0 + n) of a store -> the tiled bf16 copy (tiled_at).  A thread converts one 16-byte piece; threads
// run over (tile, piece, row of the tile), so 32 neighbours write 512 contiguous bytes.
__global__ void ivf_round_rows_tiled_kernel(const float* __restrict__ src, int64_t src_ld, int dim, int64_t pos0, int64_t n,
                                            uint16_t* __restrict__ dst16, int ld16) {
    const int pieces = ld16 >> 3;
    const int64_t first_tile = pos0 >> 5;
    const int64_t total = (((pos0 + n + 31) >> 5) - first_tile) * pieces * 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i & 31);
        const int c = (int)((i >> 5) % pieces);
        const int64_t pos = (first_tile + (i >> 5) / pieces) * 32 + r;
        if (pos < pos0 || pos >= pos0 + n) continue;
        const float* x = src + (pos - pos0) * src_ld + c * 8;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint32_t b = 0;
            if (c * 8 + e < dim) {
                const uint32_t v = __float_as_uint(x[e]);
                b = ((v & 0x7fffffffu) > 0x7f800000u) ? ((v >> 16) | 0x40u) : ((v + 0x7fffu + ((v >> 16) & 1u)) >> 16);
            }
            if (e & 1) w[e >> 1] |= b << 16; else w[e >> 1] = b & 0xffffu;
        }
        *reinterpret_cast<uint4*>(dst16 + tiled_at(pos, c, pieces)) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int launch_ivf_round_rows_tiled(const float* src, int64_t src_ld, int dim, int64_t pos0, int64_t n, uint16_t* dst16, int ld16,
                                hipStream_t s) {
    if (n <= 0) return VDB_OK;
    const int64_t total = (n + 64) * (ld16 >> 3);
    hipLaunchKernelGGL(ivf_round_rows_tiled_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s,
                       src, src_ld, dim, pos0, n, dst16, ld16);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int ivf_mfma_pitch(int dim) { return dim <= 128 ? round_up(dim, 32) : round_up(dim, 128); }

int launch_ivf_round_rows(const float* src, int64_t src_ld, int dim, int64_t n, uint16_t* dst16, int ld16, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    const int64_t total = n * ld16;
    hipLaunchKernelGGL(ivf_round_rows_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s,
                       src, src_ld, dim, n, dst16, ld16);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

template <class P, class K2, class K4, class K6, class K8, class KM>
static int launch_by_pitch(const P& p, int ld16, int n_chunks, dim3 grid, hipStream_t s, K2 k2, K4 k4, K6 k6, K8 k8, KM km) {
    if (n_chunks > 1) {
        hipLaunchKernelGGL(km, grid, dim3(IA_NT), 0, s, p);
    } else {
        switch (ld16 / 16) {
            case 2: hipLaunchKernelGGL(k2, grid, dim3(IA_NT), 0, s, p); break;
            case 4: hipLaunchKernelGGL(k4, grid, dim3(IA_NT), 0, s, p); break;
            case 6: hipLaunchKernelGGL(k6, grid, dim3(IA_NT), 0, s, p); break;
            case 8: hipLaunchKernelGGL(k8, grid, dim3(IA_NT), 0, s, p); break;
            default: set_error("bf16 row pitch %d has no matrix-core list scan", ld16); return VDB_ERR_INVALID;
        }
    }
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// Keys of every (query, centroid) pair for the ranking of the lists: one wave per 32 queries x 32 centroids, both read
// row-major from their bf16 copies (the centroids are a megabyte or so and stay in the caches).  The queries are the
// A operand here, so a lane holds one centroid's keys for sixteen queries and a store writes 32 neighbouring floats
// of a query's row.
struct IvfCentKeysParams {
    const uint16_t* q16; const uint16_t* c16; int ld16; int n_chunks;
    const float* q_sq; const float* c_sq; int nq; int n_lists;
    float* keys;                           // [nq][n_lists]
};

template <int KS, bool MULTI>
__global__ __launch_bounds__(IA_NT) void ivf_cent_keys_kernel(IvfCentKeysParams p) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, h = lane >> 5;
    const int w = blockIdx.x * (IA_NT / 64) + (threadIdx.x >> 6);
    const int c_tiles = (p.n_lists + 31) >> 5;
    const int qt = w / c_tiles, ct = w - qt * c_tiles;
    if (qt * 32 >= p.nq) return;
    constexpr int CE = KS * 16;
    const int q_own = min(qt * 32 + n, p.nq - 1), c_own = min(ct * 32 + n, p.n_lists - 1);      // rows past the end repeat the last
    const uint16_t* arow = p.q16 + (int64_t)q_own * p.ld16 + h * 8;
    const uint16_t* brow = p.c16 + (int64_t)c_own * p.ld16 + h * 8;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int chunks = MULTI ? p.n_chunks : 1;
    for (int c = 0; c < chunks; ++c) {
        i4v a[KS], b[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            a[j] = *reinterpret_cast<const i4v*>(arow + c * CE + j * 16);
            b[j] = *reinterpret_cast<const i4v*>(brow + c * CE + j * 16);
        }
#pragma unroll
        for (int j = 0; j < KS; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[j]),
                                                          acc, 0, 0, 0);
    }
    const float c_sq = p.c_sq[c_own];
    const bool c_ok = ct * 32 + n < p.n_lists;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qrow = qt * 32 + 8 * g + 4 * h + i;
            if (qrow < p.nq && c_ok)
                p.keys[(int64_t)qrow * p.n_lists + ct * 32 + n] = __fmaf_rn(-2.0f, acc[g * 4 + i], p.q_sq[qrow] + c_sq);
        }
    }
}

int launch_ivf_mfma_probe(const IvfProbeJob& j, hipStream_t s) {
    VDB_REQUIRE(j.n_probe >= 1 && j.n_probe <= 64 && j.n_probe <= j.n_lists, "matrix-core list ranking: n_probe in 1..64");
    VDB_REQUIRE(j.ld16 % 32 == 0 && (j.ld16 <= 128 || j.ld16 % 128 == 0), "bad bf16 row pitch %d", j.ld16);
    const int n_chunks = j.ld16 <= 128 ? 1 : j.ld16 / 128;
    IvfCentKeysParams kp{j.q16, j.cent16, j.ld16, n_chunks, j.q_sq, j.cent_sq, j.nq, j.n_lists, j.keys};
    const int64_t waves = (int64_t)((j.nq + 31) / 32) * ((j.n_lists + 31) / 32);
    VDB_TRY(launch_by_pitch(kp, j.ld16, n_chunks, dim3((unsigned)((waves + 3) / 4)), s, ivf_cent_keys_kernel<2, false>,
                            ivf_cent_keys_kernel<4, false>, ivf_cent_keys_kernel<6, false>, ivf_cent_keys_kernel<8, false>,
                            ivf_cent_keys_kernel<8, true>));
    IvfFinishParams e{};
    e.rows = j.cent; e.ld = j.ld; e.plan = j.sum_plan; e.q = j.q; e.ldq = j.ldq; e.take_sqrt = j.take_sqrt ? 1 : 0;
    e.queue_key = j.keys; e.queue_cap = j.n_lists;
    e.dim = j.dim; e.q_sq = j.q_sq; e.q_nrm = j.q_nrm; e.max_row_nrm = j.max_cent_nrm;
    e.k = j.n_probe; e.k_out = j.n_probe; e.out_d = j.probe_d; e.out_slots = j.probes;
    hipLaunchKernelGGL(ivf_finish_kernel<true>, dim3(j.nq), dim3(256), 0, s, e);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// The k best of every query's run of candidate distances (one float per candidate slot, as the IVF-PQ scan leaves
// them), k <= 64, under (distance, slot).  One block of four waves per query streams the run once, four 16-byte
// loads in flight per lane; every wave keeps its k best in its lanes (a candidate is offered only if it beats the wave's k-th:
// after the first few hundred nearly none does), and the four lists are ranked together by counting.  A few queries
// cannot fill the device with a block each: then gridDim.y blocks share a query's run, every one leaves the k best of
// its part as a list of (distance, slot) for the list merge.
// Four waves per block: a wave's insertions are serial chains of dependent instructions, every list costs its own
// k log(n / k) of them, and a SIMD shared by four such waves runs each at a quarter of the speed (1024 x 78 000
// candidates, k = 10: a batch 0.04 ms shorter than with sixteen waves per query; one query in parts: 31 -> 15 us).
static constexpr int CR_NT = 256;
static_assert(CR_NT == 256, "cand_rank_kernel reads four waves' bounds");

__global__ __launch_bounds__(CR_NT) void cand_rank_kernel(const float* __restrict__ cand, int64_t cand_stride,
                                                          const int64_t* __restrict__ cand_off, int n_probe, int k, int k_out,
                                                          float* __restrict__ out_d, int* __restrict__ out_slots,
                                                          float* __restrict__ part_d, int* __restrict__ part_i, int squares) {
    __shared__ __align__(16) unsigned long long sh_pair[CR_NT];          // four lists of k <= 64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int whole = (int)min(cand_off[(int64_t)q * (n_probe + 1) + n_probe], cand_stride);
    // this block's part of the run: a multiple of the 1024 floats the block reads at a time
    const int parts = gridDim.y;
    const int chunk = parts > 1 ? ((whole + parts - 1) / parts + CR_NT * 4 - 1) / (CR_NT * 4) * (CR_NT * 4) : whole;
    const int first = min((int)blockIdx.y * chunk, whole);
    const int total = min(first + chunk, whole);
    const float* run = cand + (int64_t)q * cand_stride;
    __shared__ float sh_kth[CR_NT / 64];         // every wave's k-th so far: none of the block's k best lies above any of them
    WaveTopK top;
    top.init(k);
    bool seeded = false;
    if (lane == 0) sh_kth[wave] = INFINITY;
    __syncthreads();
    // sixteen bytes per lane and load (a query's run starts on a multiple of 64 floats), four loads in flight
    constexpr int AH = 4;
    for (int i0 = first; i0 < total; i0 += CR_NT * 4 * AH) {
        // (read without a barrier: a wave's k-th only falls, and any value it has had is a valid bound)
        const float gate = fminf(fminf(sh_kth[0], sh_kth[1]), fminf(sh_kth[2], sh_kth[3]));
        // squares: the run holds squares of the distances; whatever has a root <= gate is <= pass_to (the product and
        // the correctly rounded root are each within 2^-24), and only that gets its root taken
        const float pass_to = squares ? gate * gate * 1.0000005f + 1e-37f : gate;
        float4 v[AH];
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            const int i = i0 + (a * CR_NT + tid) * 4;
            if (i + 4 <= total) {
                v[a] = *reinterpret_cast<const float4*>(run + i);
            } else {
                v[a].x = i < total ? run[i] : INFINITY;
                v[a].y = i + 1 < total ? run[i + 1] : INFINITY;
                v[a].z = i + 2 < total ? run[i + 2] : INFINITY;
                v[a].w = INFINITY;                           // (i + 3 >= total here)
            }
        }
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            const int i = i0 + (a * CR_NT + tid) * 4;
            if (i0 + (a * CR_NT + wave * 64) * 4 < total) {      // (the same for the whole wave)
                auto offer_one = [&](float x, int at) {
                    const bool pass = at < total && x <= pass_to;
                    if (__ballot(pass) == 0ull) return;           // (nearly always, after the first few hundred)
                    const float d = squares ? sqrtf(x) : x;
                    top.template offer<false, true>(pass && d <= gate, d, at, lane);
                };
                if (a == 0 && !seeded) {                         // (the list is empty: sixty-four at once)
                    const float d = squares ? sqrtf(v[a].x) : v[a].x;
                    top.seed(i < total && d == d, d, i, lane);
                    seeded = true;
                } else {
                    offer_one(v[a].x, i);
                }
                offer_one(v[a].y, i + 1);
                offer_one(v[a].z, i + 2);
                offer_one(v[a].w, i + 3);
            }
        }
        if (lane == 0) sh_kth[wave] = top.thr;
    }
    // (empty places of the lists: behind every candidate, and all different, so that every place gets written)
    if (lane < k) sh_pair[wave * k + lane] = top.eid != EMPTY_IDX ? packed(top.e, top.eid) : (0xffffffff80000000ull + (unsigned)tid);
    __syncthreads();
    const int n = (CR_NT / 64) * k;
    for (int t = tid; t < n; t += CR_NT) {
        const unsigned long long mine = sh_pair[t];
        const int place = count_before(sh_pair, n, mine);
        if (place < k) {
            const int i = (int)(uint32_t)mine;
            const bool empty = i < 0;
            const float d = empty ? INFINITY : (squares ? sqrtf(run[i]) : run[i]);
            if (parts > 1) {
                const int64_t at = ((int64_t)q * parts + blockIdx.y) * k + place;
                part_d[at] = d;
                part_i[at] = empty ? EMPTY_IDX : i;
            } else {
                out_d[(int64_t)q * k_out + place] = d;
                out_slots[(int64_t)q * k_out + place] = empty ? -1 : i;
            }
        }
    }
}

int launch_cand_rank(const float* cand, int64_t cand_stride, const int64_t* cand_off, int n_probe, int nq, int k, int k_out,
                     float* out_d, int* out_slots, hipStream_t s, int parts, float* part_d, int* part_i, bool squares) {
    if (nq <= 0) return VDB_OK;
    VDB_REQUIRE(k >= 1 && k <= 64 && cand_stride < (int64_t)0x7fffffff, "candidate ranking: k in 1..64, fewer than 2^31 candidates");
    VDB_REQUIRE(parts >= 1 && (parts == 1 || (part_d && part_i)), "candidate ranking in parts needs room for the lists");
    hipLaunchKernelGGL(cand_rank_kernel, dim3(nq, parts), dim3(CR_NT), 0, s, cand, cand_stride, cand_off, n_probe, k, k_out, out_d, out_slots,
                       part_d, part_i, squares ? 1 : 0);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// the k smallest of every row of a [nq][n] matrix of distances under (distance, column), k <= 64: the same block per
// query over exact keys
int launch_rank_rows(float* dist, int nq, int n, int k, float* out_d, int* out_i, hipStream_t s) {
    if (nq <= 0) return VDB_OK;
    VDB_REQUIRE(k >= 1 && k <= 64 && k <= n, "row ranking: k in 1..64");
    IvfFinishParams e{};
    e.queue_key = dist; e.queue_cap = n; e.exact_keys = 1; e.k = k; e.k_out = k; e.out_d = out_d; e.out_slots = out_i;
    hipLaunchKernelGGL(ivf_finish_kernel<true>, dim3(nq), dim3(256), 0, s, e);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

int launch_ivf_mfma_search(const IvfMfmaJob& j, hipStream_t s) {
    VDB_REQUIRE(j.k >= 1 && j.k <= 64 && j.k_out >= j.k, "matrix-core list scan: k in 1..64");
    VDB_REQUIRE(j.ld16 % 32 == 0 && (j.ld16 <= 128 || j.ld16 % 128 == 0), "bad bf16 row pitch %d", j.ld16);
    const int n_chunks = j.ld16 <= 128 ? 1 : j.ld16 / 128;
    IvfBoundParams bp{};
    bp.rows16 = j.rows16; bp.ld16 = j.ld16; bp.n_chunks = n_chunks; bp.row_sq = j.row_sq; bp.q16 = j.q16; bp.q_sq = j.q_sq;
    bp.probes = j.probes; bp.n_probe = j.n_probe; bp.n_lists = j.n_lists; bp.cand_off = j.cand_off; bp.list_beg = j.list_beg;
    bp.row_ids = j.row_ids; bp.mask = j.mask; bp.mins = j.mins; bp.nq = j.nq; bp.queue_len = j.queue_len; bp.overflow = j.overflow;
    VDB_TRY(launch_by_pitch(bp, j.ld16, n_chunks, dim3((unsigned)(((int64_t)j.nq * IB_UNITS * 8 + 3) / 4)), s,
                            ivf_bound_kernel<2, false>, ivf_bound_kernel<4, false>, ivf_bound_kernel<6, false>,
                            ivf_bound_kernel<8, false>, ivf_bound_kernel<8, true>));
    hipLaunchKernelGGL(ivf_threshold_kernel, dim3((j.nq + 3) / 4), dim3(256), 0, s, j.mins, j.nq, j.k, j.dim, j.q_sq, j.q_nrm,
                       j.max_row_nrm, j.thr, j.slack, j.overflow);
    VDB_HIP(hipGetLastError());
    IvfApproxParams p{};
    p.rows16 = j.rows16; p.ld16 = j.ld16; p.n_chunks = n_chunks;
    p.row_sq = j.row_sq; p.q16 = j.q16; p.q_sq = j.q_sq;
    p.items = j.plan.items; p.seg_item_off = j.plan.seg_item_off; p.n_seg = j.n_lists; p.items_cap = j.plan.items_cap;
    p.sq_qid = j.plan.sq_qid; p.sq_out = j.plan.sq_out; p.cand_stride = j.cand_stride;
    p.thr = j.thr; p.seg = j.seg; p.seg_len = j.seg_len; p.seg_cap = j.seg_cap;
    p.overflow = j.overflow; p.row_ids = j.row_ids; p.mask = j.mask;
    const int scan_blocks = ivf_mfma_scan_waves() / (IA_NT / 64);
    {
        KernelTimer timer(s, j.timed);
        VDB_TRY(launch_by_pitch(p, j.ld16, n_chunks, dim3(scan_blocks), s, ivf_approx_kernel<2, false>,
                                ivf_approx_kernel<4, false>, ivf_approx_kernel<6, false>, ivf_approx_kernel<8, false>,
                                ivf_approx_kernel<8, true>));
    }
    static const bool debug = getenv("VDB_IVF_DEBUG") != nullptr;
    if (debug) {
        std::vector<float> thr((size_t)j.nq);
        std::vector<int> len((size_t)ivf_mfma_scan_waves());
        int flag = 0;
        float big = 0;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(thr.data(), j.thr, thr.size() * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipMemcpy(len.data(), j.seg_len, len.size() * sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&flag, j.overflow, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&big, j.max_row_nrm, sizeof(float), hipMemcpyDeviceToHost);
        int inf = 0, mx = 0; long long tot = 0; float tmin = INFINITY, tmax = 0;
        for (int q = 0; q < j.nq; ++q) {
            if (!(thr[q] < INFINITY)) ++inf; else { tmin = std::min(tmin, thr[q]); tmax = std::max(tmax, thr[q]); }
        }
        for (int v : len) { mx = std::max(mx, v); tot += v; }
        fprintf(stderr, "[ivf mfma] nq=%d k=%d ld16=%d max|c|=%g thr: %d inf, finite %g..%g; kept pairs: %lld (%.1f per query), "
                "largest segment %d of %d; flag=%d\n", j.nq, j.k, j.ld16, big, inf, tmin, tmax, tot, (double)tot / j.nq, mx, j.seg_cap, flag);
    }
    IvfScatterParams sc{j.seg, j.seg_len, j.seg_cap, j.queue_d, j.queue_slot, j.queue_pos, j.queue_len, j.queue_cap, j.overflow};
    // (a segment's pairs over two blocks: segments differ in length by an order of magnitude)
    hipLaunchKernelGGL(ivf_scatter_kernel, dim3(ivf_mfma_scan_waves(), 2), dim3(256), 0, s, sc);
    VDB_HIP(hipGetLastError());
    IvfFinishParams e{};
    e.rows = j.rows; e.ld = j.ld; e.plan = j.sum_plan; e.q = j.q; e.ldq = j.ldq; e.take_sqrt = j.take_sqrt ? 1 : 0;
    e.queue_key = j.queue_d; e.queue_slot = j.queue_slot; e.queue_pos = j.queue_pos; e.queue_len = j.queue_len;
    e.queue_cap = j.queue_cap; e.slack = j.slack; e.k = j.k; e.k_out = j.k_out; e.out_d = j.out_d; e.out_slots = j.slots;
    e.overflow = j.overflow;
    hipLaunchKernelGGL(ivf_finish_kernel<false>, dim3(j.nq), dim3(256), 0, s, e);
    VDB_HIP(hipGetLastError());
    // slots -> rows
    return launch_ivf_rows(j.slots, j.nq, j.k, j.k_out, 0, j.probes, j.n_probe, j.n_lists, j.cand_off, j.list_beg, j.row_ids,
                           j.out_rows, j.out_order, s);
}

int ivf_mfma_subsets() { return IA_SUBSETS; }
int ivf_mfma_len_stride() { return IA_LEN_STRIDE; }
// waves of the scan pass (eight per CU, three tiles in flight each) = segments of kept pairs
int ivf_mfma_scan_waves() { return ctx().num_cus * 2 * (IA_NT / 64); }

}  // namespace vdb
