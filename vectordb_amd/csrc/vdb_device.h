// Device-side helpers shared by the scan kernels: per-thread sorted candidate
// lists in LDS and their block-level reduction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdb {

static constexpr int LIST_NT = 256;         // threads per block of every kernel that keeps lists
static constexpr int EMPTY_IDX = 0x7fffffff;
#ifndef VDB_WAVE_BULK_MIN
#define VDB_WAVE_BULK_MIN 12
#endif
static constexpr int WAVE_BULK_MIN = VDB_WAVE_BULK_MIN;   // candidates in one offer from which the sort-and-merge path is taken

// One LDS-DMA instruction: every lane fetches 16 bytes from its own global address; the
// wave's 1 KiB lands at lds_dst (a wave-uniform LDS byte address) in lane order.  Written
// as asm so that the compiler neither counts it nor drains it before later LDS reads;
// completion is awaited with wait_loads() below.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// The same with the address split into a wave-uniform base (an SGPR pair) and a 32-bit byte offset per lane:
// nothing of the address has to be computed on the vector ALU.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// Insert (v, id) into the calling thread's ascending list (consecutive
// positions are LIST_NT floats apart).  A thread sees ids in ascending order,
// so an equal distance is placed behind the entries already present.
// Returns the new k-th (largest kept) distance.
__device__ __noinline__ static float list_insert(float* ld, int* li, int k, float v, int id) {
    int p = k - 1;
    while (p > 0) {
        const float prev = ld[(p - 1) * LIST_NT];
        if (!(prev > v)) break;
        ld[p * LIST_NT] = prev;
        li[p * LIST_NT] = li[(p - 1) * LIST_NT];
        --p;
    }
    ld[p * LIST_NT] = v;
    li[p * LIST_NT] = id;
    return ld[(k - 1) * LIST_NT];
}

__device__ __forceinline__ static void list_init(float* Ld, int* Li, int k, int tid) {
    for (int i = 0; i < k; ++i) { Ld[i * LIST_NT + tid] = INFINITY; Li[i * LIST_NT + tid] = EMPTY_IDX; }
}

// --------------------------------------------------------------------------
// Wave-wide selection: the wave's k best (distance, id) pairs live in registers,
// lane j holding the j-th smallest (k <= 64; lanes >= k carry spill-over).  The
// k-th pair is the acceptance bound, so after a warm-up almost every tile of
// candidates is rejected with one compare and one ballot per wave.
__device__ __forceinline__ static bool pair_less(float a, int ai, float b, int bi) { return (a < b) || (a == b && ai < bi); }

// The 64 smallest of (the wave's sorted pairs (e, eid), one candidate pair per lane), sorted again: the candidates
// are sorted across the lanes by a bitonic network, reversed, merged lane by lane (the smaller pair stays) and the
// resulting bitonic sequence is sorted with six more exchanges.  Pairs are ordered by (distance, id); lanes without
// a candidate pass (INFINITY, EMPTY_IDX).  Returns this lane's new pair, id in the high word.  Not inlined: it
// is the rare path of WaveTopK::offer, taken when many lanes bring a candidate at once.
// The value of lane (l ^ J), without a trip through LDS (a bpermute costs an LDS round trip of a hundred cycles or
// more; a selection network is a chain of them): quad permutes for 1 and 2, two masked row shifts for 4, a row
// rotation for 8, the gfx950 row and half-wave swaps for 16 and 32.
template <int J> __device__ __forceinline__ static int lane_xor(int x) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "one bit");
    if (J == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);           // quad_perm:[1,0,3,2]
    if (J == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);           // quad_perm:[2,3,0,1]
    if (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);           // row_shl:4 into banks 0 and 2
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                  // row_shr:4 into banks 1 and 3
    }
    if (J == 8) return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);           // row_ror:8
    typedef unsigned lane_u2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    if (J == 16) {
        const lane_u2 r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((lane & 16) ? r.x : r.y);
    }
    const lane_u2 r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (int)((lane & 32) ? r.x : r.y);
}
template <int J> __device__ __forceinline__ static float lane_xor(float x) { return __int_as_float(lane_xor<J>(__float_as_int(x))); }
// lane l - 1's value (lane 0 keeps its own)
__device__ __forceinline__ static int lane_up1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xF, 0xF, false); }   // wave_shr:1
__device__ __forceinline__ static float lane_up1(float x) { return __int_as_float(lane_up1(__float_as_int(x))); }
// lane 63 - l's value
__device__ __forceinline__ static int lane_reverse(int x) {
    return lane_xor<32>(lane_xor<16>(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false)));   // row_mirror, then the rows
}
__device__ __forceinline__ static float lane_reverse(float x) { return __int_as_float(lane_reverse(__float_as_int(x))); }

template <int K2, int J> __device__ __forceinline__ static void bitonic_step(float& cd, int& cid, int lane) {
    const float pd = lane_xor<J>(cd);
    const int pid = lane_xor<J>(cid);
    const bool keep_min = ((lane & K2) == 0) == ((lane & J) == 0);
    const bool take = keep_min ? ((pd < cd) || (pd == cd && pid < cid)) : ((cd < pd) || (cd == pd && cid < pid));
    if (take) { cd = pd; cid = pid; }
    if constexpr (J > 1) bitonic_step<K2, J / 2>(cd, cid, lane);
}
template <int K2> __device__ __forceinline__ static void bitonic_sort_upto(float& cd, int& cid, int lane) {
    if constexpr (K2 > 2) bitonic_sort_upto<K2 / 2>(cd, cid, lane);
    bitonic_step<K2, K2 / 2>(cd, cid, lane);
}

__device__ __forceinline__ static unsigned long long wave_merge_body(float e, int eid, float cd, int cid) {
    const int lane = threadIdx.x & 63;
    bitonic_sort_upto<64>(cd, cid, lane);      // K2 = 64: (lane & 64) == 0 everywhere, ascending over the wave
    {
        const float rd = lane_reverse(cd);
        const int rid = lane_reverse(cid);
        if (pair_less(rd, rid, e, eid)) { e = rd; eid = rid; }
    }
    bitonic_step<64, 32>(e, eid, lane);
    return ((unsigned long long)(unsigned)eid << 32) | (unsigned)__float_as_int(e);
}
// The same network with LDS permutes for the exchanges: fewer vector instructions per stage, each with the latency
// of an LDS round trip.  The form for kernels that keep many waves per SIMD busy (their permutes overlap); the
// register-only form above is for a block that has a CU, or the chip, to itself.
__device__ __noinline__ static unsigned long long wave_merge_candidates(float e, int eid, float cd, int cid) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const float pd = __shfl_xor(cd, j);
            const int pid = __shfl_xor(cid, j);
            const bool keep_min = ((lane & k2) == 0) == ((lane & j) == 0);
            const bool take = keep_min ? pair_less(pd, pid, cd, cid) : pair_less(cd, cid, pd, pid);
            if (take) { cd = pd; cid = pid; }
        }
    }
    const float rd = __shfl(cd, 63 - lane);
    const int rid = __shfl(cid, 63 - lane);
    if (pair_less(rd, rid, e, eid)) { e = rd; eid = rid; }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const float pd = __shfl_xor(e, j);
        const int pid = __shfl_xor(eid, j);
        const bool take = ((lane & j) == 0) ? pair_less(pd, pid, e, eid) : pair_less(e, eid, pd, pid);
        if (take) { e = pd; eid = pid; }
    }
    return ((unsigned long long)(unsigned)eid << 32) | (unsigned)__float_as_int(e);
}

// How far the key |q|^2 + |c|^2 - 2 bf16(q).bf16(c) (products and sums in float32, any order) can lie from the float32
// forms of the squared distance, direct (sum of squared differences) or identity (|q|^2 + |c|^2 - 2 q.c with float32
// sums): S = 2 E + (4 D + 16) u (|q|^2 + |c|^2), u = 2^-24, E = (2^-7 + 2^-16 + D 2^-22) |q||c| for rounding both
// operands to 8 significant bits and accumulating D products.  qn = |q| (|q|^2 = q_sq), big >= |c|, both below 1e18.
__device__ __forceinline__ float key_slack(int dim, float qn, float big, float q_sq) {
    const float e_dot = (0.0078278f + (float)dim * 2.4e-7f) * 1.0001f * qn * big + sqrtf((float)dim) * 1.2e-38f * (qn + big) + 1e-30f;
    return 2.0f * e_dot + (4.0f * (float)dim + 16.0f) * 6.0e-8f * 1.01f * (q_sq + big * big);
}

struct WaveTopK {
    float e; int eid;            // this lane's element
    float thr; int thr_id;       // the k-th element (wave-uniform)
    int k;

    __device__ __forceinline__ void init(int k_) { e = INFINITY; eid = EMPTY_IDX; thr = INFINITY; thr_id = EMPTY_IDX; k = k_; }

    // the first candidates of an empty list, one per lane (want = this lane has one): sorted across the lanes in
    // registers instead of inserted one by one
    __device__ __forceinline__ void seed(bool want, float d, int id, int lane) {
        e = want ? d : INFINITY;
        eid = want ? id : EMPTY_IDX;
        bitonic_sort_upto<64>(e, eid, lane);
        thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), k - 1));
        thr_id = __builtin_amdgcn_readlane(eid, k - 1);
    }

    // insert one pair; v / vid must be wave-uniform.  Pairs equal to a stored one go behind it.
    template <bool INL = false>
    __device__ __forceinline__ void insert(float v, int vid, int lane) {
        const bool gt = (e > v) || (e == v && eid > vid);
        const float pe = INL ? lane_up1(e) : __shfl_up(e, 1);
        const int pid = INL ? lane_up1(eid) : __shfl_up(eid, 1);
        const bool pgt = lane > 0 && ((pe > v) || (pe == v && pid > vid));
        if (gt) { e = pgt ? pe : v; eid = pgt ? pid : vid; }
        thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), k - 1));
        thr_id = __builtin_amdgcn_readlane(eid, k - 1);
    }

    // offer every lane's candidate (want = this lane has one); lanes are taken in lane order
    // BULK = false leaves the sort-and-merge path (a function call) out of kernels whose inner loop cannot
    // afford the registers saved around it
    // INL = true: the register-only merge network, expanded in place (for blocks that run alone: see wave_merge_body)
    template <bool BULK = true, bool INL = false>
    __device__ __forceinline__ void offer(bool want, float d, int id, int lane) {
        unsigned long long m = __ballot(want && ((d < thr) || (d == thr && id < thr_id)));
        if (BULK && __popcll(m) >= WAVE_BULK_MIN) {
            // many at once (an empty or a loosely bounded list): one sort-and-merge instead of that many insertions
            const bool mine = (m >> lane) & 1ull;
            const unsigned long long r = INL ? wave_merge_body(e, eid, mine ? d : INFINITY, mine ? id : EMPTY_IDX)
                                             : wave_merge_candidates(e, eid, mine ? d : INFINITY, mine ? id : EMPTY_IDX);
            e = __int_as_float((int)(unsigned)r);
            eid = (int)(r >> 32);
            thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), k - 1));
            thr_id = __builtin_amdgcn_readlane(eid, k - 1);
            return;
        }
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
            const int vid = __builtin_amdgcn_readlane(id, l);
            if ((v < thr) || (v == thr && vid < thr_id)) insert<INL>(v, vid, lane);
        }
    }

    // lanes 0..k-1 write the sorted list
    __device__ __forceinline__ void store(float* od, int* oi, int lane) const {
        if (lane < k) { od[lane] = e; oi[lane] = eid; }
    }
};

// Reduce the 256 per-thread lists of a block to its k best under the
// (distance, id) order.  Every wave first merges its own 64 lists (k rounds of
// "smallest list head", shuffles only), parks the result in its first column
// of the list area, and after one barrier wave 0 merges the four wave lists.
// k <= 64.  The lists are destroyed.  `red` is unused scratch kept for the
// callers' LDS layout.  Thread 0..k-1 write od[0..k), oi[0..k).  All threads
// must call.
__device__ static void block_reduce_lists(float* Ld, int* Li, int k, int tid, int* red, float* od, int* oi) {
    (void)red;
    const int lane = tid & 63, wave = tid >> 6;
    int head = 0;
    float my_d = INFINITY;
    int my_i = EMPTY_IDX;
    for (int round = 0; round < k; ++round) {
        float d = head < k ? Ld[head * LIST_NT + tid] : INFINITY;
        int id = head < k ? Li[head * LIST_NT + tid] : EMPTY_IDX;
        int who = lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float d2 = __shfl_xor(d, off);
            const int i2 = __shfl_xor(id, off);
            const int w2 = __shfl_xor(who, off);
            if (d2 < d || (d2 == d && (i2 < id || (i2 == id && w2 < who)))) { d = d2; id = i2; who = w2; }
        }
        if (lane == who) ++head;
        if (lane == round) { my_d = d; my_i = id; }
    }
    __syncthreads();                    // every wave is done reading its lists
    if (lane < k) { Ld[lane * LIST_NT + wave * 64] = my_d; Li[lane * LIST_NT + wave * 64] = my_i; }
    __syncthreads();
    if (wave != 0) return;
    // lanes 0..3 walk the four wave lists
    const int col = (lane & 3) * 64;
    head = 0;
    my_d = INFINITY; my_i = EMPTY_IDX;
    for (int round = 0; round < k; ++round) {
        float d = (lane < 4 && head < k) ? Ld[head * LIST_NT + col] : INFINITY;
        int id = (lane < 4 && head < k) ? Li[head * LIST_NT + col] : EMPTY_IDX;
        int who = lane;
#pragma unroll
        for (int off = 2; off >= 1; off >>= 1) {
            const float d2 = __shfl_xor(d, off);
            const int i2 = __shfl_xor(id, off);
            const int w2 = __shfl_xor(who, off);
            if (d2 < d || (d2 == d && (i2 < id || (i2 == id && w2 < who)))) { d = d2; id = i2; who = w2; }
        }
        d = __shfl(d, 0); id = __shfl(id, 0); who = __shfl(who, 0);
        if (lane == who) ++head;
        if (lane == round) { my_d = d; my_i = id; }
    }
    if (lane < k) { od[lane] = my_d; oi[lane] = my_i; }
}

}  // namespace vdb
