// Final selection stages.
//
//  select_kernel   : per query, merge many ascending candidate lists of k into
//                    the k best under the (distance, tie key) order.  Stands for
//                    the argpartition+argsort of index/flat.py:354-362 and the
//                    stable list.sort of core/collection.py:757,
//                    index/ivf.py:631 and index/pq.py:300, applied to the
//                    per-block survivors instead of all n rows.
//  row_sort_kernel : full ascending sort of one row of distances (n <= 16384)
//                    in LDS; np.argsort(centroid_distances)[:n_probe]
//                    (index/ivf.py:614) when n_probe is large.
#include "vdb_internal.h"
#include "vdb_device.h"

namespace vdb {

template <class IdxT> struct IdxTraits;
template <> struct IdxTraits<int> {
    __device__ static bool empty(int v) { return v == EMPTY_IDX; }
};
template <> struct IdxTraits<int64_t> {
    __device__ static bool empty(int64_t v) { return v < 0; }
};

// one wave per query
template <class IdxT>
__global__ __launch_bounds__(64) void select_kernel(const float* __restrict__ d, const IdxT* __restrict__ idx,
                                                    int64_t q_stride, int64_t l_stride, int64_t l_stride_idx,
                                                    const int64_t* __restrict__ list_base, int n_lists, int k,
                                                    int tie_group, const int* __restrict__ idx_map, int64_t idx_offset,
                                                    const int64_t* __restrict__ payload, SelectOut o) {
    extern __shared__ int heads[];
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    for (int l = lane; l < n_lists; l += 64) heads[l] = 0;
    __syncthreads();
    const float* dq = d + (int64_t)q * q_stride;
    const IdxT* iq = idx + (int64_t)q * q_stride;
    const int64_t* pq = payload ? payload + (int64_t)q * q_stride : nullptr;

    float last_d = INFINITY;
    int64_t last_key = INT64_MAX;
    for (int round = 0; round < k; ++round) {
        float bd = INFINITY;
        int64_t bkey = INT64_MAX;
        int64_t bidx = -1;
        int bl = 0x7fffffff;
        for (int l = lane; l < n_lists; l += 64) {
            const int h = heads[l];
            if (h >= k) continue;
            const float v = dq[(int64_t)l * l_stride + h];
            const IdxT id = iq[(int64_t)l * l_stride_idx + h];
            if (IdxTraits<IdxT>::empty(id)) continue;
            // ids of list l count from list_base[l] (shards of consecutive rows)
            const int64_t gid = (int64_t)id + (list_base ? list_base[l] : 0);
            const int64_t key = (sizeof(IdxT) == 4) ? (((int64_t)(l / tie_group) << 32) | (uint32_t)id) : gid;
            if (v < bd || (v == bd && key < bkey)) { bd = v; bkey = key; bidx = gid; bl = l; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float d2 = __shfl_xor(bd, off);
            const int64_t k2 = __shfl_xor(bkey, off);
            const int64_t i2 = __shfl_xor(bidx, off);
            const int l2 = __shfl_xor(bl, off);
            if (d2 < bd || (d2 == bd && (k2 < bkey || (k2 == bkey && l2 < bl)))) { bd = d2; bkey = k2; bidx = i2; bl = l2; }
        }
        if (lane == 0) {
            int64_t row = -1;
            float dist = INFINITY;
            if (bl != 0x7fffffff) {
                heads[bl] += 1;
                row = pq ? pq[(int64_t)bl * l_stride_idx + heads[bl] - 1] : (idx_map ? (int64_t)idx_map[bidx] : bidx) + idx_offset;
                dist = bd;
                last_d = bd;
                last_key = bkey;
            } else {
                last_d = INFINITY;
                last_key = INT64_MAX;
            }
            const int64_t at = (int64_t)q * o.out_stride + o.out_off + round;
            if (o.out_d) o.out_d[at] = dist;
            if (o.out_i) o.out_i[at] = row;
            if (o.out_i32) o.out_i32[at] = (int)row;
        }
        __syncthreads();
    }
    if (lane == 0 && o.last_d) { o.last_d[q] = last_d; o.last_key[q] = last_key; }
}

// --------------------------------------------------------------------------
// Merge of up to 256 ascending lists of k (int ids, order (distance, id)) by one
// block: the lists are staged in LDS, every wave merges 64 of them (one list
// per lane, k rounds of "smallest head"), then wave 0 merges the wave results.
// FINAL = false: writes one list of k to out_d/out_i[(q * n_groups + group) * k].
// FINAL = true : writes the results in SelectOut form.
static constexpr int MERGE_LISTS = 256;

template <bool FINAL>
__global__ __launch_bounds__(256) void merge_lists_kernel(const float* __restrict__ d, const int* __restrict__ idx,
                                                          int n_lists, int k, float* __restrict__ mid_d, int* __restrict__ mid_i,
                                                          const int* __restrict__ idx_map, int64_t idx_offset, SelectOut o) {
    if (o.run_if && *o.run_if == 0) return;
    extern __shared__ __align__(16) float sm[];
    float* Sd = sm;                                            // [MERGE_LISTS + 4][k]
    int* Si = reinterpret_cast<int*>(Sd + (MERGE_LISTS + 4) * k);
    const int q = blockIdx.x, group = blockIdx.y, n_groups = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l0 = group * MERGE_LISTS;
    const int nl = min(MERGE_LISTS, n_lists - l0);
    const int64_t base = ((int64_t)q * n_lists + l0) * k;
    for (int i = tid; i < MERGE_LISTS * k; i += 256) {
        const bool in = i < nl * k;
        Sd[i] = in ? d[base + i] : INFINITY;
        Si[i] = in ? idx[base + i] : EMPTY_IDX;
    }
    __syncthreads();

    auto merge64 = [&](int first_list, int count, int out_list) {
        // lane l walks list first_list + l; the wave's k best go to list slot out_list
        const int mine = first_list + lane;
        int head = 0;
        float cd = lane < count ? Sd[mine * k] : INFINITY;
        int ci = lane < count ? Si[mine * k] : EMPTY_IDX;
        float keep_d = INFINITY;
        int keep_i = EMPTY_IDX;
        for (int round = 0; round < k; ++round) {
            float bd = cd; int bi = ci; int who = lane;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float d2 = __shfl_xor(bd, off);
                const int i2 = __shfl_xor(bi, off);
                const int w2 = __shfl_xor(who, off);
                if (d2 < bd || (d2 == bd && (i2 < bi || (i2 == bi && w2 < who)))) { bd = d2; bi = i2; who = w2; }
            }
            if (lane == who && lane < count) {
                ++head;
                cd = head < k ? Sd[mine * k + head] : INFINITY;
                ci = head < k ? Si[mine * k + head] : EMPTY_IDX;
            }
            if (lane == round) { keep_d = bd; keep_i = bi; }
        }
        if (lane < k) { Sd[out_list * k + lane] = keep_d; Si[out_list * k + lane] = keep_i; }
    };

    merge64(wave * 64, max(0, min(64, nl - wave * 64)), MERGE_LISTS + wave);
    __syncthreads();
    if (wave != 0) return;
    merge64(MERGE_LISTS, 4, MERGE_LISTS);          // result over slot MERGE_LISTS (read before it is written: per-round reads
                                                   // touch heads only, and the store happens after the last round)
    __builtin_amdgcn_wave_barrier();
    if (lane < k) {
        const float bd = Sd[MERGE_LISTS * k + lane];
        const int bi = Si[MERGE_LISTS * k + lane];
        if (!FINAL) {
            mid_d[((int64_t)q * n_groups + group) * k + lane] = bd;
            mid_i[((int64_t)q * n_groups + group) * k + lane] = bi;
        } else {
            const bool empty = bi == EMPTY_IDX;
            const int64_t row = empty ? -1 : (idx_map ? (int64_t)idx_map[bi] : (int64_t)bi) + idx_offset;
            const int64_t at = (int64_t)q * o.out_stride + o.out_off + lane;
            if (o.out_d) o.out_d[at] = empty ? INFINITY : bd;
            if (o.out_i) o.out_i[at] = row;
            if (o.out_i32) o.out_i32[at] = (int)row;
            if (lane == k - 1 && o.last_d) {
                o.last_d[q] = empty ? INFINITY : bd;
                o.last_key[q] = empty ? INT64_MAX : (int64_t)bi;
            }
        }
    }
}

// A handful of lists per query (e.g. the two half-wave lists of a nearest-centroid
// assignment over a million rows): one thread merges them.
static constexpr int FEW_LISTS = 8;

__global__ __launch_bounds__(256) void merge_few_kernel(const float* __restrict__ d, const int* __restrict__ idx, int nq,
                                                        int n_lists, int k, const int* __restrict__ idx_map,
                                                        int64_t idx_offset, SelectOut o) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq || (o.run_if && *o.run_if == 0)) return;
    const float* dq = d + (int64_t)q * n_lists * k;
    const int* iq = idx + (int64_t)q * n_lists * k;
    int head[FEW_LISTS];
    float cd[FEW_LISTS];
    int ci[FEW_LISTS];
#pragma unroll
    for (int l = 0; l < FEW_LISTS; ++l) {
        head[l] = 0;
        cd[l] = l < n_lists ? dq[l * k] : INFINITY;
        ci[l] = l < n_lists ? iq[l * k] : EMPTY_IDX;
    }
    float bd = INFINITY;
    int bi = EMPTY_IDX;
    for (int round = 0; round < k; ++round) {
        int bl = 0;
        bd = cd[0]; bi = ci[0];
#pragma unroll
        for (int l = 1; l < FEW_LISTS; ++l)
            if (cd[l] < bd || (cd[l] == bd && ci[l] < bi)) { bd = cd[l]; bi = ci[l]; bl = l; }
#pragma unroll
        for (int l = 0; l < FEW_LISTS; ++l) {
            if (l == bl) {
                ++head[l];
                const bool more = l < n_lists && head[l] < k;
                cd[l] = more ? dq[l * k + head[l]] : INFINITY;
                ci[l] = more ? iq[l * k + head[l]] : EMPTY_IDX;
            }
        }
        const bool empty = bi == EMPTY_IDX;
        const int64_t row = empty ? -1 : (idx_map ? (int64_t)idx_map[bi] : (int64_t)bi) + idx_offset;
        const int64_t at = (int64_t)q * o.out_stride + o.out_off + round;
        if (o.out_d) o.out_d[at] = empty ? INFINITY : bd;
        if (o.out_i) o.out_i[at] = row;
        if (o.out_i32) o.out_i32[at] = (int)row;
    }
    if (o.last_d) {
        const bool empty = bi == EMPTY_IDX;
        o.last_d[q] = empty ? INFINITY : bd;
        o.last_key[q] = empty ? INT64_MAX : (int64_t)bi;
    }
}

// A few queries with a modest number of candidates each: one wave per query offers the lists' entries, 64 at a
// time, to a register-resident selection (loads coalesced, no dependent load per output as in merge_few_kernel).
__global__ __launch_bounds__(256) void merge_wave_kernel(const float* __restrict__ d, const int* __restrict__ idx, int nq,
                                                         int n_lists, int k, const int* __restrict__ idx_map,
                                                         int64_t idx_offset, SelectOut o) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq || (o.run_if && *o.run_if == 0)) return;
    const int total = n_lists * k;
    const float* dq = d + (int64_t)q * total;
    const int* iq = idx + (int64_t)q * total;
    WaveTopK top;
    top.init(k);
    // eight loads per lane in flight: the walk is a chain of global-memory round trips otherwise
    constexpr int MW_AHEAD = 8;
    for (int i0 = 0; i0 < total; i0 += 64 * MW_AHEAD) {
        float v[MW_AHEAD];
        int id[MW_AHEAD];
#pragma unroll
        for (int a = 0; a < MW_AHEAD; ++a) {
            const int i = i0 + a * 64 + lane;
            v[a] = i < total ? dq[i] : INFINITY;
            id[a] = i < total ? iq[i] : EMPTY_IDX;
        }
#pragma unroll
        for (int a = 0; a < MW_AHEAD; ++a) {
            if (i0 + a * 64 < total) top.offer(id[a] != EMPTY_IDX, v[a], id[a], lane);
        }
    }
    if (lane < k) {
        const bool empty = top.eid == EMPTY_IDX;
        const int64_t row = empty ? -1 : (idx_map ? (int64_t)idx_map[top.eid] : (int64_t)top.eid) + idx_offset;
        const int64_t at = (int64_t)q * o.out_stride + o.out_off + lane;
        if (o.out_d) o.out_d[at] = empty ? INFINITY : top.e;
        if (o.out_i) o.out_i[at] = row;
        if (o.out_i32) o.out_i32[at] = (int)row;
        if (lane == k - 1 && o.last_d) {
            o.last_d[q] = empty ? INFINITY : top.e;
            o.last_key[q] = empty ? INT64_MAX : (int64_t)top.eid;
        }
    }
}

__global__ void merge_block_kernel(const float* __restrict__ d, const int* __restrict__ idx, int area, int k, SelectOut o,
                                   int* __restrict__ queue_len, int lists_of_k, int len_stride, const int* __restrict__ redo);
static constexpr int MB_WAVES = 16;

int launch_select(const float* part_d, const int* part_i, int nq, int n_cand, int k, int tie_group,
                  const int* idx_map, int64_t idx_offset, const SelectOut& o, hipStream_t s, DevBuf* mid) {
    if (nq <= 0) return VDB_OK;
    (void)tie_group;        // every caller orders equal distances by id alone
    VDB_REQUIRE(k >= 1 && k <= 64, "selection merges lists of at most 64, got %d", k);
    if (nq <= 1024 && (int64_t)n_cand * k <= 2048) {
        hipLaunchKernelGGL(merge_wave_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, part_d, part_i, nq, n_cand, k,
                           idx_map, idx_offset, o);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    if (n_cand <= FEW_LISTS) {
        hipLaunchKernelGGL(merge_few_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, part_d, part_i, nq, n_cand, k,
                           idx_map, idx_offset, o);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    if (nq <= 8 && idx_map == nullptr && idx_offset == 0 && (int64_t)n_cand * k <= (1 << 17)) {
        // a handful of queries: one block of sixteen waves per query walks the lists as one run of entries, instead of
        // two or three dependent launches of list merges
        hipLaunchKernelGGL(merge_block_kernel, dim3(nq), dim3(64 * MB_WAVES), 0, s, part_d, part_i, n_cand * k, k, o, (int*)nullptr, 1, 1, (const int*)nullptr);
        VDB_HIP(hipGetLastError());
        return VDB_OK;
    }
    const size_t lds = (size_t)(MERGE_LISTS + 4) * k * 8;
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(merge_lists_kernel<false>),
                                lds));
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(merge_lists_kernel<true>),
                                lds));
    const float* cur_d = part_d;
    const int* cur_i = part_i;
    int n = n_cand, level = 0;
    while (n > MERGE_LISTS) {
        const int groups = (n + MERGE_LISTS - 1) / MERGE_LISTS;
        DevBuf& md = mid[level & 1];
        DevBuf& mi = mid[2 + (level & 1)];
        VDB_TRY(md.reserve((size_t)nq * groups * k * sizeof(float)));
        VDB_TRY(mi.reserve((size_t)nq * groups * k * sizeof(int)));
        hipLaunchKernelGGL(merge_lists_kernel<false>, dim3(nq, groups), dim3(256), lds, s, cur_d, cur_i, n, k,
                           md.as<float>(), mi.as<int>(), (const int*)nullptr, (int64_t)0, o);
        cur_d = md.as<float>(); cur_i = mi.as<int>();
        n = groups;
        ++level;
    }
    hipLaunchKernelGGL(merge_lists_kernel<true>, dim3(nq, 1), dim3(256), lds, s, cur_d, cur_i, n, k,
                       (float*)nullptr, (int*)nullptr, idx_map, idx_offset, o);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// One block of up to sixteen waves per query picks the k best of the query's packed candidate area (queue_len[q]
// entries of `area`, in any order).
//   Up to 512 entries per wave, k <= 16: every thread takes its entries into registers; a wave's k-th smallest among
// its list heads (every k-th entry of an area made of sorted lists of k, lists_of_k; else every lane's smallest
// entry) is a distance that k entries do not exceed, and so is the smallest of the waves' values, B.  The entries
// within B (little more than k per wave) are gathered in LDS and each finds its place by counting the ones that
// precede it: no chain of merges.  More than 1024 within B (a bound that does not bind): the general path below.
//   Otherwise the waves share out the entries 64 at a time (256 entries or more per wave, so a short queue keeps few
// waves busy), each keeps its k best in registers, and wave 0 merges the waves' lists through LDS.
__global__ __launch_bounds__(64 * MB_WAVES) void merge_block_kernel(const float* __restrict__ d, const int* __restrict__ idx,
                                                                    int area, int k, SelectOut o, int* __restrict__ queue_len,
                                                                    int lists_of_k, int len_stride, const int* __restrict__ redo) {
    __shared__ float sh_d[MB_WAVES * 64];
    __shared__ int sh_i[MB_WAVES * 64];
    __shared__ float sh_thr[MB_WAVES];
    __shared__ int sh_cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x;
    if (o.run_if && *o.run_if == 0) return;
    if (redo && *redo == 0) return;                              // select_small_kernel answered every query
    if (queue_len) queue_len += (int64_t)q * len_stride - q;     // a query's length lies len_stride ints from its neighbour's
    const int total = queue_len ? min(__builtin_amdgcn_readfirstlane(queue_len[q]), area) : area;
    if (redo && total == 0) return;                              // ... or this one (and cleared its length)
    const int waves = max(1, min((int)(blockDim.x >> 6), (total + 255) / 256));
    const float* dq = d + (int64_t)q * area;
    const int* iq = idx + (int64_t)q * area;
    constexpr int AHEAD = 8;
    auto write_out = [&](int place, bool empty, float v, int id) {
        const int64_t row = empty ? -1 : (int64_t)id;
        const int64_t at = (int64_t)q * o.out_stride + o.out_off + place;
        if (o.out_d) o.out_d[at] = empty ? INFINITY : v;
        if (o.out_i) o.out_i[at] = row;
        if (o.out_i32) o.out_i32[at] = (int)row;
        if (place == k - 1 && o.last_d) {
            o.last_d[q] = empty ? INFINITY : v;
            o.last_key[q] = empty ? INT64_MAX : (int64_t)id;
        }
    };
    if (k <= 16 && total <= (int)blockDim.x * AHEAD) {                          // the same for every thread of the block
        if (threadIdx.x == 0) sh_cnt = 0;
        float v[AHEAD];
        int id[AHEAD];
        WaveTopK heads;
        heads.init(k);
        if (wave < waves) {
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) {
                const int i = (a * waves + wave) * 64 + lane;
                v[a] = i < total ? dq[i] : INFINITY;
                id[a] = i < total ? iq[i] : EMPTY_IDX;
            }
            if (lists_of_k) {
#pragma unroll
                for (int a = 0; a < AHEAD; ++a) {
                    const int i = (a * waves + wave) * 64 + lane;
                    if ((a * waves + wave) * 64 < total) heads.template offer<false, false>(id[a] != EMPTY_IDX && i % k == 0, v[a], id[a], lane);
                }
            } else {
                // entries in any order: every lane's smallest entry stands in for a list head (64 different entries)
                float hv = INFINITY;
                int hid = EMPTY_IDX;
#pragma unroll
                for (int a = 0; a < AHEAD; ++a)
                    if (id[a] != EMPTY_IDX && (hid == EMPTY_IDX || v[a] < hv)) { hv = v[a]; hid = id[a]; }
                heads.template offer<false, false>(hid != EMPTY_IDX, hv, hid, lane);
            }
        }
        if (lane == 0) sh_thr[wave] = wave < waves ? heads.thr : INFINITY;
        __syncthreads();
        float bound = INFINITY;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) bound = fminf(bound, sh_thr[w]);
        if (wave < waves) {
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) {
                const bool in = id[a] != EMPTY_IDX && v[a] <= bound;
                const unsigned long long m = __ballot(in);
                if (m != 0ull) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&sh_cnt, __popcll(m));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const int at = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (in && at < MB_WAVES * 64) { sh_d[at] = v[a]; sh_i[at] = id[a]; }
                }
            }
        }
        __syncthreads();
        const int within = sh_cnt;
        if (within <= MB_WAVES * 64) {
            if (queue_len && threadIdx.x == 0) queue_len[q] = 0;      // every wave has read it (the barriers above)
            const int t = threadIdx.x;
            if (t < within) {
                const float my = sh_d[t];
                const int my_id = sh_i[t];
                int place = 0;
                for (int e = 0; e < within; ++e) place += pair_less(sh_d[e], sh_i[e], my, my_id) ? 1 : 0;
                if (place < k) write_out(place, false, my, my_id);
            } else if (t < k) {
                write_out(t, true, 0.0f, 0);                           // fewer entries than k
            }
            return;
        }
        __syncthreads();                                               // sh_* are used again below
    }
    WaveTopK top;
    top.init(k);
    if (wave < waves) {
        for (int i0 = wave * 64; i0 < total; i0 += 64 * waves * AHEAD) {
            float v[AHEAD];
            int id[AHEAD];
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) {
                const int i = i0 + a * 64 * waves + lane;
                v[a] = i < total ? dq[i] : INFINITY;
                id[a] = i < total ? iq[i] : EMPTY_IDX;
            }
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) {
                if (i0 + a * 64 * waves < total) top.template offer<true, true>(id[a] != EMPTY_IDX, v[a], id[a], lane);
            }
        }
        // the waves' lists packed end to end (k entries each)
        if (lane < k) { sh_d[wave * k + lane] = top.e; sh_i[wave * k + lane] = top.eid; }
    }
    __syncthreads();
    if (wave != 0) return;
    if (queue_len && lane == 0) queue_len[q] = 0;          // every wave has read it (the barrier above)
    for (int i0 = k; i0 < waves * k; i0 += 64) {           // from wave 1's list on: wave 0's own is in `top` already
        const int i = i0 + lane;
        const float v = i < waves * k ? sh_d[i] : INFINITY;
        const int id = i < waves * k ? sh_i[i] : EMPTY_IDX;
        top.template offer<true, true>(id != EMPTY_IDX, v, id, lane);
    }
    if (lane < k) write_out(lane, top.eid == EMPTY_IDX, top.e, top.eid);
}

// The k best (k <= 16) of a query's candidate area of up to 512 entries per wave, in any order, empty entries marked by
// their id: merge_block_kernel's first way alone, in a kernel of a few kilobytes (a block per query starts on a
// cold instruction cache, and merge_block_kernel is 90 KB of merge networks).  A wave's bound is the k-th smallest of
// its lanes' smallest entries, found by counting (every lane counts the lanes whose value precedes its own).
// A longer area, or more than 1024 entries within the bound: `redo` is set and the query's length stays, for
// merge_block_kernel behind that flag.
static constexpr int SS_AHEAD = 8;
static constexpr int SS_HELD = 1024;
__global__ __launch_bounds__(1024) void select_small_kernel(const float* __restrict__ d, const int* __restrict__ idx, int area,
                                                            int k, SelectOut o, int* __restrict__ queue_len, int len_stride,
                                                            int* __restrict__ redo) {
    __shared__ float sh_d[SS_HELD];
    __shared__ int sh_i[SS_HELD];
    __shared__ float sh_thr[16];
    __shared__ int sh_cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int q = blockIdx.x;
    if (o.run_if && *o.run_if == 0) return;
    if (queue_len) queue_len += (int64_t)q * len_stride - q;
    const int total = queue_len ? min(__builtin_amdgcn_readfirstlane(queue_len[q]), area) : area;
    if (total > nw * 64 * SS_AHEAD) {                            // the same for every thread of the block
        if (threadIdx.x == 0) *redo = 1;
        return;
    }
    const int waves = max(1, min(nw, (total + 255) / 256));
    const float* dq = d + (int64_t)q * area;
    const int* iq = idx + (int64_t)q * area;
    if (threadIdx.x == 0) sh_cnt = 0;
    float v[SS_AHEAD];
    int id[SS_AHEAD];
    float thr = INFINITY;
    if (wave < waves) {
#pragma unroll
        for (int a = 0; a < SS_AHEAD; ++a) {
            const int i = (a * waves + wave) * 64 + lane;
            v[a] = i < total ? dq[i] : INFINITY;
            id[a] = i < total ? iq[i] : EMPTY_IDX;
        }
        float hv = INFINITY;                                     // lanes without an entry (or with NaN alone) stand at +inf
#pragma unroll
        for (int a = 0; a < SS_AHEAD; ++a)
            if (id[a] != EMPTY_IDX && v[a] < hv) hv = v[a];
        int before = 0;
#pragma unroll
        for (int l = 0; l < 64; ++l) {
            const float other = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), l));
            before += (other < hv || (other == hv && l < lane)) ? 1 : 0;
        }
        const unsigned long long kth = __ballot(before == k - 1);          // exactly one lane: the counts are a permutation
        thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), __builtin_ctzll(kth)));
    } else {
#pragma unroll
        for (int a = 0; a < SS_AHEAD; ++a) { v[a] = INFINITY; id[a] = EMPTY_IDX; }
    }
    if (lane == 0) sh_thr[wave] = thr;
    __syncthreads();
    float bound = INFINITY;
    for (int w = 0; w < nw; ++w) bound = fminf(bound, sh_thr[w]);
    if (wave < waves) {
#pragma unroll
        for (int a = 0; a < SS_AHEAD; ++a) {
            const bool in = id[a] != EMPTY_IDX && v[a] <= bound;
            const unsigned long long m = __ballot(in);
            if (m != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh_cnt, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int at = base + __popcll(m & ((1ull << lane) - 1ull));
                if (in && at < SS_HELD) { sh_d[at] = v[a]; sh_i[at] = id[a]; }
            }
        }
    }
    __syncthreads();
    const int within = sh_cnt;
    if (within > SS_HELD) {                                      // a bound that does not bind
        if (threadIdx.x == 0) *redo = 1;
        return;
    }
    if (queue_len && threadIdx.x == 0) queue_len[q] = 0;         // every wave has read it (the barriers above)
    for (int t = threadIdx.x; t < max(within, k); t += blockDim.x) {
        bool empty = t >= within;
        float my = 0.0f;
        int my_id = 0, place = t;
        if (!empty) {
            my = sh_d[t];
            my_id = sh_i[t];
            place = 0;
            for (int e = 0; e < within; ++e) place += pair_less(sh_d[e], sh_i[e], my, my_id) ? 1 : 0;
        } else if (t >= k) {
            continue;
        }
        if (place < k) {
            const int64_t row = empty ? -1 : (int64_t)my_id;
            const int64_t at = (int64_t)q * o.out_stride + o.out_off + place;
            if (o.out_d) o.out_d[at] = empty ? INFINITY : my;
            if (o.out_i) o.out_i[at] = row;
            if (o.out_i32) o.out_i32[at] = (int)row;
            if (place == k - 1 && o.last_d) {
                o.last_d[q] = empty ? INFINITY : my;
                o.last_key[q] = empty ? INT64_MAX : (int64_t)my_id;
            }
        }
    }
}

// `redo` (one int, zero before the call): the short kernel first, merge_block_kernel behind the flag for what it left
int launch_select_packed(const float* part_d, const int* part_i, int nq, int area, int k, const SelectOut& o, hipStream_t s,
                         int* queue_len, int len_stride, int waves, int* redo) {
    if (nq <= 0) return VDB_OK;
    VDB_REQUIRE(k >= 1 && k <= 64 && queue_len, "packed selection: k in 1..64 and a queue length per query");
    VDB_REQUIRE(waves >= 1 && waves <= MB_WAVES && waves * k <= 64 * MB_WAVES, "packed selection: 1..%d waves per query", MB_WAVES);
    if (k > 16) redo = nullptr;
    if (redo) hipLaunchKernelGGL(select_small_kernel, dim3(nq), dim3(64 * waves), 0, s, part_d, part_i, area, k, o, queue_len, len_stride, redo);
    hipLaunchKernelGGL(merge_block_kernel, dim3(nq), dim3(64 * waves), 0, s, part_d, part_i, area, k, o, queue_len, 0, len_stride,
                       (const int*)redo);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// the k best of every query's area of `area` entries in any order, empty ones marked by their id
int launch_select_area(const float* part_d, const int* part_i, int nq, int area, int k, const SelectOut& o, hipStream_t s) {
    if (nq <= 0) return VDB_OK;
    VDB_REQUIRE(k >= 1 && k <= 64, "area selection: k in 1..64");
    if (k <= 16 && area <= SS_HELD) {        // everything fits the short kernel's LDS: it cannot leave a query behind
        const int waves = std::max(1, std::min(MB_WAVES, (area + 255) / 256));
        hipLaunchKernelGGL(select_small_kernel, dim3(nq), dim3(64 * waves), 0, s, part_d, part_i, area, k, o, (int*)nullptr, 1,
                           (int*)nullptr);
    } else {
        hipLaunchKernelGGL(merge_block_kernel, dim3(nq), dim3(64 * MB_WAVES), 0, s, part_d, part_i, area, k, o, (int*)nullptr, 0, 1,
                           (const int*)nullptr);
    }
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// lists laid out [n_lists][nq][k] (the shape an all-gather of per-shard results produces); list l of
// the distances starts at d + l * stride_d, of the ids (and payload) at i + l * stride_i
int launch_select_i64(const float* d, const int64_t* i, const int64_t* payload, int64_t stride_d, int64_t stride_i,
                      const int64_t* list_base, int n_lists, int nq, int k, float* out_d, int64_t* out_i, hipStream_t s) {
    if (nq <= 0) return VDB_OK;
    SelectOut o{out_d, out_i, nullptr, k, 0, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(select_kernel<int64_t>, dim3(nq), dim3(64), (size_t)n_lists * sizeof(int), s,
                       d, i, (int64_t)k, stride_d, stride_i, list_base, n_lists, k, 1, (const int*)nullptr, (int64_t)0, payload, o);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

// --------------------------------------------------------------------------
// bitonic sort of one row in LDS, ascending (value, index)
__global__ __launch_bounds__(256) void row_sort_kernel(const float* __restrict__ dist, int n, int n_pad, int k,
                                                       float* __restrict__ out_d, int* __restrict__ out_i) {
    extern __shared__ __align__(16) float sm[];
    float* key = sm;
    int* val = reinterpret_cast<int*>(sm + n_pad);
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_pad; i += 256) {
        key[i] = i < n ? dist[(int64_t)q * n + i] : INFINITY;
        val[i] = i < n ? i : EMPTY_IDX;
    }
    __syncthreads();
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < n_pad / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const float a = key[lo], b = key[hi];
                const int ia = val[lo], ib = val[hi];
                const bool a_gt_b = (a > b) || (a == b && ia > ib) || (a != a && b == b);   // NaN sorts last
                if (a_gt_b == up) { key[lo] = b; key[hi] = a; val[lo] = ib; val[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        out_d[(int64_t)q * k + i] = i < n ? key[i] : INFINITY;
        out_i[(int64_t)q * k + i] = i < n ? val[i] : EMPTY_IDX;
    }
}

int launch_row_sort_topk(const float* dist, int nq, int n, int k, float* out_d, int* out_i, hipStream_t s) {
    VDB_REQUIRE(n >= 1 && n <= 16384, "row sort handles 1..16384 values per row, got %d", n);
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    const size_t lds = (size_t)n_pad * 8;
    VDB_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(row_sort_kernel),
                                lds));
    hipLaunchKernelGGL(row_sort_kernel, dim3(nq), dim3(256), lds, s, dist, n, n_pad, k, out_d, out_i);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
