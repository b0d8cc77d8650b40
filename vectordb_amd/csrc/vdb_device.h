// Device-side helpers shared by the scan kernels: per-thread sorted candidate
// lists in LDS and their block-level reduction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdb {

static constexpr int LIST_NT = 256;         // threads per block of every kernel that keeps lists
static constexpr int EMPTY_IDX = 0x7fffffff;

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

// Reduce the 256 per-thread lists of a block to its k best under the
// (distance, id) order: k rounds of "smallest list head".  red must hold 12
// ints of LDS.  Thread 0 writes od[0..k), oi[0..k).  All threads must call.
__device__ static void block_reduce_lists(const float* Ld, const int* Li, int k, int tid, int* red,
                                          float* od, int* oi) {
    float* red_d = reinterpret_cast<float*>(red);
    int* red_i = red + 4;
    int* red_t = red + 8;
    const int lane = tid & 63, wave = tid >> 6;
    int head = 0;
    for (int round = 0; round < k; ++round) {
        float d = head < k ? Ld[head * LIST_NT + tid] : INFINITY;
        int id = head < k ? Li[head * LIST_NT + tid] : EMPTY_IDX;
        int who = tid;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float d2 = __shfl_xor(d, off);
            const int i2 = __shfl_xor(id, off);
            const int w2 = __shfl_xor(who, off);
            if (d2 < d || (d2 == d && (i2 < id || (i2 == id && w2 < who)))) { d = d2; id = i2; who = w2; }
        }
        if (lane == 0) { red_d[wave] = d; red_i[wave] = id; red_t[wave] = who; }
        __syncthreads();
        float bd = red_d[0]; int bi = red_i[0]; int bt = red_t[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red_d[w] < bd || (red_d[w] == bd && red_i[w] < bi)) { bd = red_d[w]; bi = red_i[w]; bt = red_t[w]; }
        if (tid == bt) ++head;
        if (tid == 0) { od[round] = bd; oi[round] = bi; }
        __syncthreads();
    }
}

}  // namespace vdb
