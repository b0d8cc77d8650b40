// One Lloyd accumulation step: per-cluster sums and member counts.
// index/ivf.py:177-182 and index/pq.py:160-165 add the rows one at a time in
// Python; here every row adds into fp64 accumulators with device atomics, so
// the order of additions is free but the fp64 sums agree with a sequential
// fp32/fp64 accumulation far below fp32 resolution.
#include "vdb_internal.h"

namespace vdb {

__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float* __restrict__ x, int64_t n, int dim, int ldx,
                                                                const int* __restrict__ assign, int n_clusters,
                                                                double* __restrict__ sums,
                                                                unsigned long long* __restrict__ counts) {
    const int64_t total = n * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / dim;
        const int j = (int)(i - row * dim);
        const int c = assign[row];
        if (c < 0 || c >= n_clusters) continue;
        atomicAdd(&sums[(int64_t)c * dim + j], (double)x[row * ldx + j]);
        if (j == 0) atomicAdd(&counts[c], 1ull);
    }
}

int launch_kmeans_accumulate(const float* x, int64_t n, int dim, int ldx, const int* assign, int n_clusters,
                             double* sums, unsigned long long* counts, hipStream_t s) {
    if (n <= 0) return VDB_OK;
    const int64_t total = n * dim;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(blocks), dim3(256), 0, s, x, n, dim, ldx, assign, n_clusters, sums, counts);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
}

}  // namespace vdb
