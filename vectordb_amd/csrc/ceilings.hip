// Measured ceilings of the GPU the library runs on: what a large device-to-device copy and
// register-only matrix-core loops reach, for reporting beside the spec peaks that the roofline
// fractions are quoted against.  Nothing on the search path calls this file.
#include "flat_math.h"
#include <algorithm>
#include <cstdlib>

namespace vdb {

// Every wave runs `iters` rounds of four independent 32x32 accumulators (no memory traffic, no
// dependency between consecutive instructions of different accumulators); the result is stored so
// that the loop cannot be dropped.
template <bool BF16>
__global__ __launch_bounds__(256) void mfma_loop_kernel(float* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    const float x = 1.0f + (float)(threadIdx.x & 7) * 0.125f, y = 0.5f;
    bf16x8 xa, xb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { xa[i] = (short)(0x3f80 + (threadIdx.x & 3)); xb[i] = (short)0x3f00; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (BF16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The arithmetic of the direct-form scans (sum (q - x)^2 in NumPy's rounding: subtract, multiply and add are
// three separately rounded operations, no fma) on registers only: 16 independent packed accumulators per lane,
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_add_f32 per pair of elements.  What the vector ALUs reach on that mix is the
// ceiling of the IVF list scan and of the batched Manhattan scan.
typedef float pk2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void valu_direct_loop_kernel(float* out, int iters, float seed) {
    pk2 acc[16], x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = pk2{0.0f, 0.0f};
        x[i] = pk2{seed * (float)(threadIdx.x + i), seed * (float)(i + 1)};
    }
    pk2 q = pk2{seed, -seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const pk2 d = q - x[i];
            acc[i] = acc[i] + d * d;
        }
        q = q + pk2{1e-3f, 1e-3f};
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int time_valu_loop(float* out, int blocks, int iters, hipStream_t s, float* tops) {
    hipEvent_t e0, e1;
    VDB_HIP(hipEventCreate(&e0));
    VDB_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_direct_loop_kernel, dim3(blocks), dim3(256), 0, s, out, 16, 0.5f);
    VDB_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(valu_direct_loop_kernel, dim3(blocks), dim3(256), 0, s, out, iters, 0.5f);
    VDB_HIP(hipEventRecord(e1, s));
    VDB_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    VDB_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // per iteration and lane: 16 accumulators x 2 elements x 3 operations
    const double ops = (double)blocks * 256 * iters * 16.0 * 2.0 * 3.0;
    *tops = (float)(ops / (ms * 1e-3) / 1e12);
    return VDB_OK;
}

template <bool BF16>
static int time_mfma_loop(float* out, int blocks, int iters, hipStream_t s, float* tflops) {
    hipEvent_t e0, e1;
    VDB_HIP(hipEventCreate(&e0));
    VDB_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_loop_kernel<BF16>, dim3(blocks), dim3(256), 0, s, out, 16);          // warm-up
    VDB_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(mfma_loop_kernel<BF16>, dim3(blocks), dim3(256), 0, s, out, iters);
    VDB_HIP(hipEventRecord(e1, s));
    VDB_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    VDB_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // one instruction: 32 x 32 outputs x K products x 2 flop
    const double flop_per_inst = 2.0 * 32 * 32 * (BF16 ? 16 : 2);
    const double insts = (double)blocks * 4 /* waves */ * iters * 16.0;
    *tflops = (float)(insts * flop_per_inst / (ms * 1e-3) / 1e12);
    return VDB_OK;
}

int measure_ceilings(hipStream_t s, float* copy_gbs, float* mfma_f32_tflops, float* mfma_bf16_tflops, float* valu_direct_tops) {
    const size_t bytes = (size_t)1 << 30;
    void *a = nullptr, *b = nullptr;
    VDB_HIP(hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); set_error("out of memory in measure_ceilings"); return VDB_ERR_NOMEM; }
    int st = VDB_OK;
    auto body = [&]() -> int {
        VDB_HIP(hipMemsetAsync(a, 1, bytes, s));
        VDB_HIP(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, s));                       // warm-up
        hipEvent_t e0, e1;
        VDB_HIP(hipEventCreate(&e0));
        VDB_HIP(hipEventCreate(&e1));
        const int reps = 10;
        VDB_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) VDB_HIP(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, s));
        VDB_HIP(hipEventRecord(e1, s));
        VDB_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        VDB_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *copy_gbs = (float)(2.0 * bytes * reps / (ms * 1e-3) / 1e9);                            // bytes read + bytes written
        const int blocks = ctx().num_cus * 2;                                                   // 8 waves per CU, 2 per SIMD
        VDB_TRY(time_mfma_loop<false>(reinterpret_cast<float*>(a), blocks, 4096, s, mfma_f32_tflops));
        VDB_TRY(time_mfma_loop<true>(reinterpret_cast<float*>(a), blocks, 16384, s, mfma_bf16_tflops));
        // 4 waves per SIMD by default; VDB_CEIL_WAVES = 1 or 2 shows what fewer resident waves reach
        static const int env_waves = getenv("VDB_CEIL_WAVES") ? atoi(getenv("VDB_CEIL_WAVES")) : 4;
        VDB_TRY(time_valu_loop(reinterpret_cast<float*>(a), ctx().num_cus * std::max(1, env_waves), 4096, s, valu_direct_tops));
        return VDB_OK;
    };
    st = body();
    (void)hipStreamSynchronize(s);
    (void)hipFree(a);
    (void)hipFree(b);
    return st;
}

}  // namespace vdb
