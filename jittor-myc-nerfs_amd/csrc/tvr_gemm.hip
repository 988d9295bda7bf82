// tvr_gemm.hip — C[Ka x Kb] = A^T[Ka x M] * B[M x Kb] for tall-skinny operands (M = the ~3.5e5 appearance samples of a training batch,
// Ka, Kb <= 160): the weight gradients dW = dY^T X of the training step's Linears (tensorBase.py:69-71, tensoRF.py:150, train.py:258).
// The library GEMM picks 32x32 macro tiles with stream-K for these shapes and runs at ~15 TFLOP/s (0.9 ms per gradient, three per step);
// here every workgroup reduces its own slab of rows with fp32-input MFMAs (v_mfma_f32_32x32x2_f32, fp32 semantics, no splitting needed:
// the kernel is bound by the fp32 matrix rate, 0.1 ms for the largest of them) and adds its Ka x Kb partial into C with fp32 atomics.
//
// One workgroup per CU, one wave per SIMD (the requested LDS keeps a second workgroup off the CU): the MFMA operands come straight from
// global loads, and with a single MFMA-issuing wave per SIMD the load-behind-MFMA hazard described in tvr_shade.hip cannot occur.
#include <hip/hip_runtime.h>
#include "tvr_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TG_WAVES 4
#define TG_MAXT 5                 // 32x32 output tiles per wave (20 per workgroup: 128 x 160)
#define TG_UNROLL 4               // 2-row steps in flight

__global__ __launch_bounds__(64 * TG_WAVES) void gemm_tn_kernel(const float *__restrict__ A, const int lda, const int Ka,
                                                                 const float *__restrict__ B, const int ldb, const int Kb,
                                                                 const long long M, float *__restrict__ C, const long long rows_per_block)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, k = lane >> 5;
    const int nrb = (Ka + 31) >> 5, ncb = (Kb + 31) >> 5, ntiles = nrb * ncb;
    // this wave's tiles t = wave, wave + 4, ...: row offsets into A / B for this lane, or -1 past the edge
    int offA[TG_MAXT], offB[TG_MAXT];
    f32x16 acc[TG_MAXT];
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        const int t = wave + q * TG_WAVES;
        const int rb = t / ncb, cb = t - rb * ncb;
        offA[q] = (t < ntiles && rb * 32 + i < Ka) ? rb * 32 + i : -1;
        offB[q] = (t < ntiles && cb * 32 + i < Kb) ? cb * 32 + i : -1;
        acc[q] = f32x16{0};
    }
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    for (long long m = m0; m < m1; m += 2 * TG_UNROLL) {
        float a[TG_UNROLL][TG_MAXT], b[TG_UNROLL][TG_MAXT];
#pragma unroll
        for (int u = 0; u < TG_UNROLL; ++u) {
            const long long row = m + 2 * u + k;
            const bool in = row < m1;
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) {
                a[u][q] = (in && offA[q] >= 0) ? A[row * lda + offA[q]] : 0.0f;
                b[u][q] = (in && offB[q] >= 0) ? B[row * ldb + offB[q]] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < TG_UNROLL; ++u)
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], b[u][q], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        const int t = wave + q * TG_WAVES;
        if (t >= ntiles) continue;
        const int rb = t / ncb, cb = t - rb * ncb;
        const int col = cb * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
            if (row < Ka && col < Kb) atomicAdd(C + (size_t)row * Kb + col, acc[q][r]);
        }
    }
}

hipError_t launch_gemm_tn(const float *A, int lda, int Ka, const float *B, int ldb, int Kb, long long M, float *C, hipStream_t stream)
{
    hipError_t rc = hipMemsetAsync(C, 0, (size_t)Ka * Kb * sizeof(float), stream);
    if (rc != hipSuccess || M <= 0) return rc;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    long long grid = cus;
    long long rpb = ((M + grid - 1) / grid + 7) / 8 * 8;           // multiple of the 8-row unrolled step
    if (rpb < 64) rpb = 64;
    grid = (M + rpb - 1) / rpb;
    const int lds = 96 * 1024;                                     // unused; keeps the CU to one workgroup (see header)
    rc = hipFuncSetAttribute((const void *)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)grid), dim3(64 * TG_WAVES), lds, stream, A, lda, Ka, B, ldb, Kb, M, C, rpb);
    return hipGetLastError();
}
