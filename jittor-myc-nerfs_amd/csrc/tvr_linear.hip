// tvr_linear.hip — the input gradient of a Linear over a tall batch, with the ReLU mask of the layer in front of it fused in:
//     dX[m, k] = (sum_n dY[m, n] W[n, k]) * (mask[m, k] > 0)            M ~ 2e6 rows, N <= 128 (multiple of 8), K in {32, 64, 96, 128}
// i.e. what autograd does for `relu(Linear(..)(x))` chains in the backward of NerfPlusPlus's background network (models/nerfplusplus.py:66-140 under
// `optimizer.backward`, train.py:258) — 2.1e6 samples per step, where the library's transposed GEMMs plus the separate mask kernels were a third of the step.
// fp32-input MFMAs (v_mfma_f32_32x32x2_f32: gradients need fp32's exponent range, no splitting), W [N, K] fp32 in LDS as it lies in memory (no transpose:
// the reduction runs over W's ROWS), one 32-sample tile per wave at a time.  Operand mapping (tvr_gemm.hip's): first operand lane (i, h) = the value for
// output row i at reduction index h, second operand lane (j, h) = the value for output column j; the reduction index is enumerated so that a lane's
// float4 of dY feeds four consecutive MFMAs: chunk c, step t  <->  n = 8 c + 4 h + t.
#include <hip/hip_runtime.h>
#include "tvr_kernels.h"
#include "tvr_mfma.h"

#define LDX_WAVES 8

template <int KB>                                                    // 32-column blocks of dX
__global__ __launch_bounds__(64 * LDX_WAVES, 2) void linear_dx_kernel(const float *__restrict__ dY, const int ldy, const int N, const float *__restrict__ W, const int ldw,
                                                                     const int n_valid, const float *__restrict__ mask, const int ldm, float *__restrict__ dX,
                                                                     const int ldx, const long long M)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [N][32 KB]
    constexpr int K = 32 * KB;
    for (int e = threadIdx.x; e < N * K; e += 64 * LDX_WAVES) {
        const int n = e / K, k = e - n * K;
        wl[e] = n < n_valid ? W[(size_t)n * ldw + k] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const long long n_tiles = (M + 31) / 32;
    for (long long tile = (long long)blockIdx.x * LDX_WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * LDX_WAVES) {
        const long long s = tile * 32 + j, sr = s < M ? s : M - 1;
        f32x16 acc[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc[kb] = f32x16{0};
        const float *__restrict__ row = dY + sr * ldy + 4 * h;
        for (int c0 = 0; c0 < N / 8; c0 += 8) {                      // 8 chunks (64 reduction indices) of dY in registers at a time
            float4 dy[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) dy[c] = (c0 + c) * 8 < N ? *(const float4 *)(row + 8 * (c0 + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if ((c0 + c) * 8 >= N) break;
                const float *a = wl + (size_t)(8 * (c0 + c) + 4 * h) * K + j;
                const float b[4] = {dy[c].x, dy[c].y, dy[c].z, dy[c].w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t * K + 32 * kb], b[t], acc[kb], 0, 0, 0);
            }
        }
        if (s < M) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 32 * kb + 8 * q + 4 * h;
                    float4 v = make_float4(acc[kb][4 * q], acc[kb][4 * q + 1], acc[kb][4 * q + 2], acc[kb][4 * q + 3]);
                    if (mask) {
                        const float4 mk = *(const float4 *)(mask + s * ldm + k);
                        v.x = mk.x > 0.0f ? v.x : 0.0f; v.y = mk.y > 0.0f ? v.y : 0.0f; v.z = mk.z > 0.0f ? v.z : 0.0f; v.w = mk.w > 0.0f ? v.w : 0.0f;
                    }
                    *(float4 *)(dX + s * ldx + k) = v;
                }
        }
    }
}

hipError_t launch_linear_dx(const float *dY, int ldy, int N, const float *W, int ldw, int n_valid, int K, const float *mask, int ldm, float *dX, int ldx, long long M,
                            hipStream_t stream)
{
    const size_t lds = (size_t)N * K * sizeof(float);
    const long long n_tiles = (M + 31) / 32;
    long long blocks = (n_tiles + LDX_WAVES - 1) / LDX_WAVES;
    if (blocks > 512) blocks = 512;                                  // two workgroups per CU
    if (blocks < 1) return hipSuccess;
#define LDX_GO(KB_)                                                                                                                               \
    do {                                                                                                                                          \
        hipError_t rc = hipFuncSetAttribute((const void *)linear_dx_kernel<KB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        if (rc != hipSuccess) return rc;                                                                                                          \
        hipLaunchKernelGGL(linear_dx_kernel<KB_>, dim3((unsigned)blocks), dim3(64 * LDX_WAVES), lds, stream, dY, ldy, N, W, ldw, n_valid, mask, ldm, dX, ldx, M); \
    } while (0)
    switch (K / 32) {
    case 1: LDX_GO(1); break;
    case 2: LDX_GO(2); break;
    case 3: LDX_GO(3); break;
    default: LDX_GO(4); break;
    }
#undef LDX_GO
    return hipGetLastError();
}
