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

// F16: the products on v_mfma_f32_32x32x16_f16 with both operands split into fp16 hi + lo (three products, fp32-grade; tvr_mfma.h) and dY multiplied by the
// power of two *scale first (unscaled again on the way out) — for callers that know their gradients' range at that scale.  W is split ONCE per workgroup into an
// LDS image of A fragments (k-step-major, hi and lo as separate 1 KB blocks: two ds_read_b128 per fragment); a 16-column step of dY is two float4 loads per lane.
// 96 MFMAs of 32 cycles per 128 x 128 tile instead of 256 of 64: the layer then runs at the rate of its three [M,128] streams.  An operand dY * scale or a
// result (before it is unscaled) at or beyond fp16's 65 504 raises *sat_flag: results are the next product's operands at the same scale.
template <int KB, bool F16>                                          // 32-column blocks of dX
__global__ __launch_bounds__(64 * LDX_WAVES, 2) void linear_dx_kernel(const float *__restrict__ dY, const int ldy, const int N, const float *__restrict__ W, const int ldw,
                                                                     const int n_valid, const float *__restrict__ mask, const int ldm, float *__restrict__ dX,
                                                                     const int ldx, const long long M, const float *__restrict__ scale, unsigned *__restrict__ sat_flag,
                                                                     const unsigned long long *__restrict__ mask_bits)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];       // fp32: [N][32 KB];  F16: uint4 [(N / 16) * KB * 2][64]
    constexpr int K = 32 * KB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    if (F16) {
        uint4 *img = reinterpret_cast<uint4 *>(wl);
        const int items = (N / 16) * KB * 64;
        for (int it = threadIdx.x; it < items; it += 64 * LDX_WAVES) {
            const int l = it & 63, f = it >> 6, kb = f % KB, st = f / KB;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = 16 * st + 8 * (l >> 5) + e;
                v[e] = n < n_valid ? W[(size_t)n * ldw + 32 * kb + (l & 31)] : 0.0f;
            }
            const Frag fr = split8(v);
            img[(f * 2 + 0) * 64 + l] = fr.hi;
            img[(f * 2 + 1) * 64 + l] = fr.lo;
        }
    } else {
        for (int e = threadIdx.x; e < N * K; e += 64 * LDX_WAVES) {
            const int n = e / K, k = e - n * K;
            wl[e] = n < n_valid ? W[(size_t)n * ldw + k] : 0.0f;
        }
    }
    __syncthreads();
    const float sc = (F16 && scale) ? *scale : 1.0f, inv_sc = 1.0f / sc;
    int bad = 0;
    const long long n_tiles = (M + 31) / 32;
    for (long long tile = (long long)blockIdx.x * LDX_WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * LDX_WAVES) {
        const long long s = tile * 32 + j, sr = s < M ? s : M - 1;
        f32x16 acc[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc[kb] = f32x16{0};
        if (F16) {
            const uint4 *img = reinterpret_cast<const uint4 *>(wl) + lane;
            const float *__restrict__ row = dY + sr * ldy + 8 * h;
            float4 c0 = *(const float4 *)(row), c1 = *(const float4 *)(row + 4);
            for (int st = 0; st < N / 16; ++st) {
                const float v[8] = {c0.x * sc, c0.y * sc, c0.z * sc, c0.w * sc, c1.x * sc, c1.y * sc, c1.z * sc, c1.w * sc};
                // (the round-toward-zero split saturates at 65 504 instead of overflowing: an operand at the limit must be SAID, it would be clipped silently)
                const float am = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
                bad |= (int)!(am < 65504.0f);
                if (st + 1 < N / 16) { c0 = *(const float4 *)(row + 16 * (st + 1)); c1 = *(const float4 *)(row + 16 * (st + 1) + 4); }    // next step's columns in flight
                const Frag fb = split8(v);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const uint4 ah = img[((st * KB + kb) * 2 + 0) * 64], al = img[((st * KB + kb) * 2 + 1) * 64];
                    acc[kb] = MFMAH(al, fb.hi, acc[kb]);
                    acc[kb] = MFMAH(ah, fb.lo, acc[kb]);
                    acc[kb] = MFMAH(ah, fb.hi, acc[kb]);
                }
            }
        } else {
            const float *__restrict__ row = dY + sr * ldy + 4 * h;
            for (int c0 = 0; c0 < N / 8; c0 += 8) {                  // 8 chunks (64 reduction indices) of dY in registers at a time
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
        }
        if (s < M) {
            // the mask as bits: this lane's 16 KB results are bits 16 kb + 4 q + i of word h of its sample (the order of the kernel that wrote them)
            const unsigned long long mbits = mask_bits ? mask_bits[2 * s + h] : ~0ull;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 32 * kb + 8 * q + 4 * h;
                    float4 v = make_float4(acc[kb][4 * q] * inv_sc, acc[kb][4 * q + 1] * inv_sc, acc[kb][4 * q + 2] * inv_sc, acc[kb][4 * q + 3] * inv_sc);
                    // the result is the next product's operand at the same scale (the layer below, the weight gradient): it must fit fp16 there as well
                    if (F16) bad |= (int)!(fmaxf(fmaxf(fabsf(acc[kb][4 * q]), fabsf(acc[kb][4 * q + 1])), fmaxf(fabsf(acc[kb][4 * q + 2]), fabsf(acc[kb][4 * q + 3]))) < 65504.0f);
                    if (mask_bits) {
                        const unsigned b4 = (unsigned)(mbits >> (16 * kb + 4 * q)) & 15u;
                        v.x = (b4 & 1u) ? v.x : 0.0f; v.y = (b4 & 2u) ? v.y : 0.0f; v.z = (b4 & 4u) ? v.z : 0.0f; v.w = (b4 & 8u) ? v.w : 0.0f;
                    } else if (mask) {
                        const float4 mk = *(const float4 *)(mask + s * ldm + k);
                        v.x = mk.x > 0.0f ? v.x : 0.0f; v.y = mk.y > 0.0f ? v.y : 0.0f; v.z = mk.z > 0.0f ? v.z : 0.0f; v.w = mk.w > 0.0f ? v.w : 0.0f;
                    }
                    *(float4 *)(dX + s * ldx + k) = v;
                }
        }
    }
    if (F16 && sat_flag && __ballot(bad != 0) != 0ull && lane == 0) atomicOr(sat_flag, 1u);
}

hipError_t launch_linear_dx(const float *dY, int ldy, int N, const float *W, int ldw, int n_valid, int K, const float *mask, int ldm, float *dX, int ldx, long long M,
                            hipStream_t stream, const float *scale, unsigned *sat_flag, const unsigned long long *mask_bits)
{
    const bool f16 = scale != nullptr && !(N & 15);
    const size_t lds = (size_t)N * K * sizeof(float);               // (the fragment image of the fp16 form has the same size)
    const long long n_tiles = (M + 31) / 32;
    long long blocks = (n_tiles + LDX_WAVES - 1) / LDX_WAVES;
    if (blocks > 512) blocks = 512;                                  // two workgroups per CU
    if (blocks < 1) return hipSuccess;
#define LDX_GO(KB_, F16_)                                                                                                                         \
    do {                                                                                                                                          \
        hipError_t rc = hipFuncSetAttribute((const void *)linear_dx_kernel<KB_, F16_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        if (rc != hipSuccess) return rc;                                                                                                          \
        hipLaunchKernelGGL((linear_dx_kernel<KB_, F16_>), dim3((unsigned)blocks), dim3(64 * LDX_WAVES), lds, stream, dY, ldy, N, W, ldw, n_valid, mask, ldm, dX, ldx, M, \
                           scale, sat_flag, mask_bits);                                                                                           \
    } while (0)
    switch ((K / 32) * 2 + (f16 ? 1 : 0)) {
    case 2: LDX_GO(1, false); break;
    case 3: LDX_GO(1, true); break;
    case 4: LDX_GO(2, false); break;
    case 5: LDX_GO(2, true); break;
    case 6: LDX_GO(3, false); break;
    case 7: LDX_GO(3, true); break;
    case 8: LDX_GO(4, false); break;
    default: LDX_GO(4, true); break;
    }
#undef LDX_GO
    return hipGetLastError();
}
