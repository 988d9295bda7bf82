// probe (GPU box): in the power-limited regime (all 256 CUs issuing dense f16 MFMAs on random operands), which MFMA shape delivers more FLOP/s?
// 256 workgroups x 4 waves (one per SIMD); each wave issues back-to-back MFMAs on 4 accumulators for a few ms.  v_mfma_f32_32x32x16_f16 (32 768 FLOP, 32 cycles)
// against v_mfma_f32_16x16x32_f16 (16 384 FLOP, 16 cycles).  Also with one ds_read_b128 per 32 cycles of matrix pipe (the shade kernel's fragment traffic) and on
// half the chip.  Build: hipcc --offload-arch=gfx950 -O2 mfma_shape_power.hip -o mfma_shape_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h8 rnd8(unsigned &s)
{
    h8 x;
    for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; x[j] = (_Float16)((float)((int)(s >> 8) % 2001 - 1000) * 1e-3f); }
    return x;
}
template <int SHAPE, int DS>
__global__ __launch_bounds__(256) void burn(int iters, float *sink, unsigned long long *cyc)
{
    __shared__ u32x4 lds[1024];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = u32x4{s, s * 3u, s * 5u, s * 7u};
    __syncthreads();
    h8 a = rnd8(s), b = rnd8(s);
    f32x16 c0 = {0}, c1 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    u32x4 q = {0, 0, 0, 0};
    const unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                if (DS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(la), "n"(0));
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
                if (DS) asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(1)" : "=v"(q) : "v"(la), "n"(1024));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d1, 0, 0, 0);
                if (DS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(la), "n"(0));
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d3, 0, 0, 0);
                if (DS) asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(1)" : "=v"(q) : "v"(la), "n"(1024));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (c0[0] + c1[0] + d0[0] + d1[0] + d2[0] + d3[0] + __builtin_bit_cast(float, q[0]) == 12345.678f) sink[0] = c0[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SHAPE, int DS>
static void run(int blocks, float *sink, unsigned long long *cyc)
{
    const int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((burn<SHAPE, DS>), dim3(blocks), dim3(256), 0, 0, iters / 10, sink, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((burn<SHAPE, DS>), dim3(blocks), dim3(256), 0, 0, iters, sink, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * 4 * iters * 8.0 * 32768.0;          // 8 x 32x32x16 or 16 x 16x16x32 per iteration
    printf("%2dx%2d  %s  %3d CUs: %7.3f ms, clock %.2f GHz, %6.0f TFLOP/s\n", SHAPE, SHAPE, DS ? "+ 1 ds_read_b128 per 32 pipe cycles" : "MFMA only                          ", blocks, ms,
           (double)c / (ms * 1e6), flop / (ms * 1e-3) / 1e12);
}
int main()
{
    float *sink; unsigned long long *cyc;
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 0>(256, sink, cyc); run<16, 0>(256, sink, cyc);
        run<32, 1>(256, sink, cyc); run<16, 1>(256, sink, cyc);
        run<32, 0>(128, sink, cyc); run<16, 0>(128, sink, cyc);
    }
    return 0;
}
