// probe (run on the GPU box): what clock does the chip hold under a dense MFMA stream, and what does that make of the 2.5 PFLOP/s peak?
// 256 workgroups of W waves issue back-to-back v_mfma_f32_32x32x16_f16 (two independent accumulators per wave) for a few milliseconds;
// wall time by HIP events, cycles by s_memtime.  Operands: constant 1.0 ("ones") or pseudo-random fp16 bit patterns ("random": the
// switching activity of real data).  Build: hipcc --offload-arch=gfx950 -O2 mfma_clock.hip -o mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int VALU_PER_MFMA>
__global__ void burn(int iters, int random, float *sink, unsigned long long *cyc)
{
    f32x16 a0 = {0}, a1 = {0};
    h8 x, y;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        const float fx = random ? (float)((int)(s >> 8) % 2001 - 1000) * 1e-3f : 1.0f;
        s = s * 1664525u + 1013904223u;
        const float fy = random ? (float)((int)(s >> 8) % 2001 - 1000) * 1e-3f : 1.0f;
        x[j] = (_Float16)fx; y[j] = (_Float16)fy;
    }
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            if (VALU_PER_MFMA >= 1) v0 = __builtin_fmaf(v0, 1.0001f, v1);
            if (VALU_PER_MFMA >= 2) v1 = __builtin_fmaf(v1, 0.9999f, v2);
            if (VALU_PER_MFMA >= 3) v2 = __builtin_fmaf(v2, 1.0001f, v3);
            if (VALU_PER_MFMA >= 4) v3 = __builtin_fmaf(v3, 0.9999f, v0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
            if (VALU_PER_MFMA >= 1) v0 = __builtin_fmaf(v0, 1.0001f, v1);
            if (VALU_PER_MFMA >= 2) v1 = __builtin_fmaf(v1, 0.9999f, v2);
            if (VALU_PER_MFMA >= 3) v2 = __builtin_fmaf(v2, 1.0001f, v3);
            if (VALU_PER_MFMA >= 4) v3 = __builtin_fmaf(v3, 0.9999f, v0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a0[0] + a1[0] + v0 + v1 + v2 + v3 == 12345.678f) sink[0] = a0[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
static void run(const char *name, int waves, int random, float *sink, unsigned long long *cyc)
{
    const int iters = 20000;                      // 320 000 MFMAs per wave: ~5 ms at one wave per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((burn<V>), dim3(256), dim3(64 * waves), 0, 0, iters / 10, random, sink, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((burn<V>), dim3(256), dim3(64 * waves), 0, 0, iters, random, sink, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfmas = 256.0 * waves * iters * 16.0;
    printf("%-34s waves/CU %2d %-6s: %7.3f ms, %6.1f cycles per MFMA and wave, clock %.2f GHz, %6.0f TFLOP/s (%.2f of 2500)\n", name, waves, random ? "random" : "ones", ms,
           (double)c / (iters * 16.0), (double)c / (ms * 1e6), mfmas * 32768.0 / (ms * 1e-3) / 1e12, mfmas * 32768.0 / (ms * 1e-3) / 2.5e15);
}

int main()
{
    float *sink; unsigned long long *cyc;
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&cyc, 8);
    for (int random : {0, 1}) {
        run<0>("MFMA only", 4, random, sink, cyc);
        run<0>("MFMA only", 8, random, sink, cyc);
        run<2>("MFMA + 2 dependent VALU each", 4, random, sink, cyc);
        run<4>("MFMA + 4 dependent VALU each", 4, random, sink, cyc);
        run<4>("MFMA + 4 dependent VALU each", 8, random, sink, cyc);
    }
    return 0;
}
