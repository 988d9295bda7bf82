// probe (GPU box, round 5): does every XCD of an MI355X run at its own clock, and how far apart are they under a matrix-core load?
// The shade kernel's workgroups, given equal shares, finished 5 - 8 % apart by XCD (profiles/r05_shade_tail.txt); this probe takes the kernel out of the question.
// 256 workgroups (one per CU) x 4 waves (one per SIMD); each wave runs a dependent-free stream of MFMAs (MODE 0: v_mfma_f32_16x16x32_f16 on random operands, MODE 1:
// v_fma_f32 only, no matrix core) for a fixed number of iterations and reports s_memtime / s_memrealtime ticks (shader clock / 100 MHz) and the XCC_ID register.
// Output: per XCD (by XCC_ID, and the blockIdx % 8 it was reached with) the mean clock and the mean time to finish equal work.
// Build: hipcc --offload-arch=gfx950 -O2 xcd_clock.hip -o xcd_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(int iters, float *__restrict__ out, unsigned long long *__restrict__ rec)
{
    const unsigned seed = hash32(blockIdx.x * 256u + threadIdx.x + 1u);
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(((int)(hash32(seed + j) & 2047u) - 1024) * (1.0f / 512.0f));
        b[j] = (_Float16)(((int)(hash32(seed + 8 + j) & 2047u) - 1024) * (1.0f / 512.0f));
    }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f0 = 1.0f + (seed & 255u) * 1e-3f, f1 = f0 + 1.0f, f2 = f0 + 2.0f, f3 = f0 + 3.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                f0 = __builtin_fmaf(f0, 0.999f, 0.001f); f1 = __builtin_fmaf(f1, 0.999f, 0.002f);
                f2 = __builtin_fmaf(f2, 0.999f, 0.003f); f3 = __builtin_fmaf(f3, 0.999f, 0.004f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;       // HW_REG_XCC_ID (20), bits [3:0]
        rec[blockIdx.x * 4 + 0] = t1 - t0;
        rec[blockIdx.x * 4 + 1] = r1 - r0;
        rec[blockIdx.x * 4 + 2] = xcc;
        rec[blockIdx.x * 4 + 3] = r0;
    }
}

template <int MODE>
static void run(const char *what, int iters)
{
    const int nb = 256;
    float *out;
    unsigned long long *rec;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipMalloc(&rec, nb * 4 * sizeof(unsigned long long));
    for (int rep = 0; rep < 3; ++rep) {                       // the third launch is reported: the chip has settled into the load
        hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(256), 0, 0, iters, out, rec);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(nb * 4);
    hipMemcpy(h.data(), rec, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double clk[8] = {0}, us[8] = {0};
    int cnt[8] = {0}, mism = 0;
    for (int b = 0; b < nb; ++b) {
        const int x = (int)h[b * 4 + 2] & 7;
        if (x != (b & 7)) ++mism;
        clk[x] += 0.1 * (double)h[b * 4 + 0] / (double)h[b * 4 + 1];
        us[x] += (double)h[b * 4 + 1] / 100.0;
        ++cnt[x];
    }
    printf("%s (%d workgroups whose XCC_ID differs from blockIdx %% 8)\n", what, mism);
    printf("  XCD            :");
    for (int x = 0; x < 8; ++x) printf(" %8d", x);
    printf("\n  workgroups     :");
    for (int x = 0; x < 8; ++x) printf(" %8d", cnt[x]);
    printf("\n  clock GHz      :");
    for (int x = 0; x < 8; ++x) printf(" %8.3f", cnt[x] ? clk[x] / cnt[x] : 0.0);
    printf("\n  us for the work:");
    for (int x = 0; x < 8; ++x) printf(" %8.1f", cnt[x] ? us[x] / cnt[x] : 0.0);
    printf("\n");
    hipFree(out);
    hipFree(rec);
}

int main()
{
    run<0>("v_mfma_f32_16x16x32_f16 stream, one wave per SIMD, all 256 CUs", 60000);
    run<1>("v_fma_f32 stream (no matrix core), one wave per SIMD, all 256 CUs", 60000);
    return 0;
}
