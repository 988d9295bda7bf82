// probe (run on the GPU box): what does a no-return fp32 global atomic cost on MI355X, and in which unit — the wave-level instruction, the 64-B line
// request, or the dword?  A 64 MB float image (the size of a training step's gradient images), 256 workgroups x 16 waves (one workgroup per CU), every
// wave issues ITER x 8 atomic instructions whose 64 lanes are laid out as
//   L64x1  : 64 different random lines, one dword each
//   L16x4  : 16 random lines, 4 dwords each at a 16-B stride inside the line (a float4-per-lane layout issuing its .x / .y / .z / .w in turn)
//   L16x4c : 16 random lines, 4 CONSECUTIVE dwords each
//   L4x16  : 4 random lines, 16 consecutive dwords each (march_backward: one lane per density channel)
//   L1x64  : 1 random 256-B block, 64 consecutive dwords (4 lines)
//   same   : every lane of every wave of the chip adds into the SAME 64 dwords (contention)
// and, for comparison, the same shapes as plain stores.
// Build: hipcc --offload-arch=gfx950 -O2 atomic_rate.hip -o atomic_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 256
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

enum { L64x1, L16x4, L16x4c, L4x16, L1x64, SAME };

template <int SHAPE, bool STORE>
__global__ __launch_bounds__(1024) void probe(float *__restrict__ img, unsigned n_lines, unsigned long long *__restrict__ cyc)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned key = ((blockIdx.x * 16u + wave) * 4096u + (unsigned)(it * 8 + i)) * 64u;
            size_t off;                                                        // in dwords
            if (SHAPE == L64x1) off = (size_t)(hash32(key + lane) % n_lines) * 16u + (lane & 15u);
            else if (SHAPE == L16x4) off = (size_t)(hash32(key + (lane >> 2)) % n_lines) * 16u + (lane & 3u) * 4u + (unsigned)(i & 3);
            else if (SHAPE == L16x4c) off = (size_t)(hash32(key + (lane >> 2)) % n_lines) * 16u + (unsigned)(i & 3) * 4u + (lane & 3u);
            else if (SHAPE == L4x16) off = (size_t)(hash32(key + (lane >> 4)) % n_lines) * 16u + (lane & 15u);
            else if (SHAPE == L1x64) off = (size_t)(hash32(key) % (n_lines / 4u)) * 64u + lane;
            else off = lane;
            if (STORE) img[off] = 1.0f;
            else atomicAdd(img + off, 1.0f);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

template <int SHAPE, bool STORE>
static void run(const char *name, float *img, unsigned n_lines, unsigned long long *cyc, int lines_per_instr)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cyc, 0, 256 * 8);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<SHAPE, STORE>), dim3(256), dim3(1024), 0, 0, img, n_lines, cyc);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = 256.0 * 16 * ITER * 8;
    printf("%-7s %-6s: %8.3f ms  %7.2f G instr/s  %7.1f G line-requests/s  %8.1f G dwords/s\n", name, STORE ? "store" : "atomic", ms, instr / ms * 1e-6,
           instr * lines_per_instr / ms * 1e-6, instr * 64 / ms * 1e-6);
}

int main()
{
    float *img; unsigned long long *cyc;
    const size_t bytes = 64u << 20;
    (void)hipMalloc(&img, bytes); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(img, 0, bytes);
    const unsigned n_lines = (unsigned)(bytes / 64);
    run<L64x1, false>("L64x1", img, n_lines, cyc, 64);
    run<L16x4, false>("L16x4", img, n_lines, cyc, 16);
    run<L16x4c, false>("L16x4c", img, n_lines, cyc, 16);
    run<L4x16, false>("L4x16", img, n_lines, cyc, 4);
    run<L1x64, false>("L1x64", img, n_lines, cyc, 4);
    run<SAME, false>("same", img, n_lines, cyc, 4);
    run<L64x1, true>("L64x1", img, n_lines, cyc, 64);
    run<L16x4, true>("L16x4", img, n_lines, cyc, 16);
    run<L4x16, true>("L4x16", img, n_lines, cyc, 4);
    run<L1x64, true>("L1x64", img, n_lines, cyc, 4);
    return 0;
}
