// probe (run on the GPU box): can a global load issued BEHIND queued MFMAs overwrite their A operand before they read it?
// Each wave repeats: N back-to-back MFMAs reading A = x (all halves 1.0), B = y (1.0), then a global load of 2.0s INTO x, wait, drain, restore.
// Without a hazard every accumulator element ends at ITER * N * 16.  Build: hipcc --offload-arch=gfx950 -O2 mfma_war.hip -o mfma_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define ITER 500

template <int N, int NOPS, int SAME, int OPB>
__global__ void probe(const u32x4 *__restrict__ poison, float *__restrict__ out, int *__restrict__ bad)
{
    const u32x4 one = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};      // 8 x fp16 1.0
    u32x4 x = one, y = one;
    f32x16 acc = {0};
    const u32x4 *p = poison + (SAME ? 0 : (blockIdx.x * blockDim.x + threadIdx.x) % 4096);   // SAME: one L1-resident line for every lane (fastest return)
    for (int it = 0; it < ITER; ++it) {
        if (N == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
        if (N == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#define M4 "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n"
        if (N == 4) asm volatile(M4 : "+v"(acc) : "v"(x), "v"(y));
        if (N == 16) asm volatile(M4 M4 M4 M4 : "+v"(acc) : "v"(x), "v"(y));
        if (N == 32) asm volatile(M4 M4 M4 M4 M4 M4 M4 M4 : "+v"(acc) : "v"(x), "v"(y));
        if (NOPS == 1) asm volatile("s_nop 7");
        if (NOPS == 2) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
        // the load's destination IS the A operand of the MFMAs just issued
        if (OPB) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "+v"(y) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "+v"(x) : "v"(p) : "memory");
        // drain (compiler-visible read of the accumulator), then restore x
        float d = acc[15];
        asm volatile("" :: "v"(d));
        x = one; y = one;
        asm volatile("" : "+v"(x), "+v"(y));
    }
    const float want = (float)ITER * N * 16.0f;
    int wrong = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != want);
    if (wrong) atomicAdd(bad, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = acc[0];
}

template <int N, int NOPS, int SAME = 0, int OPB = 0>
static void run(const char *tag, int waves_per_block, const u32x4 *poison, float *out, int *bad)
{
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((probe<N, NOPS, SAME, OPB>), dim3(256), dim3(64 * waves_per_block), 0, 0, poison, out, bad);
    int h = 0; float o = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d  N %d  lanes with a wrong accumulator: %7d of %d   (acc[0] = %.0f, want %.0f)\n", tag, waves_per_block / 4, N, h,
           256 * 64 * waves_per_block, o, (float)ITER * N * 16.0f);
}

int main()
{
    std::vector<unsigned> hp(4096 * 4, 0x40004000u);                              // fp16 2.0 everywhere
    u32x4 *poison; float *out; int *bad;
    hipMalloc(&poison, hp.size() * 4); hipMalloc(&out, 64); hipMalloc(&bad, 4);
    hipMemcpy(poison, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    for (int w : {4, 8, 16}) {
        run<1, 0>("1 MFMA, load right behind", w, poison, out, bad);
        run<2, 0>("2 MFMAs, load right behind", w, poison, out, bad);
        run<4, 0>("4 MFMAs, load right behind", w, poison, out, bad);
        run<16, 0>("16 MFMAs, load right behind", w, poison, out, bad);
        run<32, 0>("32 MFMAs, load right behind", w, poison, out, bad);
        run<32, 1>("32 MFMAs, s_nop 7, load", w, poison, out, bad);
        run<32, 2>("32 MFMAs, 64 nop cycles, load", w, poison, out, bad);
        run<32, 0, 1, 0>("32 MFMAs, same-line load into A", w, poison, out, bad);
        run<32, 0, 1, 1>("32 MFMAs, same-line load into B", w, poison, out, bad);
        run<4, 0, 1, 1>("4 MFMAs, same-line load into B", w, poison, out, bad);
    }
    return 0;
}
