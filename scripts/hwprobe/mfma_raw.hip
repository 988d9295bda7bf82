// probe (run on the GPU box): is "VALU writes a VGPR, the very next instruction is an MFMA reading it as A" interlocked by the hardware,
// or does it need software wait states (which the compiler inserts for its own code but cannot for inline asm)?
// Each iteration sets dword 0 of x to 2.0 (fp16 x2) with a v_mov right in front of an MFMA, then back to 1.0 in front of the next one.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_raw.hip -o mfma_raw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ITER 1000
template <int NOPS>
__global__ void probe(float *__restrict__ out, int *__restrict__ bad)
{
    const u32x4 one = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    u32x4 y = one;
    f32x16 acc = {0};
    unsigned two = 0x40004000u, uno = 0x3C003C00u;
    for (int it = 0; it < ITER; ++it) {
        // explicit registers: v[100:103] is the A operand; its LAST dword is written by the instruction right in front of the MFMA
#define SET(v) "v_mov_b32 v100, " v "\n v_mov_b32 v101, " v "\n v_mov_b32 v102, " v "\n v_mov_b32 v103, " v "\n"
        if (NOPS == 0)
            asm volatile(SET("%2") "v_mfma_f32_32x32x16_f16 %0, v[100:103], %1, %0\n" SET("%3") "v_mfma_f32_32x32x16_f16 %0, v[100:103], %1, %0"
                         : "+v"(acc) : "v"(y), "v"(two), "v"(uno) : "v100", "v101", "v102", "v103");
        else
            asm volatile(SET("%2") "s_nop 3\n v_mfma_f32_32x32x16_f16 %0, v[100:103], %1, %0\n" SET("%3") "s_nop 3\n v_mfma_f32_32x32x16_f16 %0, v[100:103], %1, %0"
                         : "+v"(acc) : "v"(y), "v"(two), "v"(uno) : "v100", "v101", "v102", "v103");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = acc[0];
    // first MFMA: A = 2.0 everywhere -> 32 per element, second: 16
    const float want = (float)ITER * 48.0f;
    int wrong = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != want);
    if (wrong) atomicAdd(bad, 1);
}
int main()
{
    float *out; int *bad;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&bad, 4);
    for (int w : {4, 8}) {
        for (int nops = 0; nops < 2; ++nops) {
            (void)hipMemset(bad, 0, 4);
            if (nops == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(64 * w), 0, 0, out, bad);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(64 * w), 0, 0, out, bad);
            int h = 0; float o = 0;
            (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
            printf("v_mov -> MFMA, %s, waves/SIMD %d: lanes with a wrong accumulator %d of %d (acc[0] = %.0f, want %.0f)\n",
                   nops ? "s_nop 3 between" : "back to back", w / 4, h, 256 * 64 * w, o, (float)ITER * 48.0f);
        }
    }
    return 0;
}
