// probe: semantics of v_permlane16_swap / v_permlane32_swap on gfx950 (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o)
{
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[4] = {"p16.r0", "p16.r1", "p32.r0", "p32.r1"};
    for (int v = 0; v < 4; ++v) { printf("%s:", nm[v]); for (int i = 0; i < 64; ++i) printf(" %u", h[v * 64 + i]); printf("\n"); }
    return 0;
}
