// probe (run on the GPU box): does the vector L1 -> register path deliver more bytes per clock with narrower loads?  16-KB table (L1-resident),
// 256 workgroups (one per CU) of W waves, every wave issues ITER x 16 independent loads of 4 / 8 / 16 B per lane; "cont" = the wave reads one
// contiguous aligned block, "rand" = every 32-lane half reads 16-B-aligned pieces of 32 random 64-B segments (the shade kernel's shape).
// Build: hipcc --offload-arch=gfx950 -O2 l1_width.hip -o l1_width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 64
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <typename T, int RAND>
__global__ __launch_bounds__(1024) void probe(const unsigned char *__restrict__ tab, unsigned n_seg, float *__restrict__ out, unsigned long long *__restrict__ cyc)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        T v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned key = (blockIdx.x * 16u + wave) * 65536u + (unsigned)(it * 16 + i);
            unsigned off;
            if (RAND) off = (hash32(key * 32u + (lane & 31u)) % n_seg) * 64u + (lane >> 5) * 32u;
            else off = (hash32(key) % (n_seg / 16u)) * 1024u + lane * (unsigned)sizeof(T);
            v[i] = *(const T *)(tab + off);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            for (unsigned c = 0; c < sizeof(T) / 4; ++c) acc += ((const float *)&v[i])[c];        // every component is used: the load keeps its width
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
    if (acc == 12345.678f) out[0] = acc;
}

template <typename T, int RAND>
static void run(const char *name, const unsigned char *tab, int waves, float *out, unsigned long long *cyc)
{
    const unsigned n_seg = 16384 / 64;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cyc, 0, 256 * 8);
        hipLaunchKernelGGL((probe<T, RAND>), dim3(256), dim3(64 * waves), 0, 0, tab, n_seg, out, cyc);
    }
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += (double)c; mean /= 256.0;
    const double loads = (double)waves * ITER * 16;
    printf("%-26s waves/CU %2d : %6.1f cycles per wave-level load per CU, %5.1f B/clk/CU\n", name, waves, mean / loads, loads * 64.0 * sizeof(T) / mean);
}

int main()
{
    float *out; unsigned long long *cyc; unsigned char *tab;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&tab, 1 << 20);
    (void)hipMemset(tab, 0, 1 << 20);
    for (int waves : {8, 16}) {
        run<float, 0>("dword   contiguous", tab, waves, out, cyc);
        run<float2, 0>("dwordx2 contiguous", tab, waves, out, cyc);
        run<float4, 0>("dwordx4 contiguous", tab, waves, out, cyc);
        run<float, 1>("dword   32 random segments", tab, waves, out, cyc);
        run<float2, 1>("dwordx2 32 random segments", tab, waves, out, cyc);
        run<float4, 1>("dwordx4 32 random segments", tab, waves, out, cyc);
    }
    return 0;
}
