// probe (run on the GPU box): what does one wave-level global_load_dwordx4 cost the CU's texture-addresser / L1 path, by access pattern?
// 256 workgroups (one per CU) of W waves; every wave issues ITER x 16 independent dwordx4 loads (16 in flight) from a table and folds them
// into a checksum.  Reported: cycles per wave-level load per CU (= CU cycles / loads issued on that CU) and GB/s per CU.
// Patterns (lane l, load i):
//   same     every lane the same 16 B                                         (1 line per load)
//   quad     lanes 4g..4g+3 read one 64-B segment of a random texel            (16 segments per load: the march kernel's shape)
//   pair     lanes l and l+32 read two 16-B pieces of a random 64-B segment    (32 segments per load: the shade kernel's shape)
//   lane     every lane its own random 16 B                                    (64 segments per load)
//   padj     as pair, but the two lanes read ADJACENT 16-B pieces (one 32-B sector)
//   s32      16 B per lane at a stride of 32 B (2 x 1 KB regions per load: the shape of the round-1 basis-fragment layout)
//   cont     one contiguous, aligned 1 KB per load
// Table sizes: 16 KB (L1-resident), 16 MB (L2-resident across the chip), 64 MB (Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O2 ta_rate.hip -o ta_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 64

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int PAT>
__global__ __launch_bounds__(1024) void probe(const float4 *__restrict__ tab, unsigned n_seg, float *__restrict__ out, unsigned long long *__restrict__ cyc)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned key = (blockIdx.x * 16u + wave) * 65536u + (unsigned)(it * 16 + i);
            unsigned seg, piece;
            if (PAT == 0) { seg = hash32(key) % n_seg; piece = 0; }
            else if (PAT == 1) { seg = hash32(key * 16u + (lane >> 2)) % n_seg; piece = lane & 3u; }
            else if (PAT == 2) { seg = hash32(key * 32u + (lane & 31u)) % n_seg; piece = (lane >> 5) * 2u; }
            else if (PAT == 3) { seg = hash32(key * 64u + lane) % n_seg; piece = lane & 3u; }
            else if (PAT == 4) { seg = hash32(key * 32u + (lane & 31u)) % n_seg; piece = lane >> 5; }                       // pair, ADJACENT pieces 0 and 1
            else if (PAT == 5) { seg = (hash32(key) % (n_seg / 64u)) * 64u + (lane >> 5) * 16u + ((lane & 31u) >> 1); piece = (lane & 1u) * 2u; }   // 16 B at stride 32 B
            else { seg = (hash32(key) % (n_seg / 16u)) * 16u + (lane >> 2); piece = lane & 3u; }                             // one contiguous 1 KB
            v[i] = tab[seg * 4u + piece];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int PAT>
static void run(const char *name, const float4 *tab, size_t bytes, int waves, float *out, unsigned long long *cyc)
{
    hipMemset(cyc, 0, 256 * 8);
    const unsigned n_seg = (unsigned)(bytes / 64);
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(64 * waves), 0, 0, tab, n_seg, out, cyc);      // warm
    hipMemset(cyc, 0, 256 * 8);
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(64 * waves), 0, 0, tab, n_seg, out, cyc);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += (double)c; mean /= 256.0;
    const double loads = (double)waves * ITER * 16;
    printf("%-5s table %6.2f MB  waves/CU %2d : %6.1f cycles per wave-level load per CU, %5.1f B/clk/CU\n", name, bytes / 1048576.0, waves, mean / loads, loads * 1024.0 / mean);
}

int main()
{
    float *out; unsigned long long *cyc; float4 *tab;
    const size_t maxb = 64u << 20;
    hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 8); hipMalloc(&tab, maxb);
    hipMemset(tab, 0, maxb);
    for (size_t bytes : {(size_t)16384, (size_t)16 << 20, (size_t)64 << 20})
        for (int waves : {8, 16}) {
            run<0>("same", tab, bytes, waves, out, cyc);
            run<1>("quad", tab, bytes, waves, out, cyc);
            run<2>("pair", tab, bytes, waves, out, cyc);
            run<3>("lane", tab, bytes, waves, out, cyc);
            run<4>("padj", tab, bytes, waves, out, cyc);
            run<5>("s32", tab, bytes, waves, out, cyc);
            run<6>("cont", tab, bytes, waves, out, cyc);
        }
    return 0;
}
