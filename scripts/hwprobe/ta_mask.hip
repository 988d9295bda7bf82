// probe (GPU box, round 5): does a wave-level global_load_dwordx4 cost the texture-addresser / L1 path per ACTIVE QUAD or per instruction?
// The march kernel's gather is quad-per-sample (lanes 4g..4g+3 read one 64-B texel): 48 wave-level loads per 64-sample chunk, ~21 cycles each on the CU's
// vector-memory path, which bounds it.  A quad's four sub-steps serve CONSECUTIVE samples (half a voxel apart), so about half of its
// (sample, plane) fetches hit the cell of the previous sub-step: if an exec-masked quad costs nothing, the kernel can keep the 2x2 window in registers
// and skip those fetches.  This probe answers that before the kernel is touched.
// 256 workgroups (one per CU) x 16 waves; every wave issues ITER x 16 independent dwordx4 loads (quad pattern, random texels of an L2-resident 16 MB table — the
// density planes are 17 MB) with K of its 16 quads active: K = 16, 12, 8 (even quads), 8 (random per load), 4, 1.
// Build: hipcc --offload-arch=gfx950 -O2 ta_mask.hip -o ta_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 64

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE: 0 all 16 quads; 1 quads with (q % 4) != 3 (12); 2 even quads (8); 3 a random half per load (8 on average); 4 q % 4 == 0 (4); 5 quad 0 only (1);
//       6 = mode 3 with the SAME texel per wave-load for all quads (L1 hits: isolates the instruction's own cost)
template <int MODE>
__global__ __launch_bounds__(1024) void probe(const float4 *__restrict__ tab, unsigned n_seg, float *__restrict__ out, unsigned long long *__restrict__ cyc)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, q = lane >> 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned key = (blockIdx.x * 16u + wave) * 65536u + (unsigned)(it * 16 + i);
            const unsigned seg = (MODE == 6 ? hash32(key) : hash32(key * 16u + q)) % n_seg;
            bool on = true;
            if (MODE == 1) on = (q & 3u) != 3u;
            else if (MODE == 2) on = (q & 1u) == 0u;
            else if (MODE == 3 || MODE == 6) on = (hash32(key * 31u + q * 7u + 1u) & 1u) != 0u;
            else if (MODE == 4) on = (q & 3u) == 0u;
            else if (MODE == 5) on = q == 0u;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on) v[i] = tab[seg * 4u + (lane & 3u)];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int MODE>
static void run(const char *name, const float4 *tab, size_t bytes, int waves, float *out, unsigned long long *cyc)
{
    const unsigned n_seg = (unsigned)(bytes / 64);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves), 0, 0, tab, n_seg, out, cyc);      // warm
    (void)hipMemset(cyc, 0, 256 * 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves), 0, 0, tab, n_seg, out, cyc);
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += (double)c; mean /= 256.0;
    const double loads = (double)waves * ITER * 16;
    printf("%-34s table %6.2f MB  waves/CU %2d : %6.1f cycles per wave-level load per CU\n", name, bytes / 1048576.0, waves, mean / loads);
}

int main()
{
    float *out; unsigned long long *cyc; float4 *tab;
    const size_t maxb = 16u << 20;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&tab, maxb);
    (void)hipMemset(tab, 0, maxb);
    for (size_t bytes : {(size_t)16384, (size_t)16 << 20}) {
        run<0>("16 of 16 quads active", tab, bytes, 16, out, cyc);
        run<1>("12 of 16 (q % 4 != 3)", tab, bytes, 16, out, cyc);
        run<2>(" 8 of 16 (even quads)", tab, bytes, 16, out, cyc);
        run<3>(" 8 of 16 (random half per load)", tab, bytes, 16, out, cyc);
        run<4>(" 4 of 16 (q % 4 == 0)", tab, bytes, 16, out, cyc);
        run<5>(" 1 of 16", tab, bytes, 16, out, cyc);
        run<6>(" 8 of 16 random, one texel per load", tab, bytes, 16, out, cyc);
    }
    return 0;
}
