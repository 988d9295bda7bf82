// tvr_march.hip — kernel 1 of the render path: ray/AABB entry + uniform (or jittered) sampling + in-box / alpha
// mask + VM density lookup + softplus + alpha + front-to-back transmittance scan, fused.
//
// Work it replaces in the reference (paths relative to /root/reference/tensorf-myc/):
//   models/tensorBase.py:340-360 sample_ray, :491-496 alpha-mask merge, :503 normalize_coord,
//   models/tensoRF.py:209-225 compute_densityfeature, tensorBase.py:444-448 feature2density,
//   :17-24 raw2alpha, :513 app_mask, :520 acc_map, :530-531 depth_map.
//
// Launch shape: persistent workgroups, one per CU (up to 16 waves), each owning every (gridDim)-th 16-ray tile of its XCD's contiguous
// tile range; the waves of a group take rays from an LDS cursor, so the 16 rays of a tile are marched concurrently (shared texels
// in L1) and no wave idles while the group has rays left.  The three density LINES (3 x (L+1) x 64 B, 58 KB at 300^3) are copied
// into LDS once per group: a third of the gather's 64-B requests then go to LDS instead of the L1/TA path that bounds this kernel.
//
// Mapping (wave64): one wave marches one ray, 64 consecutive samples per chunk.  Position / mask / index math is
// lane-per-sample.  The density gather is quad-per-sample: 4 lanes each fetch one float4 (4 of the 16 channels) of
// every texel, so a wave-level dwordx4 load covers 16 samples x 64 contiguous bytes (one texel per quad) — coalesced
// 64-B segments instead of 64 unrelated 16-B pieces.  Four sub-steps cover the 64 samples (sub-step k serves
// samples 4g+k), so after the quad reduction lane 4g+k owns sample 4g+k again with no permutation.
// The transmittance is a 6-step wave scan (DPP/shuffle) with the carry in a register; the ray stops once T < eps_T
// (eps_T <= weight threshold, so no appearance sample is ever skipped).  Appearance samples (w > thres) are compacted
// with a ballot into a per-wave LDS list and flushed once per ray into a contiguous, sample-ordered segment of the
// global queue (one atomicAdd per ray) -> the compositing order per ray is fixed, results are deterministic.
#include <cstdlib>
#include "tvr_device.h"
#include "tvr_kernels.h"

#ifndef TVR_MARCH_RASTER
#define TVR_MARCH_RASTER 1      // 1: every group sweeps the image together (tile k*grid + group) -> the queue is in ~raster order; measured march 7.9 vs 8.8 ms and shade 15.25 vs 15.65 ms against per-XCD contiguous bands (0)
#endif
// Small launches (a 4096-ray training batch or render chunk, a rank's share of a split frame: 16 rays per CU, one per wave) — measured in round 3
// (scripts/march_timeline.py, profiles/r03_march_timeline.txt): a wave's chain takes 10.4 us per 64-sample chunk at 4096 rays, 8.3 at 16 384,
// 7.2 in the full frame; the kernel ends 80 us after it starts where its share of a full frame is 49 us.  Tried and dropped: staggering the
// waves' starts over a chunk period (no change: it is not a lock-step effect) and touching the next chunk's 12 plane texels per sample one
// chunk ahead (12.1 us per chunk, 32.8 vs 30.2 ms per 157-call frame: the extra loads cost more than the misses they hide).  What DID cost a
// small launch 100 us was this kernel's own statistics: three same-address global atomics per wave (4096 waves) — now one per group.
#define MARCH_MAX_WAVES 16
#define MARCH_TILE 16                     // rays per tile
#define MARCH_SPIN_LIMIT (1u << 22)       // s_sleep(1) each: ~0.15 s, against a legitimate wait of microseconds
#define MARCH_SLOTS 64u                   // ring of published handouts (a waiter's slot is reused 64 local tiles = 1024 cursor draws later)
#ifndef MARCH_CTR_WORD
#define MARCH_CTR_WORD 1                  // word of the scratch header that counts handed-out rays (its own 128-B line, word 32, measured the same: 7.55 vs 7.58 ms)
#endif
#ifndef MARCH_TAIL
#define MARCH_TAIL 4u                     // rays per handout at the end of a launch (16u = off)
#endif
#define MARCH_HDR 560                     // LDS header: ray cursor (16 B) + 64 slots of {local tile number + 1 | tail bit, first ray} (dynamic queue) + 3 u64 statistics sums + pad
#ifndef TVR_MARCH_DYN
#define TVR_MARCH_DYN 1                   // 1: workgroups take 16-ray tiles from ONE global counter (in order), not a fixed stride: no tail when a launch has few tiles per group
#endif
#ifndef MARCH_LSTRIDE
#define MARCH_LSTRIDE 4                   // float4 per line texel in LDS.  4 = packed; 5 (80 B: the texels of 16 consecutive cells in distinct banks) removes
#endif                                    // the line taps' bank conflicts (34 % of the LDS-active cycles) and measures SLOWER: 8.2 vs 8.0 ms — the kernel sits on the L1 path

// quad-level data movement as DPP VALU ops (quad_perm) instead of ds_bpermute: no LDS hop in front of the gather addresses
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float quad_perm_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
// broadcast lane k of each quad (quad_perm:[k,k,k,k]); k is a constant after unrolling, the switch folds away
__device__ __forceinline__ int quad_bcast_i(int v, int k)
{
    switch (k) {
    case 0: return quad_perm_i<0x00>(v);
    case 1: return quad_perm_i<0x55>(v);
    case 2: return quad_perm_i<0xAA>(v);
    default: return quad_perm_i<0xFF>(v);
    }
}
__device__ __forceinline__ float quad_bcast_f(float v, int k) { return __int_as_float(quad_bcast_i(__float_as_int(v), k)); }
#define QUAD_XOR1 0xB1                    // quad_perm:[1,0,3,2]
#define QUAD_XOR2 0x4E                    // quad_perm:[2,3,0,1]

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// vm_term<4, false> with the line taps taken from the LDS copy (same arithmetic, same order).
// Round 5: the plane taps are addressed as wave-uniform base + 32-BIT byte offset (global_load ... v_off, s[base] offset:imm): the 64-bit per-lane address
// arithmetic this replaced (v_mad_i64_i32, v_lshlrev_b64, 2 x v_lshl_add_u64 per plane and sub-step: multi-pass instructions) was a tenth of the kernel's VALU
// issue time.  A density plane of the largest grid the ABI admits (4097^2 texels x 64 B) is 1.07 GB: the offsets fit 32 bits.
#ifndef MARCH_ADDR32
#define MARCH_ADDR32 1
#endif
__device__ __forceinline__ float4 vm_term_lds(const float4 *__restrict__ P, const float4 *Ls, int W, int x0, int y0, int l0,
                                              float wx, float wy, float wl, int sub)
{
    const float ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
    const int Wp = W + 1;
#if MARCH_ADDR32
    const unsigned cell = __umul24((unsigned)y0, (unsigned)Wp) + (unsigned)x0;            // both factors below 2^24 (grid <= 4096)
    unsigned o0 = (cell << 6) + ((unsigned)sub << 4), o1 = o0 + ((unsigned)Wp << 6);
    asm volatile("" : "+v"(o0), "+v"(o1));                                                // opaque: hipcc otherwise widens the sums back into 64-bit arithmetic
    const unsigned char *pb = (const unsigned char *)P;
    const float4 t00 = *(const float4 *)(pb + (size_t)o0), t01 = *(const float4 *)(pb + (size_t)o0 + 64);
    const float4 t10 = *(const float4 *)(pb + (size_t)o1), t11 = *(const float4 *)(pb + (size_t)o1 + 64);
#else
    const float4 *p = P + ((size_t)y0 * Wp + x0) * 4 + sub;
    const float4 t00 = p[0], t01 = p[4], t10 = p[(size_t)Wp * 4], t11 = p[(size_t)Wp * 4 + 4];
#endif
    const float4 *q = Ls + l0 * MARCH_LSTRIDE + sub;
    const float4 l0v = q[0], l1v = q[MARCH_LSTRIDE];
    float4 p4 = f4_mul(ux * uy, t00);
    p4 = f4_fma(wx * uy, t01, p4);
    p4 = f4_fma(ux * wy, t10, p4);
    p4 = f4_fma(wx * wy, t11, p4);
    float4 q4 = f4_mul(ul, l0v);
    q4 = f4_fma(wl, l1v, q4);
    return make_float4(p4.x * q4.x, p4.y * q4.y, p4.z * q4.z, p4.w * q4.w);
}

// Round 5: the same term with everything that depends on the SAMPLE alone — the four bilinear weights, the two line weights, the texel byte offset, the LDS
// offset of the line texel — computed once per chunk by the lane that owns the sample (W4, WL, off, loff) and taken from it here by quad_perm DPP
// (quad_bcast_*(x, k4)): written as one broadcast per use so that hipcc's DPP combine folds it into the consuming v_mul_f32 / v_fmac_f32 / v_add_u32 (VOP2 with a
// DPP source: no v_mov_b32_dpp, no instruction at all).  Per sub-step that removes 6 broadcasts, 3 (1 - w), 12 weight products and the cell arithmetic of three
// planes: 182 -> ~150 instructions.  Same products, same order of the FMAs: bit-identical sigma features.
// MEASURED AND NOT ENABLED (profiles/r05_shade16_ab.txt, blocks r5r / r5s): -93 VALU instructions per chunk (-8 %), frames bit-identical (sha256), and the kernel
// 1.8 % SLOWER in both forms (7.86 - 7.97 against 7.71 - 7.83 ms, interleaved on two boxes) — the march is bound by its 48 wave-level loads per chunk on the L1 path
// (r05_ta_mask_probe.txt), not by its VALU count, and the 18 extra live registers per lane cost more than the instructions saved.  Kept as a build option.
#ifndef MARCH_PRECOMP
#define MARCH_PRECOMP 0
#endif
#ifndef MARCH_DPP_ASM
#define MARCH_DPP_ASM 1       // 1: the weight operand of every interpolation op is read through quad_perm DPP by the op itself (v_mul_f32_dpp / v_fmac_f32_dpp as inline asm:
#endif                        // hipcc's DPP combine folds a broadcast into v_mul_f32 only, and only in src0 position — 15 v_mov_b32_dpp per sub-step stayed); 0: builtins
// r = w[quad lane K] * t   /   acc += w[quad lane K] * t   (VOP2 with a DPP source; the asm is not volatile: a pure function of its operands, free to be scheduled)
template <int K>
__device__ __forceinline__ float mul_q(float w, float t)
{
#if MARCH_DPP_ASM
    float r;
    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(w), "v"(t), "n"(K));
    return r;
#else
    return quad_bcast_f(w, K) * t;
#endif
}
template <int K>
__device__ __forceinline__ float fma_q(float w, float t, float acc)
{
#if MARCH_DPP_ASM
    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(w), "v"(t), "n"(K));
    return acc;
#else
    return __builtin_fmaf(quad_bcast_f(w, K), t, acc);
#endif
}
template <int K>
__device__ __forceinline__ float4 vm_term_pre(const float4 *__restrict__ P, const unsigned char *Lbytes, unsigned Wp64, const float W4[4], const float WL[2], unsigned off,
                                              unsigned loff, unsigned sub16)
{
    const unsigned o0 = quad_bcast_i((int)off, K) + sub16, o1 = o0 + Wp64;
    const unsigned char *pb = (const unsigned char *)P;
    const float4 t00 = *(const float4 *)(pb + (size_t)o0), t01 = *(const float4 *)(pb + (size_t)o0 + 64);
    const float4 t10 = *(const float4 *)(pb + (size_t)o1), t11 = *(const float4 *)(pb + (size_t)o1 + 64);
    const unsigned lo = quad_bcast_i((int)loff, K) + sub16;
    const float4 l0v = *(const float4 *)(Lbytes + lo), l1v = *(const float4 *)(Lbytes + lo + 16 * MARCH_LSTRIDE);
    float4 p4, q4;
    p4.x = mul_q<K>(W4[0], t00.x); p4.y = mul_q<K>(W4[0], t00.y); p4.z = mul_q<K>(W4[0], t00.z); p4.w = mul_q<K>(W4[0], t00.w);
    p4.x = fma_q<K>(W4[1], t01.x, p4.x); p4.y = fma_q<K>(W4[1], t01.y, p4.y); p4.z = fma_q<K>(W4[1], t01.z, p4.z); p4.w = fma_q<K>(W4[1], t01.w, p4.w);
    p4.x = fma_q<K>(W4[2], t10.x, p4.x); p4.y = fma_q<K>(W4[2], t10.y, p4.y); p4.z = fma_q<K>(W4[2], t10.z, p4.z); p4.w = fma_q<K>(W4[2], t10.w, p4.w);
    p4.x = fma_q<K>(W4[3], t11.x, p4.x); p4.y = fma_q<K>(W4[3], t11.y, p4.y); p4.z = fma_q<K>(W4[3], t11.z, p4.z); p4.w = fma_q<K>(W4[3], t11.w, p4.w);
    q4.x = mul_q<K>(WL[0], l0v.x); q4.y = mul_q<K>(WL[0], l0v.y); q4.z = mul_q<K>(WL[0], l0v.z); q4.w = mul_q<K>(WL[0], l0v.w);
    q4.x = fma_q<K>(WL[1], l1v.x, q4.x); q4.y = fma_q<K>(WL[1], l1v.y, q4.y); q4.z = fma_q<K>(WL[1], l1v.z, q4.z); q4.w = fma_q<K>(WL[1], l1v.w, q4.w);
    return make_float4(p4.x * q4.x, p4.y * q4.y, p4.z * q4.z, p4.w * q4.w);
}
// the three planes of one sub-step (K a constant after unrolling)
template <int K>
__device__ __forceinline__ float vm_sum_pre(const SceneDev &sc, const unsigned char *l0, const unsigned char *l1, const unsigned char *l2, const float Wq[3][4], const float Wl[3][2],
                                            const unsigned offp[3], const unsigned offl[3], unsigned sub16)
{
    const float4 a = vm_term_pre<K>(sc.dplane[0], l0, ((unsigned)sc.grid[0] + 1u) << 6, Wq[0], Wl[0], offp[0], offl[0], sub16);
    const float4 b = vm_term_pre<K>(sc.dplane[1], l1, ((unsigned)sc.grid[0] + 1u) << 6, Wq[1], Wl[1], offp[1], offl[1], sub16);
    const float4 cc = vm_term_pre<K>(sc.dplane[2], l2, ((unsigned)sc.grid[1] + 1u) << 6, Wq[2], Wl[2], offp[2], offl[2], sub16);
    return ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((cc.x + cc.y) + (cc.z + cc.w));
}

template <bool DENSE, bool LDSL>
__global__ __launch_bounds__(64 * MARCH_MAX_WAVES) void march_kernel(const SceneDev sc, const float *__restrict__ rays,
                                                                     const int n_rays, const int S, const int s_cap,
                                                                     const MarchSampling sm, const float eps_T,
                                                                     MarchOut mo, const tvr_dense_out dn)
{
    // LDS: [cursor 16 B][lines: 3 x (L+1) x 4 float4, LDSL only][per-wave weight lists f32 s_cap][per-wave sample lists u16 s_cap]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = blockDim.x >> 6;
    unsigned *cursor = (unsigned *)lds_raw;
    const float4 *ls0 = (const float4 *)(lds_raw + MARCH_HDR);
    const int ln0 = LDSL ? (sc.grid[2] + 1) * MARCH_LSTRIDE : 0, ln1 = LDSL ? (sc.grid[1] + 1) * MARCH_LSTRIDE : 0, ln2 = LDSL ? (sc.grid[0] + 1) * MARCH_LSTRIDE : 0;
    const float4 *ls1 = ls0 + ln0, *ls2 = ls1 + ln1;                     // line i runs along axis vecMode[i] = 2 - i
    float *bufw = (float *)(ls2 + ln2) + (size_t)wave * s_cap;
    unsigned short *bufj = (unsigned short *)((float *)(ls2 + ln2) + (size_t)n_waves * s_cap) + (size_t)wave * s_cap;
#ifdef TVR_MARCH_TIMELINE                               // diagnostic build (scripts/march_timeline.py): stats[32 + 8 b ..] = {start, filled, first wave end, last wave end, chunks, rays} of group b, 100 MHz ticks
    unsigned long long tl_chunks = 0ull, tl_rays = 0ull;
    if (mo.stats && threadIdx.x == 0) mo.stats[32 + 8 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x == 0) *cursor = 0u;
    unsigned long long *slots = (unsigned long long *)(lds_raw + 16);
    if (threadIdx.x < MARCH_SLOTS) slots[threadIdx.x] = 0ull;
    unsigned long long *gstat = (unsigned long long *)(lds_raw + 16 + 8 * MARCH_SLOTS);       // per-group sums of the three counters: ONE global atomic each per group
    if (threadIdx.x < 3) gstat[threadIdx.x] = 0ull;                           // (4096 same-address atomics per launch cost a 4096-ray call ~100 us)
    if (LDSL) {
        float4 *dst = (float4 *)(lds_raw + MARCH_HDR);
        for (int i = threadIdx.x; i < (sc.grid[2] + 1) * 4; i += blockDim.x) dst[(i >> 2) * MARCH_LSTRIDE + (i & 3)] = sc.dline[0][i];
        for (int i = threadIdx.x; i < (sc.grid[1] + 1) * 4; i += blockDim.x) dst[ln0 + (i >> 2) * MARCH_LSTRIDE + (i & 3)] = sc.dline[1][i];
        for (int i = threadIdx.x; i < (sc.grid[0] + 1) * 4; i += blockDim.x) dst[ln0 + ln1 + (i >> 2) * MARCH_LSTRIDE + (i & 3)] = sc.dline[2][i];
    }
    __syncthreads();
#ifdef TVR_MARCH_TIMELINE
    if (mo.stats && threadIdx.x == 0) mo.stats[32 + 8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    const int sub = lane & 3;
    // clock probe (stats only): shader-clock and 100 MHz reference ticks over this workgroup's lifetime
    unsigned long long clk0 = 0ull, ref0 = 0ull;
    if (mo.stats && threadIdx.x == 0) { clk0 = __builtin_amdgcn_s_memtime(); ref0 = __builtin_amdgcn_s_memrealtime(); }

    // this group's tiles: XCD x (groups x, x+8, ...: observed round-robin dispatch) owns a contiguous tile range; speed only
    const int n_tiles = (n_rays + MARCH_TILE - 1) / MARCH_TILE;
    (void)n_tiles;
#if !TVR_MARCH_RASTER
    const int nx = gridDim.x < 8u ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % nx, bi = blockIdx.x / nx, nbx = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int t0 = (int)((long long)n_tiles * xcd / nx), t1 = (int)((long long)n_tiles * (xcd + 1) / nx);
#endif

    unsigned long long st_eval = 0, st_bbox = 0, st_term = 0;
    int last_start = 0;                                  // first ray of this wave's previous handout (dynamic queue: picks the tail granularity)

    for (;;) {
        unsigned ci = 0;
        if (lane == 0) ci = atomicAdd(cursor, 1u);
        ci = __builtin_amdgcn_readfirstlane(ci);
#if TVR_MARCH_DYN
        // the wave that draws the first ray of local tile k takes the next global tile and publishes it {k + 1, tile} in slot k & (MARCH_SLOTS - 1); the others
        // wait for the slot's generation to become k + 1.  Why the wait ends: the publisher stores right behind its draw (one global atomic,
        // ~2 us), and the slot is only overwritten by the publisher of local tile k + 64, which needs the group's cursor to advance by 1024 draws
        // and 64 later publishers to have finished their own global atomic first.  The wait is nevertheless BOUNDED and a miss is LOUD:
        // a waiter that finds a later generation in its slot (it was overtaken) or spins MARCH_SPIN_LIMIT times raises mo.counter[2], stops
        // drawing rays, and the composite kernel then writes NaN to every pixel of the call (tvr.h: tvr_scratch_layout.counter).
        // Round 3: the counter counts RAYS, and a handout is 16 rays (a tile: concurrent neighbours share texels in L1) until the launch's last
        // 2 x 16 x groups rays, which go out 4 at a time: the groups then end within one ray of each other instead of two (a group that draws a
        // whole tile just before the counter runs out works 2 x 60 us after everybody else stopped drawing).  Measured on rank 0's share of an 8-way
        // split (81 920 rays, interleaved A/B on one box): 1.03 - 1.06 ms against 1.06 - 1.07 — about 1 %; the rest of that share's loss against 1/8
        // of a frame (0.98 ms) is the ramp at both ends of a launch whose unit of work is a whole ray.  Local tiles keep 16 cursor positions;
        // positions >= the handout's length are no-ops.
        int ray_start, ray_len;
        {
            const unsigned k = ci / MARCH_TILE, slot = k & (MARCH_SLOTS - 1u);
            if ((ci % MARCH_TILE) == 0u) {
                unsigned t = 0, len = MARCH_TILE;
                if (lane == 0) {
                    // how far the launch is: the first ray of THIS wave's previous handout (a register; one ray-time stale, the tail zone is two tiles per
                    // group wide).  Not a fresh look at the counter: an agent-scope load of that line between the atomics of 4096 waves cost the kernel
                    // 30 % (10.1 vs 7.7 ms: every system-coherent read forces the line the queue-length atomics hammer out of L2).
                    // (launches of fewer than 8 tiles per group keep whole tiles: there the 16 concurrent neighbours' shared texels matter more than the tail)
                    if ((long long)n_rays >= 8LL * MARCH_TILE * (long long)gridDim.x && (long long)last_start + 2LL * MARCH_TILE * (long long)gridDim.x >= (long long)n_rays)
                        len = MARCH_TAIL;
                    t = atomicAdd(mo.counter + MARCH_CTR_WORD, len);
                    const unsigned long long genw = (unsigned long long)(k + 1u) | (len == MARCH_TAIL ? 0x80000000ull : 0ull);
#ifdef TVR_FAULT_INJECT_MARCH                          // test build only (tests/test_gpu_faults.py): the publisher of local tile 3 of group 0 skips a generation
                    if (blockIdx.x == 0 && k == 3u) __hip_atomic_store(&slots[slot], ((unsigned long long)(k + 1u + MARCH_SLOTS) << 32) | t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
#endif
                    __hip_atomic_store(&slots[slot], (genw << 32) | t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ray_start = (int)__builtin_amdgcn_readfirstlane(t);
                ray_len = (int)__builtin_amdgcn_readfirstlane(len);
            } else {
                unsigned long long v;
                unsigned spins = 0u, fault = 0u;
                for (;;) {
                    v = __hip_atomic_load(&slots[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned gen = (unsigned)(v >> 32) & 0x7fffffffu;
                    if (gen == k + 1u) break;
                    if (gen > k + 1u) { fault = 1u; break; }                       // overtaken: this tile's number is gone
                    if (++spins > MARCH_SPIN_LIMIT) { fault = 2u; break; }         // the publisher never stored
                    __builtin_amdgcn_s_sleep(1);
                }
                if (fault) {
                    if (lane == 0) atomicOr(mo.counter + 2, fault);
                    break;
                }
                ray_start = (int)(unsigned)v;
                ray_len = (v >> 63) ? MARCH_TAIL : MARCH_TILE;
            }
        }
        if (ray_start >= n_rays) break;
        if ((int)(ci % MARCH_TILE) >= ray_len) continue;
        last_start = ray_start;
        const int ray = ray_start + (int)(ci % MARCH_TILE);
        if (ray >= n_rays) continue;
#elif TVR_MARCH_RASTER
        const int tile = (int)(ci / MARCH_TILE) * (int)gridDim.x + (int)blockIdx.x;     // all groups sweep the image together
        if (tile >= n_tiles) break;
        const int ray = tile * MARCH_TILE + (int)(ci % MARCH_TILE);
        if (ray >= n_rays) continue;
#else
        const int tile = t0 + (int)(ci / MARCH_TILE) * nbx + bi;
        if (tile >= t1) break;
        const int ray = tile * MARCH_TILE + (int)(ci % MARCH_TILE);
        if (ray >= n_rays) continue;
#endif
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = rays[(size_t)ray * 6 + k];
            d[k] = rays[(size_t)ray * 6 + 3 + k];
        }
        const float tmin = ray_tmin(sc, o, d);
        const bool has_jit = sm.jitter != nullptr;
        const float u = has_jit ? sm.jitter[ray] : 0.0f;
        const float *__restrict__ zrow = sm.zv ? sm.zv + (size_t)ray * S : nullptr;     // explicit depths (wave-uniform choice)
        float lam6 = 1.0f;
        int n6 = 0;                                        // samples whose (1 - alpha + 1e-6) factor lam6 already holds

        float T = 1.0f, acc_l = 0.0f, dep_l = 0.0f;
        int napp = 0;
        bool seen = false, terminated = false;
        int c = 0;
        for (; c * 64 < S; ++c) {
            const int j = c * 64 + lane;
            const bool inr = j < S;
            float fj = (float)j, fj1 = (float)(j + 1);
            if (has_jit) { fj = fj + u; fj1 = fj1 + u; }
            float z = tmin + sc.step * fj;                     // tensorBase.py:354-355
            float z1 = tmin + sc.step * fj1;
            if (zrow) {
                z = inr ? zrow[j] : 0.0f;
                z1 = (j < S - 1) ? zrow[j + 1] : z;
            }
            float p[3], n[3], f[3];
            bool bbox = inr;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                p[k] = o[k] + d[k] * z;                        // :357
                bbox = bbox & !((sc.lo[k] > p[k]) | (p[k] > sc.hi[k]));   // :358
            }
            bool valid = bbox;
            if (sc.avol != nullptr) {
                if (bbox) valid = sc.abits ? alpha_positive(sc, p) : (alpha_lookup(sc, p) > 0.0f);  // :491-496
            }
            int i0[3];
            float w[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                n[k] = (p[k] - sc.lo[k]) * sc.inv[k] - 1.0f;   // :223-224
                f[k] = unnorm(n[k], sc.gm1[k]);
                const float fl = floorf(f[k]);
                i0[k] = (int)fl;
                w[k] = f[k] - fl;
            }
            const unsigned long long mb = __ballot(bbox), mv = __ballot(valid);
            if (DENSE) {
                const size_t q = (size_t)ray * S + j;
                if (inr) {
                    if (dn.z) dn.z[q] = z;
                    if (dn.valid) dn.valid[q] = valid;
                    if (dn.bbox_valid) dn.bbox_valid[q] = bbox;
                    if (dn.cell) { dn.cell[q * 3] = i0[0]; dn.cell[q * 3 + 1] = i0[1]; dn.cell[q * 3 + 2] = i0[2]; }
                }
            }
            st_bbox += __popcll(mb);
            if (mv == 0ull) {
                // no density anywhere in this chunk: alpha = 0, T unchanged (1 - 0 + 1e-10 == 1 in fp32), weights 0
                if (DENSE && inr) {
                    const size_t q = (size_t)ray * S + j;
                    if (dn.sigma_feature) dn.sigma_feature[q] = 0.f;
                    if (dn.sigma) dn.sigma[q] = 0.f;
                    if (dn.alpha) dn.alpha[q] = 0.f;
                    if (dn.weight) dn.weight[q] = 0.f;
                }
                if (mb == 0ull && seen && !DENSE) break;        // left the (convex) box: nothing further can be valid
                continue;
            }
            seen = true;
            st_eval += __popcll(mv);
#ifdef TVR_MARCH_TIMELINE
            tl_chunks++;
#endif

            // ---- density feature: 4 sub-steps, quad-per-sample gather ----
            float sf = 0.0f;
#if MARCH_PRECOMP
            // what the sub-steps need of this lane's sample, once per chunk (vm_term_pre): plane p's weights in vm_term_lds's order for its (a, b | line) axes —
            // plane 0 (x, y | z), plane 1 (x, z | y), plane 2 (y, z | x) — and the byte offsets of its first texel and of its line texel in LDS
            float Wq[3][4], Wl[3][2];
            unsigned offp[3], offl[3];
            if (LDSL) {
                const float ux = 1.0f - w[0], uy = 1.0f - w[1], uz = 1.0f - w[2];
                Wq[0][0] = ux * uy; Wq[0][1] = w[0] * uy; Wq[0][2] = ux * w[1]; Wq[0][3] = w[0] * w[1];
                Wq[1][0] = ux * uz; Wq[1][1] = w[0] * uz; Wq[1][2] = ux * w[2]; Wq[1][3] = w[0] * w[2];
                Wq[2][0] = uy * uz; Wq[2][1] = w[1] * uz; Wq[2][2] = uy * w[2]; Wq[2][3] = w[1] * w[2];
                Wl[0][0] = uz; Wl[0][1] = w[2]; Wl[1][0] = uy; Wl[1][1] = w[1]; Wl[2][0] = ux; Wl[2][1] = w[0];
                const unsigned Wp0 = (unsigned)sc.grid[0] + 1u, Wp2 = (unsigned)sc.grid[1] + 1u;
                offp[0] = (__umul24((unsigned)i0[1], Wp0) + (unsigned)i0[0]) << 6;
                offp[1] = (__umul24((unsigned)i0[2], Wp0) + (unsigned)i0[0]) << 6;
                offp[2] = (__umul24((unsigned)i0[2], Wp2) + (unsigned)i0[1]) << 6;
                offl[0] = (unsigned)i0[2] * (16u * MARCH_LSTRIDE); offl[1] = (unsigned)i0[1] * (16u * MARCH_LSTRIDE); offl[2] = (unsigned)i0[0] * (16u * MARCH_LSTRIDE);
                if (!valid) { offp[0] = offp[1] = offp[2] = 0u; offl[0] = offl[1] = offl[2] = 0u; }       // (samples outside the box: their lanes' offsets are never used, keep them harmless)
            }
            const unsigned sub16 = (unsigned)sub << 4;
#endif
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const bool v = quad_bcast_i((int)valid, k4) != 0;
                if (__ballot(v) == 0ull) continue;
#if MARCH_PRECOMP
                if (LDSL) {
                    float part = 0.0f;
                    if (v) {
                        const unsigned char *l0b = (const unsigned char *)ls0, *l1b = (const unsigned char *)ls1, *l2b = (const unsigned char *)ls2;
                        part = k4 == 0 ? vm_sum_pre<0>(sc, l0b, l1b, l2b, Wq, Wl, offp, offl, sub16) : (k4 == 1 ? vm_sum_pre<1>(sc, l0b, l1b, l2b, Wq, Wl, offp, offl, sub16)
                             : (k4 == 2 ? vm_sum_pre<2>(sc, l0b, l1b, l2b, Wq, Wl, offp, offl, sub16) : vm_sum_pre<3>(sc, l0b, l1b, l2b, Wq, Wl, offp, offl, sub16)));
                    }
                    part += quad_perm_f<QUAD_XOR1>(part);
                    part += quad_perm_f<QUAD_XOR2>(part);
                    if (sub == k4) sf = part;
                    continue;
                }
#endif
                const int ix = quad_bcast_i(i0[0], k4), iy = quad_bcast_i(i0[1], k4), iz = quad_bcast_i(i0[2], k4);
                const float wx = quad_bcast_f(w[0], k4), wy = quad_bcast_f(w[1], k4), wz = quad_bcast_f(w[2], k4);
                float part = 0.0f;
                if (v && LDSL) {
                    const float4 a = vm_term_lds(sc.dplane[0], ls0, sc.grid[0], ix, iy, iz, wx, wy, wz, sub);
                    const float4 b = vm_term_lds(sc.dplane[1], ls1, sc.grid[0], ix, iz, iy, wx, wz, wy, sub);
                    const float4 cc = vm_term_lds(sc.dplane[2], ls2, sc.grid[1], iy, iz, ix, wy, wz, wx, sub);
                    part = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((cc.x + cc.y) + (cc.z + cc.w));
                } else if (v) {
                    // plane0 (x,y)·line0(z) ; plane1 (x,z)·line1(y) ; plane2 (y,z)·line2(x)   (matMode / vecMode)
                    const float4 a = vm_term<4, false>(sc.dplane[0], sc.dline[0], sc.grid[0], sc.grid[1], sc.grid[2], ix, iy, iz, wx, wy, wz, sub);
                    const float4 b = vm_term<4, false>(sc.dplane[1], sc.dline[1], sc.grid[0], sc.grid[2], sc.grid[1], ix, iz, iy, wx, wz, wy, sub);
                    const float4 cc = vm_term<4, false>(sc.dplane[2], sc.dline[2], sc.grid[1], sc.grid[2], sc.grid[0], iy, iz, ix, wy, wz, wx, sub);
                    part = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((cc.x + cc.y) + (cc.z + cc.w));
                }
                part += quad_perm_f<QUAD_XOR1>(part);
                part += quad_perm_f<QUAD_XOR2>(part);
                if (sub == k4) sf = part;
            }

            float sigma = 0.0f;
            if (valid) sigma = (sc.act == 0) ? softplus_f(sf + sc.shift) : fmaxf(sf, 0.0f);   // :444-448
            float dist = (j < S - 1) ? (z1 - z) : 0.0f;           // :488
            dist = dist * sc.scale;                               // :511
            const float alpha = 1.0f - expf(-sigma * dist);       // :19
            const float fT = (1.0f - alpha) + 1e-10f;             // :21
            if (mo.lam6) {                                        // nerfplusplus.py:277: cumprod(1 - alpha + TINY_NUMBER), TINY_NUMBER = 1e-6
                float f6 = inr ? (1.0f - alpha) + 1e-6f : 1.0f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) f6 = f6 * __shfl_xor(f6, off);
                lam6 = lam6 * f6;
                n6 += min(64, S - c * 64);
            }
            float incl = fT;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_up(incl, off);
                if (lane >= off) incl = incl * t;
            }
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.0f;
            const float Tj = T * excl;
            const float wgt = alpha * Tj;                          // :23
            acc_l += wgt;
            dep_l += wgt * z;
            const bool app = wgt > sc.thres;                       // :513
            const unsigned long long ma = __ballot(app);
            if (ma) {
                const int pos = napp + __popcll(ma & ((1ull << lane) - 1ull));
                if (app) { bufw[pos] = wgt; bufj[pos] = (unsigned short)j; }
                napp += __popcll(ma);
            }
            T = T * __shfl(incl, 63);
            if (DENSE && inr) {
                const size_t q = (size_t)ray * S + j;
                if (dn.sigma_feature) dn.sigma_feature[q] = valid ? sf : 0.f;
                if (dn.sigma) dn.sigma[q] = sigma;
                if (dn.alpha) dn.alpha[q] = alpha;
                if (dn.weight) dn.weight[q] = wgt;
            }
            if (T < eps_T) { terminated = true; ++c; break; }
        }
        if (DENSE) {
            // samples never visited (early exit) get zeros so the dense arrays are fully defined
            for (int cc = c; cc * 64 < S; ++cc) {
                const int j = cc * 64 + lane;
                if (j < S) {
                    const size_t q = (size_t)ray * S + j;
                    const float fjj = has_jit ? ((float)j + u) : (float)j;
                    if (dn.z) dn.z[q] = zrow ? zrow[j] : tmin + sc.step * fjj;
                    if (dn.valid) dn.valid[q] = 0;
                    if (dn.bbox_valid) dn.bbox_valid[q] = 0;
                    if (dn.cell) { dn.cell[q * 3] = 0; dn.cell[q * 3 + 1] = 0; dn.cell[q * 3 + 2] = 0; }
                    if (dn.sigma_feature) dn.sigma_feature[q] = 0.f;
                    if (dn.sigma) dn.sigma[q] = 0.f;
                    if (dn.alpha) dn.alpha[q] = 0.f;
                    if (dn.weight) dn.weight[q] = 0.f;
                }
            }
        }
        st_term += terminated ? 1 : 0;
#ifdef TVR_MARCH_TIMELINE
        tl_rays++;
#endif

        const float acc = wave_sum(acc_l);
        const float dep = wave_sum(dep_l);
        unsigned base = 0;
        if (lane == 0 && napp > 0) base = atomicAdd(mo.counter, (unsigned)napp);
        base = __shfl(base, 0);
        if (lane == 0) {
            mo.ray_off[ray] = base;
            mo.ray_cnt[ray] = (unsigned)napp;
            mo.acc[ray] = acc;
            mo.depth[ray] = dep + (1.0f - acc) * d[2];            // :531 (rays[..., -1] is d_z)
            // samples of skipped chunks / behind an early exit have alpha = 0: each factor is fp32(1 + 1e-6) = 1 + 8 ulp, log = 9.5367386e-7
            if (mo.lam6) mo.lam6[ray] = lam6 * expf((float)(S - n6) * 9.5367386e-7f);
            if (DENSE) {
                if (dn.bg_weight) dn.bg_weight[ray] = T;
                if (dn.acc) dn.acc[ray] = acc;
                if (dn.t_min) dn.t_min[ray] = tmin;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = lane; i < napp; i += 64) {
            const unsigned ej = bufj[i];
            float fj = (float)ej;
            if (has_jit) fj = fj + u;
            const float z = zrow ? zrow[ej] : tmin + sc.step * fj;
            float4 qv;
            qv.x = ((o[0] + d[0] * z) - sc.lo[0]) * sc.inv[0] - 1.0f;
            qv.y = ((o[1] + d[1] * z) - sc.lo[1]) * sc.inv[1] - 1.0f;
            qv.z = ((o[2] + d[2] * z) - sc.lo[2]) * sc.inv[2] - 1.0f;
            qv.w = bufw[i];
            mo.q_pos[base + i] = qv;
            mo.q_ray[base + i] = (unsigned)ray;
            if (mo.q_j) mo.q_j[base + i] = ej;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#ifdef TVR_MARCH_TIMELINE
    if (mo.stats && lane == 0) {
        const unsigned long long te = __builtin_amdgcn_s_memrealtime();
        atomicMin((unsigned long long *)&mo.stats[32 + 8 * blockIdx.x + 2], te);
        atomicMax((unsigned long long *)&mo.stats[32 + 8 * blockIdx.x + 3], te);
        atomicAdd((unsigned long long *)&mo.stats[32 + 8 * blockIdx.x + 4], tl_chunks);
        atomicAdd((unsigned long long *)&mo.stats[32 + 8 * blockIdx.x + 5], tl_rays);
    }
#endif
    if (mo.stats) {                                      // (wave-uniform: every wave of the group reaches this barrier exactly once)
        if (lane == 0) {
            atomicAdd(&gstat[0], st_eval);
            atomicAdd(&gstat[1], st_bbox);
            atomicAdd(&gstat[2], st_term);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd((unsigned long long *)&mo.stats[TVR_STAT_SAMPLES_EVAL], gstat[0]);
            atomicAdd((unsigned long long *)&mo.stats[TVR_STAT_SAMPLES_BBOX], gstat[1]);
            atomicAdd((unsigned long long *)&mo.stats[TVR_STAT_RAYS_TERMINATED], gstat[2]);
            atomicAdd((unsigned long long *)&mo.stats[TVR_STAT_MARCH_CLK], __builtin_amdgcn_s_memtime() - clk0);
            atomicAdd((unsigned long long *)&mo.stats[TVR_STAT_MARCH_REF], __builtin_amdgcn_s_memrealtime() - ref0);
        }
    }
}

// composite tail (tensorBase.py:520-527): rgb_map = sum_j w_j*rgb_j (+ 1-acc if white_bg), clamp(0,1).
// Eight lanes per ray read the ray's contiguous, sample-ordered queue segment 128 B at a time (lane l sums entries l, l+8, ...),
// then a fixed butterfly adds the eight partial sums: the summation order depends on nothing but the ray -> deterministic.
#define COMP_LANES 8
__global__ __launch_bounds__(256) void composite_kernel(const MarchOut mo, const int n_rays, const int white_bg,
                                                        float *__restrict__ rgb_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = t / COMP_LANES, l = t % COMP_LANES;
    const bool live = r < n_rays;
    const unsigned base = live ? mo.ray_off[r] : 0u, cnt = live ? mo.ray_cnt[r] : 0u;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    // four entries in flight per lane (a 4096-ray call is one wave per 8 rays and ~11 dependent round trips per lane otherwise: 10.9 us per
    // call, rocprofv3, where the data would take 3); the additions keep their order
    unsigned i = l;
    for (; i + 3 * COMP_LANES < cnt; i += 4 * COMP_LANES) {
        const float4 e0 = mo.q_out[base + i], e1 = mo.q_out[base + i + COMP_LANES], e2 = mo.q_out[base + i + 2 * COMP_LANES],
                     e3 = mo.q_out[base + i + 3 * COMP_LANES];     // {r,g,b,w} written by the shade kernel
        c0 = c0 + e0.w * e0.x; c1 = c1 + e0.w * e0.y; c2 = c2 + e0.w * e0.z;
        c0 = c0 + e1.w * e1.x; c1 = c1 + e1.w * e1.y; c2 = c2 + e1.w * e1.z;
        c0 = c0 + e2.w * e2.x; c1 = c1 + e2.w * e2.y; c2 = c2 + e2.w * e2.z;
        c0 = c0 + e3.w * e3.x; c1 = c1 + e3.w * e3.y; c2 = c2 + e3.w * e3.z;
    }
    for (; i < cnt; i += COMP_LANES) {
        const float4 e = mo.q_out[base + i];
        c0 = c0 + e.w * e.x;
        c1 = c1 + e.w * e.y;
        c2 = c2 + e.w * e.z;
    }
#pragma unroll
    for (int off = 1; off < COMP_LANES; off <<= 1) {
        c0 = c0 + __shfl_xor(c0, off);
        c1 = c1 + __shfl_xor(c1, off);
        c2 = c2 + __shfl_xor(c2, off);
    }
    if (!live || l != 0) return;
    if (mo.counter[2] != 0u) {                        // the march raised its fault flag (tile-queue wait): no pixel of this call is trustworthy
        const float qnan = __int_as_float(0x7fc00000);
        rgb_out[(size_t)r * 3 + 0] = qnan; rgb_out[(size_t)r * 3 + 1] = qnan; rgb_out[(size_t)r * 3 + 2] = qnan;
        mo.depth[r] = qnan;
        return;
    }
    const float acc = mo.acc[r];
    if (white_bg) {
        const float bg = 1.0f - acc;
        c0 = c0 + bg; c1 = c1 + bg; c2 = c2 + bg;
    }
    rgb_out[(size_t)r * 3 + 0] = clamp01(c0);
    rgb_out[(size_t)r * 3 + 1] = clamp01(c1);
    rgb_out[(size_t)r * 3 + 2] = clamp01(c2);
}

// additional_output: scatter per-entry rgb back to the dense [n,S,3] array
__global__ __launch_bounds__(256) void scatter_rgb_kernel(const MarchOut mo, const int S, float *__restrict__ rgb_dense)
{
    const unsigned n = *mo.counter;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const float4 v = mo.q_out[e];
        const size_t q = ((size_t)mo.q_ray[e] * S + mo.q_j[e]) * 3;
        rgb_dense[q] = v.x; rgb_dense[q + 1] = v.y; rgb_dense[q + 2] = v.z;
    }
}

// ---- generic feature lookups (compute_densityfeature / sample_alpha API): arbitrary coordinates, zeros padding ----
__global__ __launch_bounds__(256) void density_feature_kernel(const SceneDev sc, const float *__restrict__ xyz, const long long m,
                                                              float *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = t >> 2;
    const int sub = threadIdx.x & 3;
    float part = 0.0f;
    if (s < m) {
        int i0[3];
        float w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float f = unnorm(xyz[s * 3 + k], sc.gm1[k]);
            const float fl = floorf(fminf(fmaxf(f, -2.0f), sc.gm1[k] + 2.0f));
            i0[k] = (int)fl;
            w[k] = f - fl;
        }
        const float4 a = vm_term<4, true>(sc.dplane[0], sc.dline[0], sc.grid[0], sc.grid[1], sc.grid[2], i0[0], i0[1], i0[2], w[0], w[1], w[2], sub);
        const float4 b = vm_term<4, true>(sc.dplane[1], sc.dline[1], sc.grid[0], sc.grid[2], sc.grid[1], i0[0], i0[2], i0[1], w[0], w[2], w[1], sub);
        const float4 c = vm_term<4, true>(sc.dplane[2], sc.dline[2], sc.grid[1], sc.grid[2], sc.grid[0], i0[1], i0[2], i0[0], w[1], w[2], w[0], sub);
        part = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w));
    }
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    if (s < m && sub == 0) out[s] = part;
}

__global__ __launch_bounds__(256) void alpha_sample_kernel(const SceneDev sc, const float *__restrict__ xyz, const long long m,
                                                           float *__restrict__ out)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    const float p[3] = {xyz[s * 3], xyz[s * 3 + 1], xyz[s * 3 + 2]};
    out[s] = alpha_lookup(sc, p);
}

// alpha volume -> bit volume (one thread per 32-voxel word)
__global__ __launch_bounds__(256) void alpha_bits_kernel(const float *__restrict__ vol, const long long n, unsigned *__restrict__ bits)
{
    const long long wd = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wd * 32 >= n) return;
    unsigned v = 0u;
    for (int b = 0; b < 32; ++b) {
        const long long i = wd * 32 + b;
        if (i < n && vol[i] > 0.0f) v |= 1u << b;
    }
    bits[wd] = v;
}

hipError_t launch_alpha_bits(const float *vol, long long n, unsigned *bits, hipStream_t stream)
{
    const long long words = (n + 31) / 32;
    hipLaunchKernelGGL(alpha_bits_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, vol, n, bits);
    return hipGetLastError();
}

// ---- host launchers ----
template <bool DENSE, bool LDSL>
static hipError_t launch_march_t(const SceneDev &sc, const float *rays, int n_rays, int S, const MarchSampling &sm, float eps_T, const MarchOut &mo,
                                 const tvr_dense_out &dn, int waves, size_t lds, unsigned grid, hipStream_t stream)
{
    hipError_t rc = hipFuncSetAttribute((const void *)march_kernel<DENSE, LDSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL((march_kernel<DENSE, LDSL>), dim3(grid), dim3(64 * waves), lds, stream, sc, rays, n_rays, S, S, sm, eps_T, mo, dn);
    return hipGetLastError();
}

static int device_cu_count()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    return cus;
}

hipError_t launch_march(const SceneDev &sc, const float *rays, int n_rays, int S, const MarchSampling &sm, float eps_T,
                        const MarchOut &mo, const tvr_dense_out *dense, hipStream_t stream)
{
    // LDS budget: the density lines (if they fit next to at least 4 waves' lists) + 6 B per sample and wave for the appearance lists
    const size_t kLds = 160 * 1024, line_bytes = ((size_t)sc.grid[0] + sc.grid[1] + sc.grid[2] + 3) * 16 * MARCH_LSTRIDE, per_wave = (size_t)S * 6;
    bool ldsl = MARCH_HDR + line_bytes + 4 * per_wave + 64 <= kLds;
    const size_t fixed = MARCH_HDR + (ldsl ? line_bytes : 0) + 64;
    int waves = (int)((kLds - fixed) / per_wave);
    waves = waves >= 16 ? 16 : (waves >= 12 ? 12 : (waves >= 8 ? 8 : (waves >= 4 ? 4 : (waves >= 2 ? 2 : 1))));
    const size_t lds = fixed + (size_t)waves * per_wave;
    const int n_tiles = (n_rays + MARCH_TILE - 1) / MARCH_TILE;
    // one group per CU when it holds 16 waves; proportionally more groups when the lists force smaller ones
    long long grid = (long long)device_cu_count() * (16 / waves > 0 ? 16 / waves : 1);
#ifdef TVR_EXP_GRID                                             // scripts/overlap_experiment.py only: a build_variant.sh -DTVR_EXP_GRID library
    if (const char *g = getenv("TVR_EXP_GRID_MARCH")) { const long long v = atoll(g); if (v > 0 && v < grid) grid = v; }
#endif
    if (grid > n_tiles) grid = n_tiles;
    if (grid < 1) grid = 1;
    tvr_dense_out none = {};
    const tvr_dense_out &dn = dense ? *dense : none;
    if (dense) return ldsl ? launch_march_t<true, true>(sc, rays, n_rays, S, sm, eps_T, mo, dn, waves, lds, (unsigned)grid, stream)
                           : launch_march_t<true, false>(sc, rays, n_rays, S, sm, eps_T, mo, dn, waves, lds, (unsigned)grid, stream);
    return ldsl ? launch_march_t<false, true>(sc, rays, n_rays, S, sm, eps_T, mo, dn, waves, lds, (unsigned)grid, stream)
                : launch_march_t<false, false>(sc, rays, n_rays, S, sm, eps_T, mo, dn, waves, lds, (unsigned)grid, stream);
}

hipError_t launch_composite(const MarchOut &mo, int n_rays, int white_bg, float *rgb, hipStream_t stream)
{
    const long long threads = (long long)n_rays * COMP_LANES;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, mo, n_rays, white_bg, rgb);
    return hipGetLastError();
}

hipError_t launch_scatter_rgb(const MarchOut &mo, int S, float *rgb_dense, hipStream_t stream)
{
    hipLaunchKernelGGL(scatter_rgb_kernel, dim3(1024), dim3(256), 0, stream, mo, S, rgb_dense);
    return hipGetLastError();
}

hipError_t launch_density_feature(const SceneDev &sc, const float *xyz, long long m, float *out, hipStream_t stream)
{
    const long long blocks = (m * 4 + 255) / 256;
    hipLaunchKernelGGL(density_feature_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, sc, xyz, m, out);
    return hipGetLastError();
}

hipError_t launch_alpha_sample(const SceneDev &sc, const float *xyz, long long m, float *out, hipStream_t stream)
{
    hipLaunchKernelGGL(alpha_sample_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, sc, xyz, m, out);
    return hipGetLastError();
}
