// tvr_shade16.hip — the render path's shade kernel on v_mfma_f32_16x16x32_f16 (round 5).
//
// The same work as tvr_shade.hip's shade_kernel<SH_SRC_QUEUE, SH_DST_QUEUE> for TensorVMSplit scenes in the default (fp32-class) arithmetic:
//   models/tensoRF.py:228-244 compute_appfeature, models/tensorBase.py:9-15 positional_encoding, :76-86 MLPRender_Fea.execute (paths relative to
//   /root/reference/tensorf-myc/) — appearance gather, basis product, positional encoding, the 150 -> 128 -> 128 -> 3 MLP, sigmoid; three fp16 hi/lo products per
//   fp32 product, fp32 accumulation.
// Why a second kernel: the 32x32x16 kernel runs against the chip's POWER limit (1.67 GHz on 256 CUs), and at that limit the 16x16x32 shape delivers 1.10 x the
// FLOP/s with the same LDS fragment traffic (profiles/r05_mfma_shape_probe.txt: 1641 vs 1491 TFLOP/s, cycles per MFMA 17.25 vs 32.67); a timing build of the old
// kernel with every MFMA replaced by two of this shape ran 11.28 ms against 12.42 (profiles/r05_shade_mfma_shape_diag.txt).  This is that kernel built for real.
//
// Tile: 32 queue entries per wave as TWO 16-column B tiles that share every A fragment.  Lane (c = lane & 15, g = lane >> 4) serves entries eA = 32 tile + c
// and eB = eA + 16 and holds, of each, k-group g (8 of the 32 k values of a k-step) as B operand and rows 4g .. 4g+3 of every 16-row block as accumulators.
//   GATHER phase  every lane gathers for ITS OWN entry (groups 0, 1: A; groups 2, 3: B): nine units of 6 taps x 2 float4 (the 32x32 kernel's count), unit u = channels
//                 16u + 8 (g & 1) .. + 7 of the 144 = of plane u / 3; offsets and weights once per plane.  A unit's fragment holds {A | B} in its lower | upper half, and
//                 v_permlane32_swap(unit 2s, unit 2s+1) IS k-step s's B operand of tile A | tile B in the natural k order; unit 8's halves are k-step 4's (the A operand of
//                 groups 2, 3 is zero there).  The interpolation is tvr_shade_common.h's taps_eval op for op: the same h values as the 32x32 kernel's.
//   MATRIX phase  (behind the SIMD's matrix token, as before)  basis 60 MFMAs -> F (8 base values per lane and entry) -> layer 1: 5 k-steps x 8 row blocks x 6
//                 MFMAs, the next k-step's [v, sin v, sin 2v, cos v, cos 2v] fragments derived under them -> layer 2: B = relu(layer-1 accumulators) as they lie
//                 (row blocks 2s, 2s+1 of lane group g = k-step s), row block by row block, layer 3 (fp32 FMAs) of row block rb-1 under the MFMAs of row block rb.
//   finish        last row block's layer 3, the sums over the four lane groups (v_permlane32_swap / v_permlane16_swap adds, fixed order), sigmoid, store.
// Per accumulator the order of the additions differs from the 32x32 kernel's (32 k per step instead of 16): results agree to fp32 rounding, not bit for bit; per ray the
// compositing order is unchanged, so every invariance the tests hold (chunking, permutation, partition) holds as before.
// Phase rule: lifted for this kernel on the round-5 evidence (profiles/r05_phase_rule_test.txt: a build with 12 global loads per lane between layer 1 and layer 2 is
// bit-identical on 8 frames x 2 and passes the reproducibility tests) — the loads are still issued in the gather phase here, nothing relies on the lift.
#include <cstdlib>
#include "tvr_device.h"
#include "tvr_kernels.h"
#include "tvr_mfma.h"
#include "tvr_shade_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef S16_WAVES
#define S16_WAVES 8         // 4 (-DS16_WAVES=4): one wave per SIMD, a timing experiment (round 5: how much of the kernel's speed is the second wave's overlap)
#endif
#define S16_THREADS (64 * S16_WAVES)
#define S16_TILE 32
#ifndef S16_PRIO_M
#define S16_PRIO_M 2      // s_setprio in the matrix phase / the gather phase / layer 3's tail (the 32x32 kernel's measured choice)
#endif
#ifndef S16_PRIO_G
#define S16_PRIO_G 0
#endif
#ifndef S16_PD
#define S16_PD 2          // A-fragment pairs are read this many row blocks ahead, ring of S16_PD + 2
#endif
#define S16_RN (S16_PD + 2)
#ifndef S16_MSLEEP
#define S16_MSLEEP 2
#endif
#ifndef S16_TOKEN_SPINS
#define S16_TOKEN_SPINS 4096
#endif
#ifndef S16_MTOKEN
#define S16_MTOKEN 1      // the per-SIMD matrix token (tvr_shade.hip: without it the two waves of a SIMD convoy); 0: A/B builds
#endif
#ifndef S16_TIMING
#define S16_TIMING 0      // diagnostic build: per-phase s_memtime sums into stats[8..15] (scripts/phase_timing.py)
#endif
#if S16_TIMING
#define S16_STAMP(x) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define S16_STAMP(x)
#endif
#define S16_SB __builtin_amdgcn_sched_barrier(0)
#define S16_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define S16_SG_VALU(n) __builtin_amdgcn_sched_group_barrier(0x402, (n), 0)      // VALU | TRANS
#define S16_SG_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#ifndef S16_FAST_SIGMOID
#define S16_FAST_SIGMOID 1    // v_exp_f32 / v_rcp_f32 instead of expf + an IEEE division: -63 instructions per tile, 12.16 -> 11.97 ms (profiles/r05_shade16_ab.txt)
#endif
#ifndef S16_FAST_SINCOS
#define S16_FAST_SINCOS 1     // sin / cos of the encoding from fract(v / 2 pi) (two VALU instructions) instead of the two-term Cody-Waite reduction (six)
#endif
// The kernel is bound by its SIMD's vector-issue port (492 MFMAs x 8 + ~1800 VALU x 4 cycles per tile against ~12 k cycles per tile and SIMD): every instruction
// removed is time.  v_sin_f32 / v_cos_f32 take revolutions; v * (1 / 2 pi) carries fp32's relative rounding, i.e. a phase error of |v| * 6e-8 rad — 2e-6 at the
// |v| <= 30 of a trained scene, 7e-5 at the |v| ~ 1100 that tests/test_gpu_parity.py::test_large_feature_magnitudes holds to the 1e-3 bar (its docstring prices exactly
// this error); fract keeps the argument inside the instructions' domain for any magnitude.  tvr_shade.hip's other modes keep sincos_pe's Cody-Waite form.
__device__ __forceinline__ void sincos_pe16(float x, float &s, float &c)
{
#if S16_FAST_SINCOS
    const float t = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
#else
    sincos_pe(x, s, c);
#endif
}
#ifndef S16_PIN
#define S16_PIN 1         // 1: one empty-asm pin per interpolated float4 (keeps hipcc from spreading a tap set's consumers over the phase); 0: none
#endif
#ifndef S16_TICKET
#define S16_TICKET 4      // tiles per ticket of the dynamic tile hand-out (0: the static stride of rounds 1 - 4)
#endif
#ifndef S16_TICKET_MIN
#define S16_TICKET_MIN 256  // tiles per wave from which the tickets are used
#endif
#ifndef S16_GDEPTH
#define S16_GDEPTH 1      // units of the gather in flight ahead of the one being interpolated (24 registers each)
#endif
#ifndef S16_DIAG_LDS
#define S16_DIAG_LDS 0    // timing stand-ins (WRONG pictures): 1 = layers 1 / 2 read every other weight fragment from LDS (the fragment bytes of a 64-column form), 2 = none
#endif
#ifndef S16_SPLITB
#define S16_SPLITB 1      // 1: tvr_mfma.h's split8b (the four pairs' operations batched: no s_nop between a pair's residuals and their packed conversion); 0: split8 (A/B)
#endif
#if S16_SPLITB
#define S16_SPLIT8 split8b
#else
#define S16_SPLIT8 split8
#endif
#ifndef S16_SCHED
#define S16_SCHED 1       // 1: sched_group_barrier windows in the matrix phase; 0: hipcc's own order (A/B)
#endif

struct AF16 { uint4 h, l; };
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
// one {hi, lo} A pair x two B tiles x three products (l*hi, h*lo, h*hi per accumulator: the order mfma3 of the 32x32 kernel has)
__device__ __forceinline__ void mfma6(const AF16 &A, const Frag &bA, const Frag &bB, f32x4 &accA, f32x4 &accB)
{
    accA = MFMA16(A.l, bA.hi, accA);
    accB = MFMA16(A.l, bB.hi, accB);
    accA = MFMA16(A.h, bA.lo, accA);
    accB = MFMA16(A.h, bB.lo, accB);
    accA = MFMA16(A.h, bA.hi, accA);
    accB = MFMA16(A.h, bB.hi, accB);
}
// an LDS pointer held in ONE register the compiler cannot see through (base + 16-bit immediates instead of one address register per read)
#define S16_LDS_BASE(name, expr)                                                                                    \
    unsigned name##_a = (unsigned)(size_t)(expr);                                                                   \
    asm volatile("" : "+v"(name##_a));                                                                              \
    const unsigned char *name = (const unsigned char *)(const void __attribute__((address_space(3))) *)(size_t)name##_a

// lower / upper 32 lanes of v in every lane (v_permlane32_swap: vdst' = {vdst.lo, src.lo}, src' = {vdst.hi, src.hi}; profiles/r05_permlane_semantics.txt)
__device__ __forceinline__ void halves_u(unsigned v, unsigned &lo, unsigned &hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lo = r[0]; hi = r[1];
}
__device__ __forceinline__ void halves_f(float v, float &lo, float &hi)
{
    unsigned a, b;
    halves_u(__float_as_uint(v), a, b);
    lo = __uint_as_float(a); hi = __uint_as_float(b);
}
// sum over the four lane groups of two per-entry partial sums: a (entry A) and b (entry B) -> lanes of groups 0, 1 hold A's total, groups 2, 3 B's.
// Fixed order: (g + (g ^ 2)) first, then the two 16-lane rows of a half.
__device__ __forceinline__ float group_sum2(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);       // r0 = {a.lo, b.lo}, r1 = {a.hi, b.hi}
    const float z = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);       // q0 = {z0, z0, z2, z2}, q1 = {z1, z1, z3, z3}
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float group_max2(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float z = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// value of lane group 0's 16-lane row in every lane group (two swaps: 16-lane rows inside a half, then the halves)
__device__ __forceinline__ float bcast_g0(float v)
{
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);      // q0 = {v0, v0, v2, v2}
    unsigned lo, hi;
    halves_u(q[0], lo, hi);
    return __uint_as_float(lo);
}
// value of lane group 1's row in lane groups 0 and 1 (and group 3's in 2 and 3)
__device__ __forceinline__ float from_g1(float v)
{
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);      // q1 = {v1, v1, v3, v3}
    return __uint_as_float(q[1]);
}

// REF (round 6): REFTensoRF's render path (models/REFTensoRF.py:107-133, 174-256; what configs/Scar.txt:28 runs) on the same tiles.  The four heads on h (normal 3, specular
// tint 1, diffuse 3, rho 1) are a THIRD 16-row block of the basis product — 30 more MFMAs, A fragments from global memory (10 KB, L1 / L2 resident; the LDS is full),
// requested behind the phase boundary and used after the two feature blocks; the normalised normal gives d.n and the reflection direction, which take the places of the
// view direction (base rows 27..29) and of the unused row 30 (-d.n: MLPRender_Fea_Ref's input 0) in layer 1's B operands; the colour is relu(tint) * rgb_s + rgb_d.
template <bool RC, bool REF>
__global__ __launch_bounds__(S16_THREADS, S16_WAVES == 4 ? 1 : 2) void shade16_kernel(const SceneDev sc, const ShadeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const bool up = g >= 2;                                       // upper half of the wave: loads entry B's queue record, packs entry B's k-step 4
    {
        const uint4 *src = (const uint4 *)sc.img16;
        for (int i = tid; i < TVR16_IMAGE_BYTES / 16; i += S16_THREADS) ((uint4 *)smem)[i] = src[i];
        if (tid < 4) ((int *)(smem + TVR16_IMAGE_BYTES))[tid] = 0;
        __syncthreads();
    }
#ifdef TVR_DEBUG_SIMD
    // debug build (tests/test_gpu_faults.py): the token pairs waves w and w + 4, assuming they sit on one SIMD (tvr_shade.hip's check, same slots)
    {
        __shared__ int simd_of[S16_WAVES];
        const int hw = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
        if (lane == 0) simd_of[wave] = hw;
        __syncthreads();
        if (a.stats && tid < S16_WAVES / 2 && simd_of[tid] != simd_of[tid + S16_WAVES / 2]) atomicAdd((unsigned long long *)&a.stats[15], 1ull);
        if (a.stats && blockIdx.x == 0 && tid < S16_WAVES) a.stats[16 + tid] = (unsigned long long)simd_of[tid];
        __syncthreads();
    }
#endif
    bool have_tok = false;
    int *mtok = (int *)(smem + TVR16_IMAGE_BYTES) + (wave & 3);
    const long long n_total = (long long)(*a.counter);
    const long long n_tiles = (n_total + S16_TILE - 1) / S16_TILE;
    unsigned long long clk0 = 0ull, ref0 = 0ull;
    if (a.stats && tid == 0) { clk0 = __builtin_amdgcn_s_memtime(); ref0 = __builtin_amdgcn_s_memrealtime(); }
#if S16_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const long long tile_stride = (long long)gridDim.x * S16_WAVES;
    const unsigned lblk = xcd_remap(blockIdx.x, gridDim.x);
    // A-fragment addresses of the basis product (compact image, tvr_device.h)
    const int c11 = c < TVR16_BAS_ROWS1 ? c : TVR16_BAS_ROWS1 - 1;
    S16_LDS_BASE(bas0, smem + TVR16_BASH + (g * 16 + c) * 16);                                        // row block 0, k-steps 0..3: + s * 1728 (lo: + BASL - BASH)
    S16_LDS_BASE(bas1, smem + TVR16_BASH + TVR16_BAS_RB1 + (g * TVR16_BAS_ROWS1 + c11) * 16);         // row block 1
    S16_LDS_BASE(bas40, smem + (up ? TVR16_ZERO : TVR16_BASH + TVR16_BAS_S4 + (g * 16 + c) * 16));    // k-step 4: groups 2, 3 multiply the other entry's channels by zero
    S16_LDS_BASE(bas41, smem + (up ? TVR16_ZERO : TVR16_BASH + TVR16_BAS_S4 + TVR16_BAS_S4_RB1 + (g * TVR16_BAS_ROWS1 + c11) * 16));
    S16_LDS_BASE(bal40, smem + (up ? TVR16_ZERO : TVR16_BASL + TVR16_BASL_S4 + (g * 16 + c) * 16));
    S16_LDS_BASE(bal41, smem + (up ? TVR16_ZERO : TVR16_BASL + TVR16_BASL_S4 + TVR16_BAS_S4_RB1 + (g * TVR16_BAS_ROWS1 + c11) * 16));

    float4 qe_next = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned qray_next = 0;
#if S16_TICKET
    // Tiles are handed out in tickets of S16_TICKET consecutive tiles: the first ticket of a wave is its static share, every later one an atomicAdd on word 16 of the
    // scratch header (zeroed per call by zero_header with the queue counter it sits beside).  The chip's eight XCDs do not run at one speed under this kernel
    // (scripts/march_timeline.py, profiles/r05_shade_tail.txt: their workgroups finished 10.83 ... 11.71 ms after the launch with equal static shares); tickets let a
    // fast XCD take more tiles.  The atomic for ticket k + 2 is issued when ticket k + 1 is taken up, one ticket's work before its value is read: no wait on it.
    // Which wave shades an entry does not matter to the entry (q_out[ent] depends on the entry alone): pixels are unchanged, bit for bit.
    // Only for launches with at least S16_TICKET_MIN tiles per wave: a ticket is a 4-tile quantum, and a call with a handful of tiles per wave (a 4096-ray chunk: 5) is
    // better off with the static stride's one-tile quantum (157 such calls: 32.1 ms with tickets for everyone against 27.2).
    unsigned *const tk = const_cast<unsigned *>(a.counter) + 16;
    const bool dyn = n_tiles >= (long long)gridDim.x * S16_WAVES * S16_TICKET_MIN;               // (uniform over the grid)
    const int tkn = dyn ? S16_TICKET : 1;
    const long long tick0 = (long long)gridDim.x * S16_WAVES * S16_TICKET;                     // tiles covered by the static first tickets
    long long tile_first = ((long long)lblk * S16_WAVES + wave) * tkn;
    unsigned tk_pending = 0;                                                                     // lane 0: the returned value of the ticket atomic in flight
    if (dyn && lane == 0) tk_pending = atomicAdd(tk, (unsigned)S16_TICKET);
    int tk_sub = 0;
    const int tk_len = dyn ? S16_TICKET : 0x7fffffff;
    const long long tk_step = dyn ? 1 : tile_stride;
#else
    long long tile_first = (long long)lblk * S16_WAVES + wave;
#endif
    if (n_total > 0) {
        const long long e0 = tile_first * S16_TILE + c + (up ? 16 : 0);
        const long long le = e0 < n_total ? e0 : n_total - 1;
        qe_next = a.q_pos[le];
        qray_next = a.q_ray[le];
    }
    long long tile_next = 0;
    for (long long tile = tile_first; tile < n_tiles; tile = tile_next) {
#if S16_TICKET
        if (++tk_sub < tk_len) tile_next = tile + tk_step;               // (static mode: one endless "ticket" whose tiles lie a grid stride apart)
        else {
            tile_next = tick0 + (long long)__builtin_amdgcn_readfirstlane(tk_pending);
            tk_sub = 0;
            if (lane == 0 && tile_next < n_tiles) tk_pending = atomicAdd(tk, (unsigned)S16_TICKET);
        }
#else
        tile_next = tile + tile_stride;
#endif
        const long long ent = tile * S16_TILE + c + (up ? 16 : 0);        // the entry this lane stores: A in groups 0, 1, B in groups 2, 3
        const bool live = ent < n_total;
#if S16_TIMING
        unsigned long long tg0 = 0, tg1 = 0, tgW = 0, tg2 = 0, tg3 = 0, tgL = 0, tg4 = 0;
#endif
        S16_STAMP(tg0);
        // ---------------------------------------------------------------- GATHER phase ----
        const float4 qe = qe_next;
        const unsigned qray = qray_next;
        {
            const long long en = tile_next * S16_TILE + c + (up ? 16 : 0);
            const long long le = en < n_total ? en : n_total - 1;
            qe_next = a.q_pos[le];
            qray_next = a.q_ray[le];
        }
        float dA[3], dB[3];
        // Every lane gathers for ITS OWN entry only (groups 0, 1: entry A; groups 2, 3: entry B): unit u = channels 16u + 8 (g & 1) .. + 7 of the 144, i.e. of plane
        // u / 3 — nine units per lane, each inside one plane, one set of offsets / weights per plane.  A unit's fragment X therefore holds {entry A | entry B} in its
        // lower | upper half, and v_permlane32_swap(X_2s, X_2s+1) = {X_2s.lo, X_2s+1.lo} | {X_2s.hi, X_2s+1.hi} IS k-step s's B operand of tile A | of tile B, in the natural
        // k order 32 s + 8 g + j.  Unit 8 (channels 128..143) stands alone: its halves are k-step 4's operands, whose A fragments are zero in groups 2, 3.
        struct PP { unsigned o0, o1, ol; float a00, a01, a10, a11, ul, wl; };
        auto plane_params = [&](const float f[3], int p, PP &P) {
            const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;            // matMode / vecMode (tensorBase.py:168-169)
            const float x0f = floorf(f[ax]), y0f = floorf(f[bx]), l0f = floorf(f[vx]);
            const float wx = f[ax] - x0f, wy = f[bx] - y0f;
            P.wl = f[vx] - l0f; P.ul = 1.0f - P.wl;
            const float ux = 1.0f - wx, uy = 1.0f - wy;
            P.a00 = ux * uy; P.a01 = wx * uy; P.a10 = ux * wy; P.a11 = wx * wy;            // the products taps_eval forms, in its order
            // BYTE offsets (32 bits: a plane of the largest grid the ABI admits, 4097^2 texels x 192 B, is 3.2 GB) against the plane's wave-uniform base pointer:
            // global_load ... v_off, s[base] — no 64-bit address arithmetic per lane.  Opaque to the compiler, or it widens the sums (v_mad_u64_u32, quarter rate).
            const unsigned Wp = (unsigned)sc.grid[ax] + 1u;
            const unsigned cell = __umul24((unsigned)(int)y0f, Wp) + (unsigned)(int)x0f;     // grid <= 4096: both factors below 2^24
            P.o0 = (cell << 7) + (cell << 6);                                               // x 192 B per texel
            P.o1 = P.o0 + Wp * 192u;
            const unsigned l0 = (unsigned)(int)l0f;
            P.ol = (l0 << 7) + (l0 << 6);
            asm volatile("" : "+v"(P.o0), "+v"(P.o1), "+v"(P.ol));
        };
        auto ld = [](const float4 *base, unsigned byteoff, int imm) { return *(const float4 *)((const unsigned char *)base + (size_t)byteoff + imm); };   // (the immediate is added in 64 bits: it folds into the instruction)
        auto load_unit = [&](Taps &T, int p, const PP &P, unsigned qb) {               // qb: byte offset of the lane's first float4 inside the texel
            const unsigned oa = P.o0 + qb, ob = P.o1 + qb, ol = P.ol + qb;
            T.t[0][0] = ld(sc.aplane[p], oa, 0); T.t[0][1] = ld(sc.aplane[p], oa, 16); T.t[1][0] = ld(sc.aplane[p], oa, 192); T.t[1][1] = ld(sc.aplane[p], oa, 208);
            T.t[2][0] = ld(sc.aplane[p], ob, 0); T.t[2][1] = ld(sc.aplane[p], ob, 16); T.t[3][0] = ld(sc.aplane[p], ob, 192); T.t[3][1] = ld(sc.aplane[p], ob, 208);
            T.lv[0][0] = ld(sc.aline[p], ol, 0); T.lv[0][1] = ld(sc.aline[p], ol, 16); T.lv[1][0] = ld(sc.aline[p], ol, 192); T.lv[1][1] = ld(sc.aline[p], ol, 208);
        };
        // bilinear(plane) * linear(line) for the 8 channels in T: taps_eval's operations in taps_eval's order (tvr_shade_common.h), weights handed in
        auto eval_unit = [&](const Taps &T, const PP &P, float out[8]) {
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const float t0[4] = {T.t[0][gq].x, T.t[0][gq].y, T.t[0][gq].z, T.t[0][gq].w}, t1[4] = {T.t[1][gq].x, T.t[1][gq].y, T.t[1][gq].z, T.t[1][gq].w};
                const float t2[4] = {T.t[2][gq].x, T.t[2][gq].y, T.t[2][gq].z, T.t[2][gq].w}, t3[4] = {T.t[3][gq].x, T.t[3][gq].y, T.t[3][gq].z, T.t[3][gq].w};
                const float l0[4] = {T.lv[0][gq].x, T.lv[0][gq].y, T.lv[0][gq].z, T.lv[0][gq].w}, l1[4] = {T.lv[1][gq].x, T.lv[1][gq].y, T.lv[1][gq].z, T.lv[1][gq].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pv = P.a00 * t0[j];
                    pv = __builtin_fmaf(P.a01, t1[j], pv);
                    pv = __builtin_fmaf(P.a10, t2[j], pv);
                    pv = __builtin_fmaf(P.a11, t3[j], pv);
                    float qv = P.ul * l0[j];
                    qv = __builtin_fmaf(P.wl, l1[j], qv);
                    out[4 * gq + j] = pv * qv;
                }
#if S16_PIN >= 1
                asm volatile("" : "+v"(out[4 * gq]), "+v"(out[4 * gq + 1]), "+v"(out[4 * gq + 2]), "+v"(out[4 * gq + 3]));      // one pin per float4: keeps the loads' consumers together
#endif
            }
        };
        PP pp[3];
        {
            const float *rp = a.rays + (size_t)qray * 6 + 3;
            const float d0 = rp[0], d1 = rp[1], d2 = rp[2];
            halves_f(d0, dA[0], dB[0]); halves_f(d1, dA[1], dB[1]); halves_f(d2, dA[2], dB[2]);       // (rows 27..29 of BOTH tiles sit in groups 2, 3)
            const float f[3] = {unnorm(qe.x, sc.gm1[0]), unnorm(qe.y, sc.gm1[1]), unnorm(qe.z, sc.gm1[2])};
#pragma unroll
            for (int p = 0; p < 3; ++p) plane_params(f, p, pp[p]);
        }
        float rmax = 0.0f;                             // RC: running max |x| of this lane's fp16-split operands (its own entry's in the gather, both tiles' below)
        Frag hf[9];
        uint4 balg0, balg1;                           // lo parts of basis k-step 3 (global)
        {
            Taps T[S16_GDEPTH + 1];
            const unsigned qb0 = 32u * (unsigned)(g & 1);                                  // bytes: unit u reads float4s 4 (u % 3) + 2 (g & 1), + 1 of the texel
            auto issue = [&](int u) { load_unit(T[u % (S16_GDEPTH + 1)], u / 3, pp[u / 3], qb0 + 64u * (unsigned)(u % 3)); };
#pragma unroll
            for (int u0 = 0; u0 < S16_GDEPTH; ++u0) issue(u0);
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                if (u + S16_GDEPTH < 9) issue(u + S16_GDEPTH);
                if (u == 7) {                          // the tile's last global loads ride behind the last taps
                    unsigned boff = (unsigned)(lane * 16);
                    asm volatile("" : "+v"(boff));
                    balg0 = *(const uint4 *)((const unsigned char *)sc.basg16 + boff);
                    balg1 = *(const uint4 *)((const unsigned char *)sc.basg16 + (boff + TVR16_FRAG));
                }
                float hv[8];
                eval_unit(T[u % (S16_GDEPTH + 1)], pp[u / 3], hv);
                if (RC) {
#pragma unroll
                    for (int j = 0; j < 8; j += 2) rmax = absmax2(hv[j], hv[j + 1], rmax);
                    asm volatile("" : "+v"(rmax));
                }
                hf[u] = S16_SPLIT8(hv);
                asm volatile("" : "+v"(hf[u].hi.x), "+v"(hf[u].hi.y), "+v"(hf[u].hi.z), "+v"(hf[u].hi.w), "+v"(hf[u].lo.x), "+v"(hf[u].lo.y), "+v"(hf[u].lo.z), "+v"(hf[u].lo.w));
                S16_SB;
            }
        }
        // the B operands of the basis product: k-step s < 4 of tile A | B = the lower | upper halves of units 2s and 2s + 1 side by side; k-step 4 = unit 8's halves
        Frag hA[5], hB[5];
        {
            auto sw = [](unsigned x, unsigned y, unsigned &o0, unsigned &o1) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); o0 = r[0]; o1 = r[1]; };
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const Frag &X = hf[2 * s4], &Y = hf[2 * s4 + 1];
                sw(X.hi.x, Y.hi.x, hA[s4].hi.x, hB[s4].hi.x); sw(X.hi.y, Y.hi.y, hA[s4].hi.y, hB[s4].hi.y); sw(X.hi.z, Y.hi.z, hA[s4].hi.z, hB[s4].hi.z); sw(X.hi.w, Y.hi.w, hA[s4].hi.w, hB[s4].hi.w);
                sw(X.lo.x, Y.lo.x, hA[s4].lo.x, hB[s4].lo.x); sw(X.lo.y, Y.lo.y, hA[s4].lo.y, hB[s4].lo.y); sw(X.lo.z, Y.lo.z, hA[s4].lo.z, hB[s4].lo.z); sw(X.lo.w, Y.lo.w, hA[s4].lo.w, hB[s4].lo.w);
            }
            const Frag &Z = hf[8];
            halves_u(Z.hi.x, hA[4].hi.x, hB[4].hi.x); halves_u(Z.hi.y, hA[4].hi.y, hB[4].hi.y); halves_u(Z.hi.z, hA[4].hi.z, hB[4].hi.z); halves_u(Z.hi.w, hA[4].hi.w, hB[4].hi.w);
            halves_u(Z.lo.x, hA[4].lo.x, hB[4].lo.x); halves_u(Z.lo.y, hA[4].lo.y, hB[4].lo.y); halves_u(Z.lo.z, hA[4].lo.z, hB[4].lo.z); halves_u(Z.lo.w, hA[4].lo.w, hB[4].lo.w);
        }
        float rmaxA = 0.0f, rmaxB = 0.0f;
        if (RC) {                                      // the gather's maximum belongs to the lane's own entry
            rmaxA = up ? 0.0f : rmax;
            rmaxB = up ? rmax : 0.0f;
        }

        // ---------------------------------------------------------------- phase boundary + basis product ----
        float FA[8], FB[8];                            // base values: r < 4: row 4g + r; r >= 4: row 16 + 4g + r - 4 of the feature tile
        float GA[4] = {0.f, 0.f, 0.f, 0.f}, GB[4] = {0.f, 0.f, 0.f, 0.f}, dotA = 0.0f, dotB = 0.0f;       // REF: head outputs of this lane's rows; -d.n of both entries
        {
            constexpr int LO = TVR16_BASL - TVR16_BASH;
            AF16 br[4];                                // ring over the ten (k-step, row block) fragments
            auto bld = [&](int q) {                    // q = 2 s + rb
                const int s = q >> 1, rb = q & 1;
                AF16 &A = br[q & 3];
                if (s < 4) {
                    A.h = *(const uint4 *)((rb ? bas1 : bas0) + s * TVR16_BAS_STEP);
                    if (s < 3) A.l = *(const uint4 *)((rb ? bas1 : bas0) + LO + s * TVR16_BAS_STEP);
                    else A.l = rb ? balg1 : balg0;
                } else {
                    A.h = *(const uint4 *)(rb ? bas41 : bas40);
                    A.l = *(const uint4 *)(rb ? bal41 : bal40);
                }
            };
            bld(0); bld(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            S16_SB;
            AF16 rg[REF ? 5 : 1];                      // REF: the heads' fragments (k-steps 0..4), in flight under the token wait and the 60 MFMAs of the feature blocks
            float4 rbias = make_float4(0.f, 0.f, 0.f, 0.f);
            if (REF) {
                unsigned roff = (unsigned)(lane * 16);
                asm volatile("" : "+v"(roff));
#pragma unroll
                for (int s5 = 0; s5 < 5; ++s5) {
                    rg[s5].h = *(const uint4 *)((const unsigned char *)sc.refg16 + (roff + (unsigned)((2 * s5) * TVR16_FRAG)));
                    rg[s5].l = *(const uint4 *)((const unsigned char *)sc.refg16 + (roff + (unsigned)((2 * s5 + 1) * TVR16_FRAG)));
                }
                rbias = *(const float4 *)((const unsigned char *)sc.refg16 + 10 * TVR16_FRAG + 16 * g);      // head biases of rows 4g .. 4g+3 (zeros in groups 2, 3)
            }
            S16_STAMP(tg1);
            {                                          // take the SIMD's matrix token (bounded: a stuck token costs speed, never a hang or a pixel)
#if S16_MTOKEN
                int got, n = 0;
                do {
                    int r = 1;
                    if (lane == 0) r = atomicCAS(mtok, 0, 1);
                    got = __builtin_amdgcn_readfirstlane(r);
                    if (got) __builtin_amdgcn_s_sleep(S16_MSLEEP);
                } while (got && ++n < S16_TOKEN_SPINS);
                have_tok = !got;
#endif
                __builtin_amdgcn_s_setprio(S16_PRIO_M);
            }
            S16_STAMP(tgW);
            f32x4 aF[2][2] = {{f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}};
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int s = q >> 1, rb = q & 1;
                if (q + 2 < 10) bld(q + 2);
                mfma6(br[q & 3], hA[s], hB[s], aF[rb][0], aF[rb][1]);
#if S16_SCHED
                if (q + 2 < 10) { if (((q + 2) >> 1) == 3) S16_SG_DSR(1); else S16_SG_DSR(2); }
                S16_SG_MFMA(6);
#endif
                S16_SB;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { FA[r] = aF[0][0][r]; FA[4 + r] = aF[1][0][r]; FB[r] = aF[0][1][r]; FB[4 + r] = aF[1][1][r]; }
            if (REF) {                                 // the heads: rows 4g + r of the third block = {normal x y z, tint} in group 0, {diffuse r g b, rho} in group 1
                f32x4 gA = f32x4{rbias.x, rbias.y, rbias.z, rbias.w}, gB = gA;
#pragma unroll
                for (int s5 = 0; s5 < 5; ++s5) mfma6(rg[s5], hA[s5], hB[s5], gA, gB);
#pragma unroll
                for (int r = 0; r < 4; ++r) { GA[r] = gA[r]; GB[r] = gB[r]; }
            }
        }
        S16_STAMP(tg2);
        if (REF) {
            // REFTensoRF.execute :215-227 in every lane for both entries (the normal sits in group 0: broadcast): normalise, d = -view, dot = d.n,
            // reflection = 2 dot n - d; the MLP takes the reflection as its direction and -dot as input 0 (row 30: only its t = 0 slot has a weight)
            auto reflect = [&](const float G[4], float dv[3], float &dotin) {
                const float n0 = bcast_g0(G[0]), n1 = bcast_g0(G[1]), n2 = bcast_g0(G[2]);
                // n / max(|n|, 1e-15) with ONE v_rsq_f32 (1 ulp) instead of an IEEE sqrt and three IEEE divisions (~45 vector instructions per entry pair in a kernel whose
                // issue port is full): the normal moves by <= 2 ulp, the colour by < 1e-6
                const float inv = __builtin_amdgcn_rsqf(fmaxf((n0 * n0 + n1 * n1) + n2 * n2, 1e-30f));
                const float nx = n0 * inv, ny = n1 * inv, nz = n2 * inv;
                const float dx = -dv[0], dy = -dv[1], dz = -dv[2];
                const float dot = (dx * nx + dy * ny) + dz * nz;
                dv[0] = 2.0f * dot * nx - dx; dv[1] = 2.0f * dot * ny - dy; dv[2] = 2.0f * dot * nz - dz;
                dotin = -dot;
            };
            reflect(GA, dA, dotA);
            reflect(GB, dB, dotB);
        }
        // rows 27 (group 2, r = 7), 28, 29 (group 3, r = 4, 5): the view direction; row 30 unused; row 31 (group 3, r = 7) the constant 1 whose column is b1
        if (g == 2) { FA[7] = dA[0]; FB[7] = dB[0]; }
        if (g == 3) { FA[4] = dA[1]; FA[5] = dA[2]; FA[6] = REF ? dotA : 0.0f; FA[7] = 1.0f; FB[4] = dB[1]; FB[5] = dB[2]; FB[6] = REF ? dotB : 0.0f; FB[7] = 1.0f; }
        if (RC) {
#pragma unroll
            for (int r = 0; r < 8; r += 2) { rmaxA = absmax2(FA[r], FA[r + 1], rmaxA); rmaxB = absmax2(FB[r], FB[r + 1], rmaxB); }
            asm volatile("" : "+v"(rmaxA), "+v"(rmaxB));
        }

        // ---------------------------------------------------------------- layers 1 and 2 ----
        f32x4 acc1[8][2];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) { acc1[rb][0] = f32x4{0, 0, 0, 0}; acc1[rb][1] = f32x4{0, 0, 0, 0}; }
        float s3A[3] = {0.f, 0.f, 0.f}, s3B[3] = {0.f, 0.f, 0.f};       // layer 3's running sums (this lane's 32 hidden units of each entry)
        f32x4 a2lastA, a2lastB;                                          // layer-2 accumulators of row block 7: their layer 3 runs behind the token hand-over
        {
            float SA[8], CA[8], SB_[8], CB[8];
            // slot i = 8 s + j of a lane is derived value (i % 5) of base value i / 5; sin / cos of a base value are taken in the k-step that first needs them
            auto l1_frag = [&](int s, const float *F, float *S1, float *C1, Frag &b) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * s + j, r = i / 5, t = i % 5;
                    if (t == 0) sincos_pe16(F[r], S1[r], C1[r]);
                    v[j] = t == 0 ? F[r] : (t == 1 ? S1[r] : (t == 2 ? 2.0f * S1[r] * C1[r] : (t == 3 ? C1[r] : __builtin_fmaf(-2.0f * S1[r], S1[r], 1.0f))));
                }
                b = S16_SPLIT8(v);
            };
            auto relu_frag = [&](int ks, int ct, Frag &b) {           // layer 2's B fragment of k-step ks: the layer-1 accumulators of row blocks 2 ks, 2 ks + 1
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = relu_f(acc1[2 * ks][ct][j]); v[4 + j] = relu_f(acc1[2 * ks + 1][ct][j]); }
                if (RC) {
                    float &rm = ct ? rmaxB : rmaxA;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) rm = fmaxf(fmaxf(v[j], v[j + 1]), rm);
                    asm volatile("" : "+v"(rm));
                }
                b = S16_SPLIT8(v);
            };
            S16_LDS_BASE(W1Hb, smem + TVR16_W1H + lane * 16);
            S16_LDS_BASE(W1Lb, smem + TVR16_W1L + lane * 16);
            S16_LDS_BASE(W2Hb, smem + TVR16_W2H + lane * 16);
            S16_LDS_BASE(W2Lb, smem + TVR16_W2L + lane * 16);
            AF16 ring[S16_RN];
            AF16 dfix[2];
            if (S16_DIAG_LDS == 3) { dfix[0].h = *(const uint4 *)(W1Hb); dfix[0].l = *(const uint4 *)(W1Lb); dfix[1].h = *(const uint4 *)(W1Hb + TVR16_FRAG); dfix[1].l = *(const uint4 *)(W1Lb + TVR16_FRAG); }
            Frag bA, bB, nA, nB;
            Frag frA[4], frB[4];
#pragma unroll
            for (int q0 = 0; q0 < S16_PD; ++q0) { ring[q0].h = *(const uint4 *)(W1Hb + q0 * TVR16_FRAG); ring[q0].l = *(const uint4 *)(W1Lb + q0 * TVR16_FRAG); }
            l1_frag(0, FA, SA, CA, bA);
            l1_frag(0, FB, SB_, CB, bB);
            S16_SB;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) {
                    const int q = 8 * s + rb;
                    if (S16_DIAG_LDS == 2 || (S16_DIAG_LDS == 1 && ((q + S16_PD) & 1))) {        // timing stand-in (wrong pictures): no / every other fragment read
                    } else if (S16_DIAG_LDS == 4) {    // real reads into the ring, real waits — of two fragment addresses only (same data every other step)
                        ring[(q + S16_PD) % S16_RN].h = *(const uint4 *)(W1Hb + ((q + S16_PD) & 1) * TVR16_FRAG);
                        ring[(q + S16_PD) % S16_RN].l = *(const uint4 *)(W1Lb + ((q + S16_PD) & 1) * TVR16_FRAG);
                    } else if (q + S16_PD < 40) {
                        ring[(q + S16_PD) % S16_RN].h = *(const uint4 *)(W1Hb + (q + S16_PD) * TVR16_FRAG);
                        ring[(q + S16_PD) % S16_RN].l = *(const uint4 *)(W1Lb + (q + S16_PD) * TVR16_FRAG);
                    } else {                           // layer 2's first fragments ride in the ring behind layer 1's last
                        const int q2 = q + S16_PD - 40;                  // (row block 0, k-step q2)
                        ring[(q + S16_PD) % S16_RN].h = *(const uint4 *)(W2Hb + (q2 * 8) * TVR16_FRAG);
                        ring[(q + S16_PD) % S16_RN].l = *(const uint4 *)(W2Lb + (q2 * 8) * TVR16_FRAG);
                    }
                    if (S16_DIAG_LDS == 3) {           // every read issued, waited for and written to registers; the MFMAs take two fixed fragments
                        { const uint4 rh = ring[q % S16_RN].h, rl = ring[q % S16_RN].l; asm volatile("" :: "v"(rh.x), "v"(rh.y), "v"(rh.z), "v"(rh.w), "v"(rl.x), "v"(rl.y), "v"(rl.z), "v"(rl.w)); }
                        mfma6(dfix[q & 1], bA, bB, acc1[rb][0], acc1[rb][1]);
                    } else
                    mfma6(ring[S16_DIAG_LDS == 2 ? (q & 1) : (S16_DIAG_LDS == 1 ? (q & ~1) % S16_RN : q % S16_RN)], bA, bB, acc1[rb][0], acc1[rb][1]);
                }
                if (s + 1 < 5) { l1_frag(s + 1, FA, SA, CA, nA); l1_frag(s + 1, FB, SB_, CB, nB); }
                else { relu_frag(0, 0, frA[0]); relu_frag(0, 1, frB[0]); }              // (acc1[0], acc1[1] are complete after row block 1 of this k-step)
#if S16_SCHED
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) {
                    if (s + 1 < 5) {                   // {read} M V M VV {read} M V M VV M V M VV: nine VALU per row block, 72 per k-step
                        S16_SG_DSR(1); S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                        S16_SG_DSR(1); S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                        S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                    } else if (rb < 2) {
                        S16_SG_DSR(1); S16_SG_MFMA(2); S16_SG_DSR(1); S16_SG_MFMA(4);
                    } else {                           // relu(acc1[0..1]) + split under row blocks 2..7: 48 VALU over 36 MFMAs
                        S16_SG_DSR(1); S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                        S16_SG_DSR(1); S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                        S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(1);
                    }
                }
#endif
                if (s + 1 < 5) { bA = nA; bB = nB; }
                S16_SB;
            }
            S16_STAMP(tg3);
            // layer 2, row block by row block: position q = 4 rb + ks of the (row block, k-step) sequence, fragment (ks, rb) at (8 ks + rb) KB of W2's images;
            // the ring continues from layer 1 (positions 40 + q)
            {
                S16_LDS_BASE(W3b, smem + TVR16_W3 + 16 * g);
                S16_LDS_BASE(B2b, smem + TVR16_B2 + 16 * g);
                f32x4 a2A, a2B, a2pA = f32x4{0, 0, 0, 0}, a2pB = f32x4{0, 0, 0, 0};
                float4 w3[3];
                auto l3_block = [&](const f32x4 &xA, const f32x4 &xB) {       // hidden units 16 rb + 4 g .. + 3 (tensorBase.py:83-84: Linear(128 -> 3) on relu(h2))
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ya = relu_f(xA[r]), yb = relu_f(xB[r]);
#pragma unroll
                        for (int c3 = 0; c3 < 3; ++c3) {
                            const float w = r == 0 ? w3[c3].x : (r == 1 ? w3[c3].y : (r == 2 ? w3[c3].z : w3[c3].w));
                            s3A[c3] = __builtin_fmaf(ya, w, s3A[c3]);
                            s3B[c3] = __builtin_fmaf(yb, w, s3B[c3]);
                        }
                    }
                    asm volatile("" : "+v"(s3A[0]), "+v"(s3A[1]), "+v"(s3A[2]), "+v"(s3B[0]), "+v"(s3B[1]), "+v"(s3B[2]));
                };
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) {
                    {
                        const float4 bv = *(const float4 *)(B2b + 64 * rb);                  // b2 of rows 16 rb + 4 g .. + 3: the initial accumulator of both tiles
                        a2A = f32x4{bv.x, bv.y, bv.z, bv.w}; a2B = a2A;
                    }
                    if (rb > 0) {
#pragma unroll
                        for (int c3 = 0; c3 < 3; ++c3) w3[c3] = *(const float4 *)(W3b + c3 * 512 + 64 * (rb - 1));
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int q = 4 * rb + ks, qn = q + S16_PD;
                        if (S16_DIAG_LDS == 2 || ((S16_DIAG_LDS == 1 || S16_DIAG_LDS == 5) && (qn & 1))) {
                        } else if (S16_DIAG_LDS == 4) {
                            if (qn < 32) {
                                ring[(40 + qn) % S16_RN].h = *(const uint4 *)(W1Hb + (qn & 1) * TVR16_FRAG);
                                ring[(40 + qn) % S16_RN].l = *(const uint4 *)(W1Lb + (qn & 1) * TVR16_FRAG);
                            }
                        } else if (qn < 32) {
                            ring[(40 + qn) % S16_RN].h = *(const uint4 *)(W2Hb + (8 * (qn & 3) + (qn >> 2)) * TVR16_FRAG);
                            ring[(40 + qn) % S16_RN].l = *(const uint4 *)(W2Lb + (8 * (qn & 3) + (qn >> 2)) * TVR16_FRAG);
                        }
                        if (S16_DIAG_LDS == 3) {
                            { const uint4 rh = ring[(40 + q) % S16_RN].h, rl = ring[(40 + q) % S16_RN].l; asm volatile("" :: "v"(rh.x), "v"(rh.y), "v"(rh.z), "v"(rh.w), "v"(rl.x), "v"(rl.y), "v"(rl.z), "v"(rl.w)); }
                            mfma6(dfix[q & 1], frA[ks], frB[ks], a2A, a2B);
                        } else
                        mfma6(ring[S16_DIAG_LDS == 2 ? (q & 1) : ((S16_DIAG_LDS == 1 || S16_DIAG_LDS == 5) ? (40 + (q & ~1)) % S16_RN : (40 + q) % S16_RN)], frA[ks], frB[ks], a2A, a2B);
                        if (rb == 0 && ks + 1 < 4) { relu_frag(ks + 1, 0, frA[ks + 1]); relu_frag(ks + 1, 1, frB[ks + 1]); }
                    }
                    if (rb > 0) l3_block(a2pA, a2pB);
#if S16_SCHED
                    if (rb == 0) {
                        S16_SG_DSR(1);                                          // b2
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            if (ks < 3) {                                       // the next k-step's two fragments: 48 VALU under six MFMAs
                                S16_SG_DSR(1); S16_SG_MFMA(1); S16_SG_VALU(8); S16_SG_MFMA(1); S16_SG_VALU(8);
                                S16_SG_DSR(1); S16_SG_MFMA(1); S16_SG_VALU(8); S16_SG_MFMA(1); S16_SG_VALU(8);
                                S16_SG_MFMA(1); S16_SG_VALU(8); S16_SG_MFMA(1); S16_SG_VALU(8);
                            } else {
                                S16_SG_DSR(1); S16_SG_MFMA(2); S16_SG_DSR(1); S16_SG_MFMA(4);
                            }
                        }
                    } else {
                        S16_SG_DSR(4);                                          // b2, W3's three quads
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {                        // 32 VALU of layer 3 over 24 MFMAs
                            if (4 * rb + ks + S16_PD < 32) { S16_SG_DSR(1); }
                            S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                            if (4 * rb + ks + S16_PD < 32) { S16_SG_DSR(1); }
                            S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                            S16_SG_MFMA(1); S16_SG_VALU(1); S16_SG_MFMA(1); S16_SG_VALU(2);
                        }
                    }
#endif
                    a2pA = a2A; a2pB = a2B;
                    S16_SB;
                }
                a2lastA = a2pA; a2lastB = a2pB;
            }
        }
        S16_STAMP(tgL);
        if (have_tok && lane == 0) atomicExch(mtok, 0);
        __builtin_amdgcn_s_setprio(S16_PRIO_G);
        // ---------------------------------------------------------------- finish: row block 7's layer 3, the sums over the lane groups, sigmoid, store ----
        {
            const float *W3 = (const float *)(smem + TVR16_W3 + 16 * g);
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
                const float4 w = *(const float4 *)(W3 + c3 * 128 + 16 * 7);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wr = r == 0 ? w.x : (r == 1 ? w.y : (r == 2 ? w.z : w.w));
                    s3A[c3] = __builtin_fmaf(relu_f(a2lastA[r]), wr, s3A[c3]);
                    s3B[c3] = __builtin_fmaf(relu_f(a2lastB[r]), wr, s3B[c3]);
                }
            }
            // (the order above interleaves c3 outermost where l3_block has r outermost: per sum s3X[c3] the additions run r = 0..3 either way)
            const float4 b3 = *(const float4 *)(smem + TVR16_B3);
            float r0 = group_sum2(s3A[0], s3B[0]), r1 = group_sum2(s3A[1], s3B[1]), r2 = group_sum2(s3A[2], s3B[2]);
#if S16_FAST_SIGMOID
            // 1 / (1 + exp(-x)) with v_exp_f32 / v_rcp_f32 (1 ulp each) instead of expf + an IEEE division: ~25 fewer VALU instructions per tile, < 2e-7 in the colour
            r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (r0 + b3.x)));
            r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (r1 + b3.y)));
            r2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (r2 + b3.z)));
#else
            r0 = sigmoid_f(r0 + b3.x); r1 = sigmoid_f(r1 + b3.y); r2 = sigmoid_f(r2 + b3.z);
#endif
            if (REF) {                                 // REFTensoRF.py:232  specular_tint * clamp(rgb_s, 0) + rgb_d, for the entry this lane stores (group 0: A, group 2: B)
                // tint sits in group 0 (row 3), rgb_d in group 1 (rows 4..6): entry A's into group 0 by one row swap, entry B's into group 2 by a row swap and a half swap
                float tA = GA[3], tB, tdummy;
                halves_f(GB[3], tB, tdummy);                                     // group 0's tint of entry B in both halves
                const float a0 = from_g1(GA[0]), a1 = from_g1(GA[1]), a2 = from_g1(GA[2]);
                float b0, b1, b2, bd;
                halves_f(from_g1(GB[0]), b0, bd); halves_f(from_g1(GB[1]), b1, bd); halves_f(from_g1(GB[2]), b2, bd);
                const float tint = fmaxf(up ? tB : tA, 0.0f);
                r0 = tint * fmaxf(r0, 0.0f) + (up ? b0 : a0);
                r1 = tint * fmaxf(r1, 0.0f) + (up ? b1 : a1);
                r2 = tint * fmaxf(r2, 0.0f) + (up ? b2 : a2);
            }
            if (RC) {                                  // an operand of this entry left fp16's range: the colour is NaN, not a clipped product
                const float m = group_max2(rmaxA, rmaxB);
                if (!(m < TVR_F16_MAX)) r0 = r1 = r2 = __builtin_nanf("");
            }
            if (live && (g & 1) == 0) a.q_out[ent] = make_float4(r0, r1, r2, qe.w);        // group 0 stores entry A, group 2 entry B (qe is the lane's own record)
        }
        S16_STAMP(tg4);
#if S16_TIMING
        tsum[1] += tg1 - tg0; tsum[6] += tgW - tg1; tsum[2] += tg2 - tgW; tsum[3] += tg3 - tg2; tsum[4] += tgL - tg3; tsum[7] += tg4 - tgL;
#endif
    }
#if S16_TIMING
    if (a.stats && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)&a.stats[8 + i], tsum[i]);     // -, gather, basis, L1 (+PE), L2, -, wait for the matrix token, hand-over + layer 3 + store
#endif
#ifdef TVR_MARCH_TIMELINE                               // the timeline build (scripts/march_timeline.py): stats[32 + 8 b + 6 / 7] = this kernel's group b start / last wave end, 100 MHz ticks
    if (a.stats && lane == 0) atomicMax((unsigned long long *)&a.stats[32 + 8 * blockIdx.x + 7], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    if (a.stats && tid == 0) a.stats[32 + 8 * blockIdx.x + 6] = ref0;
#ifdef S16_XCDCLK                                       // ... and, over the march's chunk / ray counts, wave 0's own shader-clock and reference ticks: the clock of each workgroup's CU
    if (a.stats && tid == 0) { a.stats[32 + 8 * blockIdx.x + 4] = __builtin_amdgcn_s_memtime() - clk0; a.stats[32 + 8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime() - ref0; }
#endif
#endif
    if (a.stats && tid == 0) {
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_SHADE_CLK], __builtin_amdgcn_s_memtime() - clk0);
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_SHADE_REF], __builtin_amdgcn_s_memrealtime() - ref0);
    }
    if (a.stats && blockIdx.x == 0 && tid == 0) atomicAdd((unsigned long long *)&a.stats[TVR_STAT_APP], (unsigned long long)n_total);
}

hipError_t launch_shade16(const SceneDev &sc, const ShadeArgs &a, hipStream_t stream)
{
    const int lds = TVR16_IMAGE_BYTES + 16;
    const bool rc = sc.range_check != 0, ref = sc.variant == 1;
    const void *kf = ref ? (rc ? (const void *)shade16_kernel<true, true> : (const void *)shade16_kernel<false, true>)
                         : (rc ? (const void *)shade16_kernel<true, false> : (const void *)shade16_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    unsigned grid = 256;       // one workgroup per CU (the LDS holds the weights), persistent over 32-entry tiles
#ifdef TVR_EXP_GRID
    if (const char *gs = getenv("TVR_EXP_GRID_SHADE")) { const long long v = atoll(gs); if (v > 0 && v < 256) grid = (unsigned)v; }
#endif
    if (ref) {
        if (rc) hipLaunchKernelGGL((shade16_kernel<true, true>), dim3(grid), dim3(S16_THREADS), lds, stream, sc, a);
        else hipLaunchKernelGGL((shade16_kernel<false, true>), dim3(grid), dim3(S16_THREADS), lds, stream, sc, a);
    } else {
        if (rc) hipLaunchKernelGGL((shade16_kernel<true, false>), dim3(grid), dim3(S16_THREADS), lds, stream, sc, a);
        else hipLaunchKernelGGL((shade16_kernel<false, false>), dim3(grid), dim3(S16_THREADS), lds, stream, sc, a);
    }
    return hipGetLastError();
}

// ---- weights -> the fragment images (tvr_device.h, TVR16_*).  One thread per (fragment, lane, j). ----
//  mode 0: W1 [5][8] fragments; mode 1: W2 [4][8]; mode 2: basis [5][2] (compact, hi -> the image, lo -> the image or the global k-step-3 fragments)
//  mode 3 (REFTensoRF): the heads' fragments -> refg [5 k-steps][hi | lo] as full fragments: row ci < 8 = {normal 0..2, specular, diffuse 0..2, rho}, k natural as mode 2
struct RefHeads { const float *W[4]; };      // tvr_scene_params.ref_W order: normal [3,144], diffuse [3,144], specular [1,144], rho [1,144]
__global__ __launch_bounds__(256) void pack16_kernel(const float *__restrict__ W, const float *__restrict__ bias, unsigned char *__restrict__ img,
                                                     unsigned char *__restrict__ basg, int mode, const MlpShape sh, const RefHeads rh, int ref)
{
    const int nfr = mode == 0 ? 40 : (mode == 1 ? 32 : (mode == 3 ? 5 : 10));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfr * 512) return;
    const int fr = i >> 9, lane = (i >> 3) & 63, j = i & 7, ci = lane & 15, g = lane >> 4;
    float w = 0.0f;
    unsigned short *ph = nullptr, *pl = nullptr;
    if (mode == 0) {
        const int s = fr >> 3, rb = fr & 7, row = 16 * rb + ci, ii = 8 * s + j, r = ii / 5, t = ii % 5;
        const int cbase = r < 4 ? 4 * g + r : 16 + 4 * g + (r - 4);
        int idx = ref_in_index(cbase, t, sh.fea_pe, sh.view_pe);
        if (ref) idx = cbase == 30 ? (t == 0 ? 0 : -1) : (idx >= 0 ? idx + 1 : -1);        // MLPRender_Fea_Ref (REFTensoRF.py:19-24): input 0 is the dot product, everything else moves up by one
        if (idx >= 0 && row < sh.featureC) w = W[(size_t)row * sh.n_in + idx];
        if (cbase == 31 && t == 0 && row < sh.featureC) w = bias[row];          // the constant-1 input: b1 rides in the weight image
        ph = (unsigned short *)(img + TVR16_W1H + fr * TVR16_FRAG + lane * 16) + j;
        pl = (unsigned short *)(img + TVR16_W1L + fr * TVR16_FRAG + lane * 16) + j;
    } else if (mode == 1) {
        const int s = fr >> 3, rb = fr & 7, row = 16 * rb + ci;
        const int u = j < 4 ? 32 * s + 4 * g + j : 32 * s + 16 + 4 * g + (j - 4);
        if (row < sh.featureC && u < sh.featureC) w = W[(size_t)row * sh.featureC + u];
        ph = (unsigned short *)(img + TVR16_W2H + fr * TVR16_FRAG + lane * 16) + j;
        pl = (unsigned short *)(img + TVR16_W2L + fr * TVR16_FRAG + lane * 16) + j;
    } else if (mode == 3) {
        const int s = fr, k = 32 * s + 8 * g + j;
        if (k < TVR_KAPP && ci < 8) {
            const float *Wh = ci < 3 ? rh.W[0] + (size_t)ci * TVR_KAPP : (ci == 3 ? rh.W[2] : (ci < 7 ? rh.W[1] + (size_t)(ci - 4) * TVR_KAPP : rh.W[3]));
            w = Wh[k];
        }
        ph = (unsigned short *)(basg + (2 * s) * TVR16_FRAG + lane * 16) + j;          // (`basg` is the heads' buffer in this mode)
        pl = (unsigned short *)(basg + (2 * s + 1) * TVR16_FRAG + lane * 16) + j;
    } else {
        // k slot (s, g, j) = 32 s + 8 g + j, the kernels' natural k = 48 * plane + channel (k-step 4: groups 0, 1 only — the kernel reads zeros in groups 2, 3)
        const int s = fr >> 1, rb = fr & 1, row = 16 * rb + ci, k = 32 * s + 8 * g + j;
        if (k < TVR_KAPP && row < TVR_APPDIM) {
            const int pl_ = k / TVR_CA, ch = k - pl_ * TVR_CA;                  // basis_mat's column = its plane's offset + channel
            if (ch < sh.app_n_comp[pl_]) w = W[(size_t)row * sh.k_app + sh.app_off[pl_] + ch];
        }
        const bool in_img = (rb == 0 || ci < TVR16_BAS_ROWS1) && (s < 4 || g < 2);
        const int off = (s < 4 ? s * TVR16_BAS_STEP : TVR16_BAS_S4) + (rb == 0 ? (g * 16 + ci) * 16 : (s < 4 ? TVR16_BAS_RB1 : TVR16_BAS_S4_RB1) + (g * TVR16_BAS_ROWS1 + ci) * 16);
        if (in_img) {
            ph = (unsigned short *)(img + TVR16_BASH + off) + j;
            if (s != 3) pl = (unsigned short *)(img + TVR16_BASL + (s < 3 ? off : off - TVR16_BAS_S4 + TVR16_BASL_S4)) + j;
        }
        if (s == 3) pl = (unsigned short *)(basg + rb * TVR16_FRAG + lane * 16) + j;       // full fragments: rows >= 27 are zero
    }
    unsigned hi, lo;
    split2(w, 0.0f, hi, lo);
    if (ph) *ph = (unsigned short)hi;
    if (pl) *pl = (unsigned short)lo;
}

// the heads' biases as the initial accumulators of the third row block: [4 groups][4 rows] floats behind the ten fragments (groups 2, 3: zeros)
__global__ void pack16_ref_bias_kernel(const float *__restrict__ bn, const float *__restrict__ bd, const float *__restrict__ bs, const float *__restrict__ br, float *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= 16) return;
    out[i] = i < 3 ? bn[i] : (i == 3 ? bs[0] : (i < 7 ? bd[i - 4] : (i == 7 ? br[0] : 0.0f)));
}

hipError_t launch_pack16(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3, const float *basis, void *img, void *basg,
                         const MlpShape &sh, hipStream_t stream, const float *const *ref_W, const float *const *ref_b, void *refg)
{
    unsigned char *im = (unsigned char *)img;
    hipError_t e;
    // b2, b3, W3, the zero bytes (hidden units >= featureC stay zero), then the copies — as kernels: tvr_scene_update runs inside captured training steps (tvr_step.hip)
    if ((e = launch_zero_f32((float *)(im + TVR16_B2), (TVR16_BASH - TVR16_B2) / 4, stream)) != hipSuccess) return e;
    if ((e = launch_copy_f32((float *)(im + TVR16_B2), b2, sh.featureC, stream)) != hipSuccess) return e;
    if ((e = launch_copy_f32((float *)(im + TVR16_B3), b3, 3, stream)) != hipSuccess) return e;
    for (int r = 0; r < 3; ++r)
        if ((e = launch_copy_f32((float *)(im + TVR16_W3 + r * 512), W3 + (size_t)r * sh.featureC, sh.featureC, stream)) != hipSuccess) return e;
    RefHeads rh = {{nullptr, nullptr, nullptr, nullptr}};
    const int ref = refg != nullptr;
    if (ref) for (int i = 0; i < 4; ++i) rh.W[i] = ref_W[i];
    hipLaunchKernelGGL(pack16_kernel, dim3(40 * 512 / 256), dim3(256), 0, stream, W1, b1, im, (unsigned char *)basg, 0, sh, rh, ref);
    hipLaunchKernelGGL(pack16_kernel, dim3(32 * 512 / 256), dim3(256), 0, stream, W2, nullptr, im, (unsigned char *)basg, 1, sh, rh, ref);
    hipLaunchKernelGGL(pack16_kernel, dim3(10 * 512 / 256), dim3(256), 0, stream, basis, nullptr, im, (unsigned char *)basg, 2, sh, rh, ref);
    if (ref) {
        hipLaunchKernelGGL(pack16_kernel, dim3(5 * 512 / 256), dim3(256), 0, stream, nullptr, nullptr, im, (unsigned char *)refg, 3, sh, rh, ref);
        hipLaunchKernelGGL(pack16_ref_bias_kernel, dim3(1), dim3(64), 0, stream, ref_b[0], ref_b[1], ref_b[2], ref_b[3], (float *)((unsigned char *)refg + 10 * TVR16_FRAG));
    }
    return hipGetLastError();
}
