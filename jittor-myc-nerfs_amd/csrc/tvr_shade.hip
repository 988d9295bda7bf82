// tvr_shade.hip — kernel 2 of the render path: appearance VM lookup + basis + positional encoding + 3-layer MLP.
//
// Work it replaces in the reference (paths relative to /root/reference/tensorf-myc/):
//   models/tensoRF.py:228-244 compute_appfeature (6 grid_samples + Linear(144->27, no bias)),
//   models/tensorBase.py:9-15 positional_encoding, :76-86 MLPRender_Fea.execute (150->128->128->3, sigmoid).
//
// Design (CDNA4): every wave is an independent pipeline over 32 queue entries, entry = MFMA column (lane & 31), and the
// whole chain stays in registers — no LDS round trip, no barrier in the tile loop:
//   gather   lane (e, h) fetches 8 channels (2 x float4) of each tap of entry e per k-step; the interpolated plane*line
//            products ARE the B fragment of the basis product
//   basis    F^T[32 x 32e]   = Bas[32 x 144] · h^T          A (basis) from L1-cached global fragments
//   PE       lane (e, h) owns 16 base values (its accumulator rows); [v, sin v, sin 2v, cos v, cos 2v] of them, in that
//            order, are the B fragments of layer 1 (W1's columns are permuted to this order at pack time)
//   L1, L2   H^T[128 x 32e]  = W[128 x K] · X^T             A (weights) from LDS (resident for the workgroup's lifetime),
//            B = previous accumulators converted in place ("accumulator tile as the next MFMA's operand": the k order
//            inside a step is a fixed permutation, folded into the packed weight columns)
//   L3       rgb^T[32(3) x 32e] = W3 · H2^T, sigmoid, 3 floats written back into the entry's queue slot
// Arithmetic: v_mfma_f32_32x32x16_f16 with every fp32 operand split into fp16 hi + lo and three products per step
// (hi·hi + hi·lo + lo·hi, fp32 accumulate): ~2^-22 relative error per product — fp32-class accuracy at 16/3 the rate of
// the fp32-input MFMA.  b1 rides in W1's image as the column of a constant-1 input (base row 31); b2, b3 are the initial accumulators.
#include "tvr_device.h"
#include "tvr_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef TVR_NOSB
#define TVR_SB
#else
#define TVR_SB __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef TVR_PF
#define TVR_PF 1          // gather prefetch distance in k-steps (ring of TVR_PF + 1 tap sets); 2 spills at 256 VGPRs
#endif
#define TVR_CHK (SRC != SH_SRC_QUEUE)
#ifndef TVR_TIMING
#define TVR_TIMING 0      // diagnostic build: per-phase s_memtime sums into stats[8..14] (scripts/phase_timing.py passes 16 slots)
#endif
#if TVR_TIMING
#define TVR_STAMP(x) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define TVR_STAMP(x)
#endif
#ifndef TVR_PIPE
#define TVR_PIPE 0        // 1: the queue path runs shade_pipe_kernel (one wave per SIMD, two tiles software-pipelined per wave)
#endif
#ifndef TVR_BPF
#define TVR_BPF 0         // 1: issue the basis-fragment loads inside gather k-step TVR_BPF_AT (measured: 14.47 vs 14.38 ms at step 8; step 7 spills)
#endif
#ifndef TVR_BPF_AT
#define TVR_BPF_AT 8
#endif
#ifndef TVR_HWSIN
#define TVR_HWSIN 1       // positional encoding by v_sin_f32 / v_cos_f32 (4 instructions per value instead of ~25 for the polynomial
                          // sincos_fast, which still serves |v| > 256): shade 15.2 -> 14.2 ms, RGB error against the oracle unchanged (scripts/accuracy_report.py)
#endif
#ifndef TVR_QPF
#define TVR_QPF 0         // 1: prefetch the next tile's queue positions one tile ahead (measured: no gain, 15.3 vs 15.0-15.3 ms; +6 VGPRs)
#endif
#ifndef TVR_SHADE_XCD
#define TVR_SHADE_XCD 0
#endif
#ifndef TVR_TOKEN
#define TVR_TOKEN 0       // 1: the two waves of a SIMD (w, w + 4) pass a token and only its holder runs the hidden layers, so one
#endif                    //    wave's gather always sits beside the other's MFMAs.  Measured 15.7 vs 15.5 ms (stagger only): the
                          //    layers run faster alone (8.7k vs 10.1k cycles per tile) but the waits eat it (scripts/phase_timing.py)
#ifndef TVR_STAGGER
#define TVR_STAGGER 1
#endif
#ifndef TVR_SGB
#define TVR_SGB 0         // >0: sched_group_barrier recipe in the hidden-layer k-steps, TVR_SGB VALU ops behind every MFMA
#endif
#ifndef TVR_PRIO
#define TVR_PRIO 0
#endif
#ifndef TVR_COAL
#define TVR_COAL 0        // 1: coalesced gather (4 lanes per 64-B segment + v_permlane16_swap): halves the L1 tag lookups of the
#endif                    //    gather but measured slower (17.8 vs 16.1 ms) — the kernel is not L1-lookup-bound, the swaps cost VALU
// Phase rule (empirical): an earlier version of this kernel that issued global loads between the MFMAs of a tile, with
// two MFMA-issuing waves per SIMD, returned 16 queue entries wrong by ~1e-2, different ones each run; one wave per SIMD was clean.  The
// kernel is therefore phased per tile: a GATHER phase (global loads + VALU, no MFMA) and a MATRIX phase (MFMA + LDS + VALU, no global
// load: the basis fragments are fetched before its first MFMA), and the epilogue's read of the last accumulator drains the wave's
// MFMAs before the next tile's loads.  A stand-alone probe of "global load into the operand registers of just-issued MFMAs" does NOT
// reproduce the corruption (scripts/hwprobe/mfma_war.hip), so the root cause is unconfirmed and the rule is kept as validated practice
// (bit-reproducibility tests); what IS confirmed is that a VALU write directly in front of an MFMA reading it needs software wait
// states (scripts/hwprobe/mfma_raw.hip) — never feed an MFMA from inline asm.  SH_WAVES = 8 gives two waves per SIMD.
#ifndef SH_WAVES
#define SH_WAVES 8
#endif
#define SH_THREADS (64 * SH_WAVES)
#define SH_MINW (SH_WAVES / 4)
#ifndef SH_NCB
#define SH_NCB 1            // column blocks (32 entries each) a wave processes together (2 measured no faster: 20.4 vs 20.1 ms)
#endif
#define SH_TILE (32 * SH_NCB)

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// relu as ONE VALU op: integer max on the bit pattern (negative floats, -0.0 included, are negative integers).  fmaxf(x, 0) costs
// two v_max_f32 (hipcc canonicalises the operand first), and an inline-asm v_max_f32 would bypass the compiler's MFMA-result
// wait states (the accumulator read hazard is software-managed) — that variant rendered non-reproducibly.
__device__ __forceinline__ float relu_f(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// fp32 pair -> packed fp16 hi and lo words (x = hi + lo up to ~2^-22 |x|; round-toward-zero never overflows to inf)
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    // x - hi as fma(hi, -1, x): one v_fma_mix_f32 reading the packed half in place (exact: the product by -1 is exact); hipcc
    // does not select the mixed-precision form by itself (it emits v_cvt_f32_f16 + v_sub), hence the asm.  Its results feed the
    // compiler-visible v_cvt_pkrtz below, never an MFMA directly: VALU-write -> MFMA-read needs software wait states on gfx950
    // (scripts/hwprobe/mfma_raw.hip) and the compiler cannot pad inline asm.  (v_fma_mixlo/mixhi_f16 would fuse the final conversion
    // too — 3 ops per pair — but then asm results ARE the MFMA operands: that variant rendered non-reproducibly, and 15 % fewer VALU
    // ops bought no time anyway: measured.)
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hb), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hb), "v"(b));
    const auto l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
    hi = hb;
    lo = __builtin_bit_cast(unsigned, l);
}

struct Frag {   // one 8-element fp16 operand fragment, hi and lo parts
    uint4 hi, lo;
};

__device__ __forceinline__ Frag split8(const float v[8])
{
    Frag f;
    split2(v[0], v[1], f.hi.x, f.lo.x);
    split2(v[2], v[3], f.hi.y, f.lo.y);
    split2(v[4], v[5], f.hi.z, f.lo.z);
    split2(v[6], v[7], f.hi.w, f.lo.w);
    return f;
}

// One k-step of a single-row-block product (basis, layer 3) for all column blocks: the three hi/lo products are interleaved
// across the column blocks, so no MFMA directly follows an MFMA it depends on.
__device__ __forceinline__ void mfma3cb(const uint4 ah, const uint4 al, const Frag b[SH_NCB], f32x16 acc[SH_NCB])
{
    const h8 Ah = __builtin_bit_cast(h8, ah), Al = __builtin_bit_cast(h8, al);
#pragma unroll
    for (int cb = 0; cb < SH_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, b[cb].hi), acc[cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < SH_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, b[cb].lo), acc[cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < SH_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, b[cb].hi), acc[cb], 0, 0, 0);
}

// A fragments (hi, lo) of the four 32-row blocks of one k-step, from the LDS weight image
struct AFrag4 { uint4 h[4], l[4]; };

__device__ __forceinline__ void load_afrag4(AFrag4 &A, const unsigned char *WH, const unsigned char *WL, int off0, int rb_stride)
{
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        A.h[rb] = *(const uint4 *)(WH + off0 + rb * rb_stride);
        A.l[rb] = *(const uint4 *)(WL + off0 + rb * rb_stride);
    }
}

// one k-step of a 128-row layer: the three hi/lo products interleaved across row and column blocks (no MFMA directly follows
// an MFMA it depends on)
__device__ __forceinline__ void mfma3x4(const AFrag4 &A, const Frag b[SH_NCB], f32x16 acc[SH_NCB][4])
{
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < SH_NCB; ++cb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.l[rb]), __builtin_bit_cast(h8, b[cb].hi), acc[cb][rb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < SH_NCB; ++cb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.h[rb]), __builtin_bit_cast(h8, b[cb].lo), acc[cb][rb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < SH_NCB; ++cb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.h[rb]), __builtin_bit_cast(h8, b[cb].hi), acc[cb][rb], 0, 0, 0);
}

// scheduling recipe for one hidden-layer k-step region: LDS reads of the NEXT step's weights first, then one MFMA followed by
// five VALU ops (the next step's B-fragment conversion), twelve times — keeps one wave's vector and matrix pipes busy together
__device__ __forceinline__ void sched_layer_step()
{
#if TVR_SGB
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 12 * SH_NCB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, TVR_SGB, 0);
    }
#endif
}

// the 6 taps (4 plane texels, 2 line texels) x 8 channels of one entry for one k-step, plus the interpolation weights
struct Taps {
    float4 t[4][2], lv[2][2];
};

template <bool CHECK>
__device__ __forceinline__ void load_taps(Taps &T, const float4 *__restrict__ P, const float4 *__restrict__ Ln, int W, int H, int L,
                                          float fx, float fy, float fl, int q0)
{
    float x0f, y0f, l0f;
    if (CHECK) {
        x0f = floorf(fminf(fmaxf(fx, -2.0f), (float)W + 1.0f));
        y0f = floorf(fminf(fmaxf(fy, -2.0f), (float)H + 1.0f));
        l0f = floorf(fminf(fmaxf(fl, -2.0f), (float)L + 1.0f));
    } else {
        x0f = floorf(fx); y0f = floorf(fy); l0f = floorf(fl);
    }
    const int x0 = (int)x0f, y0 = (int)y0f, l0 = (int)l0f;
    const int Wp = W + 1;
    if (!CHECK) {
        const float4 *p = P + ((size_t)y0 * Wp + x0) * 12 + q0;
        T.t[0][0] = p[0]; T.t[0][1] = p[1];
        T.t[1][0] = p[12]; T.t[1][1] = p[13];
        T.t[2][0] = p[(size_t)Wp * 12]; T.t[2][1] = p[(size_t)Wp * 12 + 1];
        T.t[3][0] = p[(size_t)Wp * 12 + 12]; T.t[3][1] = p[(size_t)Wp * 12 + 13];
        const float4 *q = Ln + (size_t)l0 * 12 + q0;
        T.lv[0][0] = q[0]; T.lv[0][1] = q[1];
        T.lv[1][0] = q[12]; T.lv[1][1] = q[13];
    } else {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool xi[2] = {(x0 >= 0) && (x0 < W), (x0 + 1 >= 0) && (x0 + 1 < W)};
        const bool yi[2] = {(y0 >= 0) && (y0 < H), (y0 + 1 >= 0) && (y0 + 1 < H)};
        const bool li[2] = {(l0 >= 0) && (l0 < L), (l0 + 1 >= 0) && (l0 + 1 < L)};
        const int xc[2] = {min(max(x0, 0), W - 1), min(max(x0 + 1, 0), W - 1)};
        const int yc[2] = {min(max(y0, 0), H - 1), min(max(y0 + 1, 0), H - 1)};
        const int lc[2] = {min(max(l0, 0), L - 1), min(max(l0 + 1, 0), L - 1)};
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const float4 *p = P + ((size_t)yc[ty] * Wp + xc[tx]) * 12 + q0;
                const bool in = xi[tx] && yi[ty];
                T.t[ty * 2 + tx][0] = in ? p[0] : z;
                T.t[ty * 2 + tx][1] = in ? p[1] : z;
            }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const float4 *q = Ln + (size_t)lc[tl] * 12 + q0;
            T.lv[tl][0] = li[tl] ? q[0] : z;
            T.lv[tl][1] = li[tl] ? q[1] : z;
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

// bilinear(plane) * linear(line) for the 8 channels held in T (packed fp32 math: two channels per VALU op)
template <bool CHECK>
__device__ __forceinline__ void taps_eval(const Taps &T, int W, int H, int L, float fx, float fy, float fl, float out[8])
{
    float x0f, y0f, l0f;
    if (CHECK) {
        x0f = floorf(fminf(fmaxf(fx, -2.0f), (float)W + 1.0f));
        y0f = floorf(fminf(fmaxf(fy, -2.0f), (float)H + 1.0f));
        l0f = floorf(fminf(fmaxf(fl, -2.0f), (float)L + 1.0f));
    } else {
        x0f = floorf(fx); y0f = floorf(fy); l0f = floorf(fl);
    }
    const float wx = fx - x0f, wy = fy - y0f, wlf = fl - l0f;
    const float ux = 1.0f - wx, uy = 1.0f - wy, ulf = 1.0f - wlf;
    const float a00 = ux * uy, a01 = wx * uy, a10 = ux * wy, a11 = wx * wy;
    const f32x2 w00 = {a00, a00}, w01 = {a01, a01}, w10 = {a10, a10}, w11 = {a11, a11};
    const f32x2 ul = {ulf, ulf}, wl = {wlf, wlf};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x2 t0 = hh ? f32x2{T.t[0][g].z, T.t[0][g].w} : f32x2{T.t[0][g].x, T.t[0][g].y};
            const f32x2 t1 = hh ? f32x2{T.t[1][g].z, T.t[1][g].w} : f32x2{T.t[1][g].x, T.t[1][g].y};
            const f32x2 t2 = hh ? f32x2{T.t[2][g].z, T.t[2][g].w} : f32x2{T.t[2][g].x, T.t[2][g].y};
            const f32x2 t3 = hh ? f32x2{T.t[3][g].z, T.t[3][g].w} : f32x2{T.t[3][g].x, T.t[3][g].y};
            const f32x2 l0 = hh ? f32x2{T.lv[0][g].z, T.lv[0][g].w} : f32x2{T.lv[0][g].x, T.lv[0][g].y};
            const f32x2 l1 = hh ? f32x2{T.lv[1][g].z, T.lv[1][g].w} : f32x2{T.lv[1][g].x, T.lv[1][g].y};
            f32x2 p = w00 * t0;
            p = pk_fma(w01, t1, p);
            p = pk_fma(w10, t2, p);
            p = pk_fma(w11, t3, p);
            f32x2 q = ul * l0;
            q = pk_fma(wl, l1, q);
            const f32x2 r = p * q;
            out[g * 4 + hh * 2] = r.x;
            out[g * 4 + hh * 2 + 1] = r.y;
        }
    }
}

// ---- coalesced gather (queue path) -------------------------------------------------------------------------------------
// A lane (e, h) needs two float4 pieces (2h, 2h+1) of each tap's 64-B channel segment.  Loading them directly makes every
// wave-level load touch 32 cache lines (two lanes per line).  Instead lane L loads piece (L >> 4) of the segment of entry
// (L & 15) [X] and of entry 16 + (L & 15) [Y]: 4 lanes x 16 B cover one 64-B segment, 16 lines per load (half the L1 tag
// lookups), and ONE v_permlane16_swap per dword (a' = [a0, b0, a2, b2], b' = [a1, b1, a3, b3] by 16-lane rows — probed on
// gfx950, scripts/hwprobe/permlane.hip) hands every lane exactly its two pieces: X' = piece 2h, Y' = piece 2h + 1.
struct TapOff { unsigned p1, p2, l1, l2; };      // float4-unit offsets of the (x0, y0) texel / l0 texel of entry c and entry 16 + c

__device__ __forceinline__ TapOff tap_offsets(int W, float fx, float fy, float fl)
{
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy), l0 = (int)floorf(fl);
    const unsigned op = (unsigned)((y0 * (W + 1) + x0) * 12), ol = (unsigned)(l0 * 12);
    const auto sp = __builtin_amdgcn_permlane16_swap(op, op, false, false);
    const auto sl = __builtin_amdgcn_permlane16_swap(ol, ol, false, false);
    TapOff o;
    o.p1 = sp[0]; o.p2 = sp[1]; o.l1 = sl[0]; o.l2 = sl[1];
    return o;
}

struct TapsXY { float4 x[6], y[6]; };            // taps 0..3 plane (x0y0, x1y0, x0y1, x1y1), 4..5 line; X = entry c, Y = entry 16 + c

__device__ __forceinline__ void load_taps_coal(TapsXY &T, const float4 *__restrict__ P, const float4 *__restrict__ Ln, int W, const TapOff &o,
                                               int seg_row)                       // seg_row = 4 * (s % 3) + (lane >> 4)
{
    const size_t ys = (size_t)(W + 1) * 12;
    const float4 *a1 = P + o.p1 + seg_row, *a2 = P + o.p2 + seg_row;
    const float4 *b1 = Ln + o.l1 + seg_row, *b2 = Ln + o.l2 + seg_row;
    T.x[0] = a1[0]; T.x[1] = a1[12]; T.x[2] = a1[ys]; T.x[3] = a1[ys + 12]; T.x[4] = b1[0]; T.x[5] = b1[12];
    T.y[0] = a2[0]; T.y[1] = a2[12]; T.y[2] = a2[ys]; T.y[3] = a2[ys + 12]; T.y[4] = b2[0]; T.y[5] = b2[12];
}

__device__ __forceinline__ void swap_rows(float4 &x, float4 &y)
{
#define TVR_SWAP1(c)                                                                                                          \
    {                                                                                                                         \
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.c), __float_as_uint(y.c), false, false);             \
        x.c = __uint_as_float(r[0]);                                                                                          \
        y.c = __uint_as_float(r[1]);                                                                                          \
    }
    TVR_SWAP1(x) TVR_SWAP1(y) TVR_SWAP1(z) TVR_SWAP1(w)
#undef TVR_SWAP1
}

// swap the rows into place, then bilinear(plane) * linear(line) for this lane's 8 channels
__device__ __forceinline__ void taps_eval_coal(TapsXY &T, float fx, float fy, float fl, float out[8])
{
#pragma unroll
    for (int i = 0; i < 6; ++i) swap_rows(T.x[i], T.y[i]);
    const float x0f = floorf(fx), y0f = floorf(fy), l0f = floorf(fl);
    const float wx = fx - x0f, wy = fy - y0f, wlf = fl - l0f;
    const float ux = 1.0f - wx, uy = 1.0f - wy, ulf = 1.0f - wlf;
    const float a00 = ux * uy, a01 = wx * uy, a10 = ux * wy, a11 = wx * wy;
    const f32x2 w00 = {a00, a00}, w01 = {a01, a01}, w10 = {a10, a10}, w11 = {a11, a11};
    const f32x2 ul = {ulf, ulf}, wl = {wlf, wlf};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float4 *t = g ? T.y : T.x;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x2 t0 = hh ? f32x2{t[0].z, t[0].w} : f32x2{t[0].x, t[0].y};
            const f32x2 t1 = hh ? f32x2{t[1].z, t[1].w} : f32x2{t[1].x, t[1].y};
            const f32x2 t2 = hh ? f32x2{t[2].z, t[2].w} : f32x2{t[2].x, t[2].y};
            const f32x2 t3 = hh ? f32x2{t[3].z, t[3].w} : f32x2{t[3].x, t[3].y};
            const f32x2 l0 = hh ? f32x2{t[4].z, t[4].w} : f32x2{t[4].x, t[4].y};
            const f32x2 l1 = hh ? f32x2{t[5].z, t[5].w} : f32x2{t[5].x, t[5].y};
            f32x2 p = w00 * t0;
            p = pk_fma(w01, t1, p);
            p = pk_fma(w10, t2, p);
            p = pk_fma(w11, t3, p);
            f32x2 q = ul * l0;
            q = pk_fma(wl, l1, q);
            const f32x2 r = p * q;
            out[g * 4 + hh * 2] = r.x;
            out[g * 4 + hh * 2 + 1] = r.y;
        }
    }
}

// accumulator register r of lane half h  <->  row of the 32x32 tile
__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// sin/cos with a 3-term Cody-Waite reduction and short minimax polynomials (|err| < ~2e-7 for |x| < ~1e4)
__device__ __forceinline__ void sincos_fast(float x, float &s, float &c)
{
    const float k = rintf(x * 0.636619772367581343f);          // x / (pi/2)
    float r = __builtin_fmaf(k, -1.5707962512969971f, x);
    r = __builtin_fmaf(k, -7.5497894158615964e-8f, r);
    r = __builtin_fmaf(k, -5.3903029534742384e-15f, r);
    const float r2 = r * r;
    float sp = __builtin_fmaf(r2, 2.7183114939898219e-6f, -1.9839334836096632e-4f);
    sp = __builtin_fmaf(sp, r2, 8.3333095484102479e-3f);
    sp = __builtin_fmaf(sp, r2, -1.6666665461976921e-1f);
    sp = __builtin_fmaf(sp * r2, r, r);
    float cp = __builtin_fmaf(r2, 2.4433157656e-5f, -1.3887316255e-3f);
    cp = __builtin_fmaf(cp, r2, 4.1666645683e-2f);
    cp = __builtin_fmaf(cp, r2, -0.5f);
    cp = __builtin_fmaf(cp, r2, 1.0f);
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// REF = REFTensoRF (models/REFTensoRF.py:107-133, 174-256): a second basis row block gives normal / diffuse / specular / rho from the
// same h, the view direction is replaced by the reflection about the normalised normal, layer 1 takes one more input (-dot) and the
// colour is  specular_tint * rgb_s + rgb_d.
// hardware sine / cosine (v_sin_f32 / v_cos_f32 take revolutions): 3 instructions per pair instead of ~25
__device__ __forceinline__ void sincos_hw(float x, float &s, float &c)
{
    const float r = __builtin_amdgcn_fractf(x * 0.15915494309189535f);     // keeps the argument inside the instructions' +-256 domain
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}

template <int SRC, int DST, bool REF>
__global__ __launch_bounds__(SH_THREADS, SH_MINW) void shade_kernel(const SceneDev sc, const ShadeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 31, h = lane >> 5;
    if (DST != SH_DST_FEAT || REF) {              // MLP weights -> LDS once per workgroup
        const uint4 *src = (const uint4 *)sc.mlp_image;
        for (int i = tid; i < (REF ? TVR_MLP_IMAGE_BYTES_REF : TVR_MLP_IMAGE_BYTES) / 16; i += SH_THREADS) ((uint4 *)smem)[i] = src[i];
        if (tid < 4) ((volatile unsigned *)(smem + (REF ? TVR_MLP_IMAGE_BYTES_REF : TVR_MLP_IMAGE_BYTES)))[tid] = 0u;
        __syncthreads();
    }
    const long long n_total = (SRC == SH_SRC_QUEUE) ? (long long)(*a.counter) : a.n;
    const long long n_tiles = (n_total + SH_TILE - 1) / SH_TILE;
#if TVR_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // matrix-phase token of this wave's SIMD pair: value = the half (wave >> 2) allowed in, 2 = partner has left (free for all)
    unsigned *const turn = (unsigned *)(smem + (REF ? TVR_MLP_IMAGE_BYTES_REF : TVR_MLP_IMAGE_BYTES)) + (wave & 3);
    const unsigned my_half = (unsigned)(wave >> 2);
    const bool use_token = TVR_TOKEN && SH_WAVES == 8 && DST != SH_DST_FEAT;
#if TVR_STAGGER && !TVR_TOKEN
    // the two waves of a SIMD (w and w + 4) run the same program: start the second half a tile late so that one gathers while
    // the other multiplies (MI355X_MICROARCH 'Two waves per SIMD', item 9)
    if (wave >= 4)
        for (int i = 0; i < TVR_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    // A wave owns SH_NCB column blocks of 32 entries at a time (entry = MFMA column).  The blocks are independent chains, so the
    // VALU work of one overlaps the MFMAs of the other, and every weight fragment read from LDS feeds SH_NCB MFMAs.
#if TVR_SHADE_XCD
    const unsigned lblk = xcd_remap(blockIdx.x, gridDim.x);       // each XCD takes a contiguous eighth of every window of tiles
#else
    const unsigned lblk = blockIdx.x;
#endif
    // the queue position of the NEXT tile is fetched one tile ahead (it heads the dependent chain position -> tap address -> tap)
    const long long tile_stride = (long long)gridDim.x * SH_WAVES;
    float4 qnext[SH_NCB];
#pragma unroll
    for (int cb = 0; cb < SH_NCB; ++cb) {
        qnext[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TVR_QPF && SRC == SH_SRC_QUEUE) {
            const long long en = ((long long)lblk * SH_WAVES + wave) * SH_TILE + cb * 32 + e;
            if (en < n_total) qnext[cb] = a.q_pos[en];
        }
    }
    for (long long tile = (long long)lblk * SH_WAVES + wave; tile < n_tiles; tile += tile_stride) {
        long long ent[SH_NCB];
        bool live[SH_NCB];
        float F[SH_NCB][16];                       // base values: row c = acc_row(r, h) of the feature tile, column = entry
        float dir[SH_NCB][3], wq[SH_NCB];
        float G[SH_NCB][8];                        // REF: rows acc_row(r, h) of the second block (h=0: normal, tint, rgb_d, rho; h=1: normal)
#pragma unroll
        for (int cb = 0; cb < SH_NCB; ++cb) {
#pragma unroll
            for (int r = 0; r < 8; ++r) G[cb][r] = 0.f;
            ent[cb] = tile * SH_TILE + cb * 32 + e;
            live[cb] = ent[cb] < n_total;
            dir[cb][0] = dir[cb][1] = dir[cb][2] = 0.f;
            wq[cb] = 0.f;
        }

#if TVR_TIMING
        unsigned long long tg0 = 0, tg1 = 0, tg2 = 0, tg3 = 0, tg4 = 0, tgA = 0, tgB = 0, tgC = 0;
#endif
        TVR_STAMP(tg0);
        if (SRC != SH_SRC_FEAT) {
            float fc[SH_NCB][3];
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb) {
                float pn[3] = {0.f, 0.f, 0.f};
                if (live[cb]) {
                    if (SRC == SH_SRC_QUEUE) {
                        const float4 q = TVR_QPF ? qnext[cb] : a.q_pos[ent[cb]];
                        pn[0] = q.x; pn[1] = q.y; pn[2] = q.z; wq[cb] = q.w;
                        const float *rp = a.rays + (size_t)a.q_ray[ent[cb]] * 6 + 3;
                        dir[cb][0] = rp[0]; dir[cb][1] = rp[1]; dir[cb][2] = rp[2];
                    } else {
                        pn[0] = a.xyz[ent[cb] * 3]; pn[1] = a.xyz[ent[cb] * 3 + 1]; pn[2] = a.xyz[ent[cb] * 3 + 2];
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) fc[cb][k] = unnorm(pn[k], sc.gm1[k]);
                if (TVR_QPF && SRC == SH_SRC_QUEUE) {
                    const long long en = ent[cb] + tile_stride * SH_TILE;
                    if (en < n_total) qnext[cb] = a.q_pos[en];
                }
            }
            // ---- gather + basis: 9 k-steps of 16 channels (plane p = s/3, channels 16(s%3) + 8h .. +7 of this lane) ----
            // GATHER phase: global loads + VALU only.  The 9 k-steps' B fragments (plane*line products, fp16 hi/lo) stay in registers.
            Frag hf[9][SH_NCB];
            uint4 bah[9], bal[9];
            if (TVR_COAL && SRC == SH_SRC_QUEUE && SH_NCB == 1) {
                TapsXY T[TVR_PF + 1];
                TapOff off[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;      // matMode / vecMode
                    off[p] = tap_offsets(sc.grid[ax], fc[0][ax], fc[0][bx], fc[0][vx]);
                }
                const int row = lane >> 4;
#pragma unroll
                for (int s0 = 0; s0 < TVR_PF; ++s0) {
                    const int p = s0 / 3;
                    load_taps_coal(T[s0], sc.aplane[p], sc.aline[p], sc.grid[(p == 2) ? 1 : 0], off[p], 4 * (s0 % 3) + row);
                }
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    if (s + TVR_PF < 9) {
                        const int s2 = s + TVR_PF, p = s2 / 3;
                        load_taps_coal(T[s2 % (TVR_PF + 1)], sc.aplane[p], sc.aline[p], sc.grid[(p == 2) ? 1 : 0], off[p], 4 * (s2 % 3) + row);
                    }
                    const int p = s / 3;
                    const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;
                    float hv[8];
                    taps_eval_coal(T[s % (TVR_PF + 1)], fc[0][ax], fc[0][bx], fc[0][vx], hv);
                    hf[s][0] = split8(hv);
                    TVR_SB;
                }
            } else {
            Taps T[TVR_PF + 1][SH_NCB];                            // ring: taps of k-steps s .. s+TVR_PF in flight
#pragma unroll
            for (int s0 = 0; s0 < TVR_PF; ++s0) {
                const int p = s0 / 3;
                const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb)
                    load_taps<TVR_CHK>(T[s0][cb], sc.aplane[p], sc.aline[p], sc.grid[ax], sc.grid[bx], sc.grid[vx], fc[cb][ax], fc[cb][bx],
                                       fc[cb][vx], 4 * (s0 % 3) + 2 * h);
            }
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                if (s + TVR_PF < 9) {
                    const int s2 = s + TVR_PF, p = s2 / 3;
                    const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;      // matMode / vecMode
#pragma unroll
                    for (int cb = 0; cb < SH_NCB; ++cb)
                        load_taps<TVR_CHK>(T[s2 % (TVR_PF + 1)][cb], sc.aplane[p], sc.aline[p], sc.grid[ax], sc.grid[bx], sc.grid[vx],
                                           fc[cb][ax], fc[cb][bx], fc[cb][vx], 4 * (s2 % 3) + 2 * h);
                }
                if (TVR_BPF && s == TVR_BPF_AT) {
                    // the basis A fragments (the tile's last global loads) ride behind the last taps instead of stalling the first MFMA
#pragma unroll
                    for (int s3 = 0; s3 < 9; ++s3) {
                        const uint4 *ap = (const uint4 *)sc.basis_frag + ((s3 * 2 + h) * 32 + e) * 2;
                        bah[s3] = ap[0];
                        bal[s3] = ap[1];
                    }
                }
                const int p = s / 3;
                const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) {
                    float hv[8];
                    taps_eval<TVR_CHK>(T[s % (TVR_PF + 1)][cb], sc.grid[ax], sc.grid[bx], sc.grid[vx], fc[cb][ax], fc[cb][bx], fc[cb][vx], hv);
                    hf[s][cb] = split8(hv);
                }
                TVR_SB;
            }
            }
#if TVR_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            TVR_STAMP(tg1);
            // MATRIX phase starts: the basis A fragments are the last global loads of this tile, fetched before its first MFMA
            if (!(TVR_BPF && !(TVR_COAL && SRC == SH_SRC_QUEUE && SH_NCB == 1))) {
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    const uint4 *ap = (const uint4 *)sc.basis_frag + ((s * 2 + h) * 32 + e) * 2;
                    bah[s] = ap[0];
                    bal[s] = ap[1];
                }
            }
            TVR_SB;
            // three independent accumulation chains (hi*lo products) summed at the end: no MFMA directly follows its producer
            f32x16 accA[SH_NCB], accB[SH_NCB], accC[SH_NCB];
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb) { accA[cb] = f32x16{0}; accB[cb] = f32x16{0}; accC[cb] = f32x16{0}; }
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const h8 Ah = __builtin_bit_cast(h8, bah[s]), Al = __builtin_bit_cast(h8, bal[s]);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) accA[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, hf[s][cb].hi), accA[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) accB[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, hf[s][cb].lo), accB[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) accC[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, hf[s][cb].hi), accC[cb], 0, 0, 0);
            }
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) F[cb][r] = (accA[cb][r] + accB[cb][r]) + accC[cb][r];
            if (REF) {
                // second row block: A from the LDS image (8 weight rows; lanes 4..6 re-read the normal rows so that both lane halves
                // hold the normal, every other lane reads the zero row), biases as the initial accumulator
                const int rr = e < 4 ? e : (e < 7 ? e - 4 : ((e >= 8 && e < 12) ? e - 4 : -1));
                const unsigned char *rowp = rr >= 0 ? smem + TVR_IMG_REFW + rr * TVR_IMG_REF_ROW : smem + TVR_IMG_REF_ZROW;
                const float4 g0 = *(const float4 *)(smem + TVR_IMG_REFB + 16 * h), g1 = *(const float4 *)(smem + TVR_IMG_REFB + 32 + 16 * h);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) {
                    accA[cb] = f32x16{0}; accB[cb] = f32x16{0}; accC[cb] = f32x16{0};
                    accA[cb][0] = g0.x; accA[cb][1] = g0.y; accA[cb][2] = g0.z; accA[cb][3] = g0.w;
                    accA[cb][4] = g1.x; accA[cb][5] = g1.y; accA[cb][6] = g1.z; accA[cb][7] = g1.w;
                }
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    const uint4 *ap = (const uint4 *)(rowp + (s * 2 + h) * 32);
                    const h8 Ah = __builtin_bit_cast(h8, ap[0]), Al = __builtin_bit_cast(h8, ap[1]);
#pragma unroll
                    for (int cb = 0; cb < SH_NCB; ++cb) accA[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, hf[s][cb].hi), accA[cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < SH_NCB; ++cb) accB[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, hf[s][cb].lo), accB[cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < SH_NCB; ++cb) accC[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, hf[s][cb].hi), accC[cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 8; ++r) G[cb][r] = (accA[cb][r] + accB[cb][r]) + accC[cb][r];
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = acc_row(r, h);
                    F[cb][r] = (live[cb] && c < TVR_APPDIM) ? a.feats[ent[cb] * TVR_APPDIM + c] : 0.0f;
                }
                if (live[cb]) {
                    dir[cb][0] = a.viewdirs[ent[cb] * 3]; dir[cb][1] = a.viewdirs[ent[cb] * 3 + 1]; dir[cb][2] = a.viewdirs[ent[cb] * 3 + 2];
                }
            }
        }

        if (DST == SH_DST_FEAT) {
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb)
                if (live[cb]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = acc_row(r, h);
                        if (c < TVR_APPDIM) a.out[ent[cb] * TVR_APPDIM + c] = F[cb][r];
                    }
                    if (REF && h == 0 && a.out2) {       // REFTensoRF.compute_appfeature :126-133 {normal, rgb_d, relu(tint), relu(rho)}
                        float *o = a.out2 + ent[cb] * 8;
                        o[0] = G[cb][0]; o[1] = G[cb][1]; o[2] = G[cb][2];
                        o[3] = G[cb][4]; o[4] = G[cb][5]; o[5] = G[cb][6];
                        o[6] = fmaxf(G[cb][3], 0.0f); o[7] = fmaxf(G[cb][7], 0.0f);
                    }
                }
            continue;
        }

        TVR_STAMP(tg2);
        // view direction occupies base rows 27 (h=0, r=15), 28, 29 (h=1, r=12, 13); rows 30, 31 stay zero (zero weights)
        float dotin[SH_NCB];
#pragma unroll
        for (int cb = 0; cb < SH_NCB; ++cb) {
            dotin[cb] = 0.f;
            if (REF) {
                if (SRC != SH_SRC_FEAT) {
                    // REFTensoRF.execute :215-227: normalise the normal, d = -view, dot = d.n, reflection = 2 dot n - d; the MLP
                    // takes the reflection as its direction and -dot as input 0 (row 30: only its t=0 slot has a weight)
                    const float nrm = sqrtf(fmaxf((G[cb][0] * G[cb][0] + G[cb][1] * G[cb][1]) + G[cb][2] * G[cb][2], 1e-30f));
                    const float nx = G[cb][0] / nrm, ny = G[cb][1] / nrm, nz = G[cb][2] / nrm;
                    const float dx = -dir[cb][0], dy = -dir[cb][1], dz = -dir[cb][2];
                    const float dot = (dx * nx + dy * ny) + dz * nz;
                    dir[cb][0] = 2.0f * dot * nx - dx; dir[cb][1] = 2.0f * dot * ny - dy; dir[cb][2] = 2.0f * dot * nz - dz;
                    dotin[cb] = -dot;
                } else if (live[cb]) {
                    dotin[cb] = a.dots[ent[cb]];
                }
            }
            // base row 31's plain slot is the constant 1: its W1 column holds b1, so layer 1 starts from zero accumulators (no bias reads, no moves)
            if (h == 0) F[cb][15] = dir[cb][0];
            else { F[cb][12] = dir[cb][1]; F[cb][13] = dir[cb][2]; F[cb][14] = dotin[cb]; F[cb][15] = 1.0f; }
        }

        // ---- layer 1: 10 k-steps; slot i = 8s + j of this lane is derived value (i % 5) of base value i / 5 ----
        const unsigned char *W1H = smem + TVR_IMG_W1H, *W1L = smem + TVR_IMG_W1L;
        const unsigned char *W2H = smem + TVR_IMG_W2H, *W2L = smem + TVR_IMG_W2L;
        f32x16 acc[SH_NCB][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb) acc[cb][rb] = f32x16{0};
        {
            float S1[SH_NCB][16], C1[SH_NCB][16];
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // hardware sin / cos while |v| is moderate (their error is the fp32 reduction to revolutions, |v| * 6e-8 rad);
                    // a wave in which any lane holds a large value takes the Cody-Waite polynomial for that base row
                    if (TVR_HWSIN && __ballot(fabsf(F[cb][r]) > 256.0f) == 0ull) sincos_hw(F[cb][r], S1[cb][r], C1[cb][r]);
                    else sincos_fast(F[cb][r], S1[cb][r], C1[cb][r]);
                }
            // the hidden layers (240 of the tile's 267 MFMAs) run under the SIMD pair's token; gather, basis product and the
            // positional encoding above are the part that overlaps the partner's turn
            TVR_SB;
            TVR_STAMP(tgA);
            if (use_token) {
                for (;;) {
                    const unsigned t = *(volatile unsigned *)turn;
                    if (t == my_half || t == 2u) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            TVR_STAMP(tgB);
            TVR_SB;
            const int rowoff = e * TVR_IMG_W1_ROW + h * 16;
            auto l1_frag = [&](int s, Frag b[SH_NCB]) {
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * s + j, r = i / 5, t = i % 5;
                        v[j] = t == 0 ? F[cb][r] : (t == 1 ? S1[cb][r] : (t == 2 ? 2.0f * S1[cb][r] * C1[cb][r]                 // sin 2v
                                      : (t == 3 ? C1[cb][r] : __builtin_fmaf(-2.0f * S1[cb][r], S1[cb][r], 1.0f))));          // cos 2v
                    }
                    b[cb] = split8(v);
                }
            };
            Frag bcur[SH_NCB], bnxt[SH_NCB];
            AFrag4 acur, anxt;
            l1_frag(0, bcur);
            load_afrag4(acur, W1H, W1L, rowoff, 32 * TVR_IMG_W1_ROW);
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                if (s + 1 < 10) {
                    load_afrag4(anxt, W1H, W1L, rowoff + (s + 1) * 32, 32 * TVR_IMG_W1_ROW);
                    l1_frag(s + 1, bnxt);
                }
                mfma3x4(acur, bcur, acc);
                if (s + 1 < 10) sched_layer_step();
                acur = anxt;
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) bcur[cb] = bnxt[cb];
                TVR_SB;
            }
        }
        // ---- layer 2: B fragments are the relu'd layer-1 accumulators, 8 registers per k-step ----
        f32x16 acc2[SH_NCB][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4 *)(smem + TVR_IMG_B2 + (32 * rb + 8 * q + 4 * h) * 4);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) {
                    acc2[cb][rb][4 * q] = bv.x; acc2[cb][rb][4 * q + 1] = bv.y; acc2[cb][rb][4 * q + 2] = bv.z; acc2[cb][rb][4 * q + 3] = bv.w;
                }
            }
        {
            const int rowoff = e * TVR_IMG_W2_ROW + h * 16;
            auto relu_frag = [&](f32x16 (&src)[SH_NCB][4], int s, Frag b[SH_NCB]) {
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = relu_f(src[cb][s >> 1][8 * (s & 1) + j]);
                    b[cb] = split8(v);
                }
            };
            Frag bcur[SH_NCB], bnxt[SH_NCB];
            AFrag4 acur, anxt;
            relu_frag(acc, 0, bcur);
            load_afrag4(acur, W2H, W2L, rowoff, 32 * TVR_IMG_W2_ROW);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8) {
                    load_afrag4(anxt, W2H, W2L, rowoff + (s + 1) * 32, 32 * TVR_IMG_W2_ROW);
                    relu_frag(acc, s + 1, bnxt);
                }
                mfma3x4(acur, bcur, acc2);
                if (s + 1 < 8) sched_layer_step();
                acur = anxt;
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) bcur[cb] = bnxt[cb];
                TVR_SB;
            }
        }
        TVR_STAMP(tg3);
        // ---- layer 3: rows 0..2 of W3 (+ a shared zero row) from LDS, bias b3 as the initial accumulator ----
        // three independent accumulation chains (as in the basis product): with one column block the hi/lo products of a k-step
        // would otherwise be 24 MFMAs each waiting for its predecessor
        f32x16 acc3[SH_NCB], acc3b[SH_NCB], acc3c[SH_NCB];
        {
            const float b30 = sc.b3[0], b31 = sc.b3[1], b32 = sc.b3[2];
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb) {
                acc3[cb] = f32x16{0}; acc3b[cb] = f32x16{0}; acc3c[cb] = f32x16{0};
                acc3[cb][0] = h == 0 ? b30 : 0.0f; acc3[cb][1] = h == 0 ? b31 : 0.0f; acc3[cb][2] = h == 0 ? b32 : 0.0f;
            }
        }
        {
            Frag bcur[SH_NCB], bnxt[SH_NCB];
            auto relu2_frag = [&](int s, Frag b[SH_NCB]) {
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = relu_f(acc2[cb][s >> 1][8 * (s & 1) + j]);
                    b[cb] = split8(v);
                }
            };
            relu2_frag(0, bcur);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                // rows >= 3 of W3 are zero: every such lane reads the shared zero row (address select)
                const uint4 *ap = (const uint4 *)(smem + TVR_IMG_W3 + (e < 3 ? e : 3) * TVR_IMG_W3_ROW + (s * 2 + h) * 32);
                const h8 Ah = __builtin_bit_cast(h8, ap[0]), Al = __builtin_bit_cast(h8, ap[1]);
                if (s + 1 < 8) relu2_frag(s + 1, bnxt);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) acc3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, bcur[cb].hi), acc3[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) acc3b[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, bcur[cb].lo), acc3b[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) acc3c[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, bcur[cb].hi), acc3c[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < SH_NCB; ++cb) bcur[cb] = bnxt[cb];
                TVR_SB;
            }
            // the last MFMA of this tile is issued: hand the token over (unless the partner has left); the epilogue is VALU only
            TVR_STAMP(tgC);
            if (use_token && lane == 0) atomicCAS(turn, my_half, 1u - my_half);
            TVR_SB;
#pragma unroll
            for (int cb = 0; cb < SH_NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 3; ++r) acc3[cb][r] = (acc3[cb][r] + acc3b[cb][r]) + acc3c[cb][r];
        }
#pragma unroll
        for (int cb = 0; cb < SH_NCB; ++cb) {
            // every lane reads the last accumulator (drains this wave's MFMAs before the next tile's global loads)
            float r0 = sigmoid_f(acc3[cb][0]), r1 = sigmoid_f(acc3[cb][1]), r2 = sigmoid_f(acc3[cb][2]);
            if (REF && SRC != SH_SRC_FEAT) {           // :232  specular_tint * clamp(rgb_s, 0) + rgb_d
                const float tint = fmaxf(G[cb][3], 0.0f);
                r0 = tint * fmaxf(r0, 0.0f) + G[cb][4]; r1 = tint * fmaxf(r1, 0.0f) + G[cb][5]; r2 = tint * fmaxf(r2, 0.0f) + G[cb][6];
            }
            if (live[cb] && h == 0) {
                if (DST == SH_DST_QUEUE) {
                    a.q_out[ent[cb]] = make_float4(r0, r1, r2, wq[cb]);
                } else {
                    a.out[ent[cb] * 3] = r0; a.out[ent[cb] * 3 + 1] = r1; a.out[ent[cb] * 3 + 2] = r2;
                }
            }
        }
        TVR_SB;
#if TVR_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#if TVR_TIMING
        TVR_STAMP(tg4);
        tsum[0] += tg1 - tg0; tsum[1] += tg2 - tg1; tsum[2] += tgA - tg2; tsum[3] += tgB - tgA; tsum[4] += tg3 - tgB; tsum[5] += tgC - tg3; tsum[6] += tg4 - tgC;
#endif
    }
    if (use_token && lane == 0) atomicExch(turn, 2u);                             // no more tiles here: the partner runs freely
#if TVR_TIMING
    if (a.stats && lane == 0)
        for (int i = 0; i < 7; ++i) atomicAdd((unsigned long long *)&a.stats[8 + i], tsum[i]);     // diagnostic build: 16-slot stats
#endif
    if (a.stats && SRC == SH_SRC_QUEUE && blockIdx.x == 0 && tid == 0)
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_APP], (unsigned long long)n_total);
}

// ---- software-pipelined variant (build with -DTVR_PIPE=1): ONE wave per SIMD, two tiles in flight per wave -------------------------
// The two-waves-per-SIMD kernel above runs each wave's gather and matrix phases back to back (~25 k cycles per tile and wave), so a
// SIMD needs ~13 k cycles per tile however its two waves interleave.  Here a wave computes the MATRIX part of tile i while it issues
// the loads and the interpolation arithmetic of tile i+1's GATHER between the MFMAs of the hidden layers (one gather k-step per two
// layer k-steps), with the queue entry and the basis fragments of the following stage prefetched.  With a single MFMA-issuing wave per
// SIMD the load-behind-MFMA hazard cannot occur.  Parity-green (tests/test_gpu_parity.py with this build), measured 15.0 ms against
// 14.2 ms for the kernel above: phase stamps give 16.8 k cycles per tile — basis 1.2 k, entry + PE 0.8 k, L1 6.1 k and L2 5.0 k (6.9 k of
// MFMA pipe between them, stalled on the tap loads issued two layer steps earlier), L3 2.6 k, epilogue 1.1 k — and nothing fills the
// MFMA-idle stages.  A deeper tap ring or earlier basis prefetch spills: a 512-register wave still has only 256 registers the VALU can
// address (the other 256 are accumulator registers), and ring + fragments + weights already need ~330.  Kept as a build switch.
#if TVR_PIPE
#define PW_WAVES 4
#ifndef PW_RING
#define PW_RING 2
#endif
#ifndef PW_SGB
#define PW_SGB 0          // >0: sched_group_barrier recipe, PW_SGB VALU operations behind every hidden-layer MFMA
#endif
#ifndef PW_SGX
#define PW_SGX 3          // extra VALU operations per MFMA in the k-steps that carry a gather step
#endif
// order inside one hidden-layer k-step: the next step's 8 weight reads (and a gather step's 12 tap loads) first, then each of the 12
// MFMAs followed by NV VALU operations
template <int NV, bool LOADS>
__device__ __forceinline__ void sched_pipe_step()
{
#if PW_SGB
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    if (LOADS) __builtin_amdgcn_sched_group_barrier(0x020, 12, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
#endif
}

struct PipeTile {
    long long ent;
    bool live;
    float fc[3], dir[3], wq;
    Frag hf[9];
};

__global__ __launch_bounds__(64 * PW_WAVES, 1) void shade_pipe_kernel(const SceneDev sc, const ShadeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 31, h = lane >> 5;
    {
        const uint4 *src = (const uint4 *)sc.mlp_image;
        for (int i = tid; i < TVR_MLP_IMAGE_BYTES / 16; i += 64 * PW_WAVES) ((uint4 *)smem)[i] = src[i];
        __syncthreads();
    }
    const long long n_total = (long long)(*a.counter);
    const long long n_tiles = (n_total + 31) / 32;
    const long long stride = (long long)gridDim.x * PW_WAVES;
    const long long first = (long long)blockIdx.x * PW_WAVES + wave;
    if (first >= n_tiles) return;
    const unsigned char *W1H = smem + TVR_IMG_W1H, *W1L = smem + TVR_IMG_W1L;
    const unsigned char *W2H = smem + TVR_IMG_W2H, *W2L = smem + TVR_IMG_W2L;

    Taps T[PW_RING];                                     // tap ring: gather steps k .. k + PW_RING - 1 in flight
    auto plane_args = [&](int s, int &p, int &ax, int &bx, int &vx) { p = s / 3; ax = (p == 2) ? 1 : 0; bx = (p == 0) ? 1 : 2; vx = 2 - p; };
    auto issue_taps = [&](PipeTile &g, int s) {
        int p, ax, bx, vx;
        plane_args(s, p, ax, bx, vx);
        load_taps<false>(T[s % PW_RING], sc.aplane[p], sc.aline[p], sc.grid[ax], sc.grid[bx], sc.grid[vx], g.fc[ax], g.fc[bx], g.fc[vx], 4 * (s % 3) + 2 * h);
    };
    // the queue entry of a tile is fetched a whole stage ahead (it heads the dependent chain entry -> coordinates -> tap address -> tap)
    long long qn_ent = 0;
    bool qn_live = false;
    float4 qn_q = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned qn_ray = 0;
    auto fetch_entry = [&](long long tile) {
        long long en = tile * 32 + e;
        qn_live = (tile < n_tiles) && (en < n_total);
        if (en >= n_total) en = n_total - 1;                // a dead entry gathers a valid address; nothing of it is stored
        qn_ent = en;
        qn_q = a.q_pos[en];
        qn_ray = a.q_ray[en];
    };
    // gather step 0': prefetched entry -> coordinates, view direction and the first tap sets in flight
    auto gather_begin = [&](PipeTile &g) {
        g.ent = qn_ent; g.live = qn_live; g.wq = qn_q.w;
        const float *rp = a.rays + (size_t)qn_ray * 6 + 3;
        g.dir[0] = rp[0]; g.dir[1] = rp[1]; g.dir[2] = rp[2];
        g.fc[0] = unnorm(qn_q.x, sc.gm1[0]); g.fc[1] = unnorm(qn_q.y, sc.gm1[1]); g.fc[2] = unnorm(qn_q.z, sc.gm1[2]);
#pragma unroll
        for (int s = 0; s < PW_RING - 1; ++s) issue_taps(g, s);
    };
    // the basis A fragments (the same 18 x 16 B for every tile, but 72 VGPRs the hidden layers cannot spare) are re-read during
    // layer 3 of the previous stage, a few thousand cycles before the basis product needs them
    uint4 bh[9], bl[9];
    auto fetch_basis = [&]() {
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const uint4 *ap = (const uint4 *)sc.basis_frag + ((s * 2 + h) * 32 + e) * 2;
            bh[s] = ap[0]; bl[s] = ap[1];
        }
    };
    auto gather_step = [&](PipeTile &g, int s) {            // s = 0..8: issue step s + PW_RING - 1, evaluate step s
        if (s + PW_RING - 1 < 9) issue_taps(g, s + PW_RING - 1);
        int p, ax, bx, vx;
        plane_args(s, p, ax, bx, vx);
        float hv[8];
        taps_eval<false>(T[s % PW_RING], sc.grid[ax], sc.grid[bx], sc.grid[vx], g.fc[ax], g.fc[bx], g.fc[vx], hv);
        g.hf[s] = split8(hv);
    };
    // matrix part of tile M with the gather of tile N (for `next_tile`) threaded through the hidden layers
#if TVR_TIMING
    unsigned long long psum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto stage = [&](PipeTile &M, PipeTile &N, long long following_tile) {
#if TVR_TIMING
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0;
#endif
        TVR_SB; TVR_STAMP(p0);
        float F[16];
        {   // basis: A fragments prefetched by the previous stage
            f32x16 accA = f32x16{0}, accB = f32x16{0}, accC = f32x16{0};
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const h8 Ah = __builtin_bit_cast(h8, bh[s]), Al = __builtin_bit_cast(h8, bl[s]);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, M.hf[s].hi), accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, M.hf[s].lo), accB, 0, 0, 0);
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, M.hf[s].hi), accC, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) F[r] = (accA[r] + accB[r]) + accC[r];
        }
        TVR_SB; TVR_STAMP(p1);
        gather_begin(N);
        if (h == 0) F[15] = M.dir[0];
        else { F[12] = M.dir[1]; F[13] = M.dir[2]; F[14] = 0.f; F[15] = 1.0f; }      // row 31 = 1: W1's bias column
        f32x16 acc[1][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[0][rb] = f32x16{0};
        {
            const int rowoff = e * TVR_IMG_W1_ROW + h * 16;
            // sin / cos are re-derived where a k-step needs them (2-3 base values per step, 3 instructions each) instead of being
            // held in 32 registers across the layer
            auto l1_frag = [&](int s, Frag b[1]) {
                float v[8];
                float S1[3], C1[3];
                const int r0 = (8 * s) / 5;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (r0 + q < 16) sincos_hw(F[r0 + q], S1[q], C1[q]);
                    else { S1[q] = 0.f; C1[q] = 1.f; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * s + j, r = i / 5, t = i % 5, q = r - r0;
                    v[j] = t == 0 ? F[r] : (t == 1 ? S1[q] : (t == 2 ? 2.0f * S1[q] * C1[q] : (t == 3 ? C1[q] : __builtin_fmaf(-2.0f * S1[q], S1[q], 1.0f))));
                }
                b[0] = split8(v);
            };
            Frag bcur[1], bnxt[1];
            AFrag4 acur, anxt;
            l1_frag(0, bcur);
            load_afrag4(acur, W1H, W1L, rowoff, 32 * TVR_IMG_W1_ROW);
            TVR_SB; TVR_STAMP(p2);
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                if (s + 1 < 10) {
                    load_afrag4(anxt, W1H, W1L, rowoff + (s + 1) * 32, 32 * TVR_IMG_W1_ROW);
                    l1_frag(s + 1, bnxt);
                }
                mfma3x4(acur, bcur, acc);
                if ((s & 1) == 0) gather_step(N, s >> 1);          // gather steps 0..4
                if (s + 1 < 10) { if ((s & 1) == 0) sched_pipe_step<PW_SGB + PW_SGX, true>(); else sched_pipe_step<PW_SGB, false>(); }
                acur = anxt;
                bcur[0] = bnxt[0];
                TVR_SB;
            }
        }
        TVR_SB; TVR_STAMP(p3);
        f32x16 acc2[1][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4 *)(smem + TVR_IMG_B2 + (32 * rb + 8 * q + 4 * h) * 4);
                acc2[0][rb][4 * q] = bv.x; acc2[0][rb][4 * q + 1] = bv.y; acc2[0][rb][4 * q + 2] = bv.z; acc2[0][rb][4 * q + 3] = bv.w;
            }
        {
            const int rowoff = e * TVR_IMG_W2_ROW + h * 16;
            auto relu_frag = [&](int s, Frag b[1]) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu_f(acc[0][s >> 1][8 * (s & 1) + j]);
                b[0] = split8(v);
            };
            Frag bcur[1], bnxt[1];
            AFrag4 acur, anxt;
            relu_frag(0, bcur);
            load_afrag4(acur, W2H, W2L, rowoff, 32 * TVR_IMG_W2_ROW);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8) {
                    load_afrag4(anxt, W2H, W2L, rowoff + (s + 1) * 32, 32 * TVR_IMG_W2_ROW);
                    relu_frag(s + 1, bnxt);
                }
                mfma3x4(acur, bcur, acc2);
                if ((s & 1) == 0) gather_step(N, 5 + (s >> 1));   // gather steps 5..8
                if (s + 1 < 8) { if ((s & 1) == 0) sched_pipe_step<PW_SGB + PW_SGX, true>(); else sched_pipe_step<PW_SGB, false>(); }
                acur = anxt;
                bcur[0] = bnxt[0];
                TVR_SB;
            }
        }
        TVR_SB; TVR_STAMP(p4);
        fetch_entry(following_tile);                            // the tile after N
        f32x16 acc3 = f32x16{0}, acc3b = f32x16{0}, acc3c = f32x16{0};
        {
            const float b30 = sc.b3[0], b31 = sc.b3[1], b32 = sc.b3[2];
            acc3[0] = h == 0 ? b30 : 0.0f; acc3[1] = h == 0 ? b31 : 0.0f; acc3[2] = h == 0 ? b32 : 0.0f;
            Frag bcur, bnxt;
            auto relu2_frag = [&](int s, Frag &b) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu_f(acc2[0][s >> 1][8 * (s & 1) + j]);
                b = split8(v);
            };
            relu2_frag(0, bcur);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint4 *ap = (const uint4 *)(smem + TVR_IMG_W3 + (e < 3 ? e : 3) * TVR_IMG_W3_ROW + (s * 2 + h) * 32);
                const h8 Ah = __builtin_bit_cast(h8, ap[0]), Al = __builtin_bit_cast(h8, ap[1]);
                if (s + 1 < 8) relu2_frag(s + 1, bnxt);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, bcur.hi), acc3, 0, 0, 0);
                acc3b = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, bcur.lo), acc3b, 0, 0, 0);
                acc3c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, bcur.hi), acc3c, 0, 0, 0);
                bcur = bnxt;
                TVR_SB;
            }
        }
        TVR_SB; TVR_STAMP(p5);
        const float r0 = sigmoid_f((acc3[0] + acc3b[0]) + acc3c[0]), r1 = sigmoid_f((acc3[1] + acc3b[1]) + acc3c[1]),
                    r2 = sigmoid_f((acc3[2] + acc3b[2]) + acc3c[2]);
        if (M.live && h == 0) a.q_out[M.ent] = make_float4(r0, r1, r2, M.wq);
        fetch_basis();                                          // for the next stage (issued any earlier, the 72 registers spill)
        TVR_SB; TVR_STAMP(p6);
#if TVR_TIMING
        psum[0] += p1 - p0; psum[1] += p2 - p1; psum[2] += p3 - p2; psum[3] += p4 - p3; psum[4] += p5 - p4; psum[5] += p6 - p5;
#endif
    };

    PipeTile ga, gb;
    fetch_entry(first);
    gather_begin(ga);
#pragma unroll
    for (int s = 0; s < 9; ++s) gather_step(ga, s);           // prologue: the first tile's gather runs alone
    fetch_basis();
    fetch_entry(first + stride);
    for (long long tile = first; tile < n_tiles; tile += 2 * stride) {
        stage(ga, gb, tile + 2 * stride);                     // matrix(tile), gather(tile + stride), entry prefetch(tile + 2 stride)
        if (tile + stride < n_tiles) stage(gb, ga, tile + 3 * stride);
    }
#if TVR_TIMING
    if (a.stats && lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long *)&a.stats[8 + i], psum[i]);     // basis, gather begin + PE, L1, L2, L3, epilogue
#endif
    if (a.stats && blockIdx.x == 0 && tid == 0) atomicAdd((unsigned long long *)&a.stats[TVR_STAT_APP], (unsigned long long)n_total);
}

#endif  // TVR_PIPE

template <int SRC, int DST, bool REF>
static hipError_t launch_shade_t(const SceneDev &sc, const ShadeArgs &a, hipStream_t stream)
{
    const int lds = (REF ? TVR_MLP_IMAGE_BYTES_REF : ((DST == SH_DST_FEAT) ? 0 : TVR_MLP_IMAGE_BYTES)) + 16;    // + the token words
    hipError_t rc = hipFuncSetAttribute((const void *)shade_kernel<SRC, DST, REF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (rc != hipSuccess) return rc;
    unsigned grid = 256;       // one workgroup per CU (LDS holds the MLP weights), persistent over SH_TILE-entry tiles
    if (SRC != SH_SRC_QUEUE) {
        const long long groups = (a.n + SH_TILE * SH_WAVES - 1) / (SH_TILE * SH_WAVES);
        if (groups < grid) grid = (unsigned)(groups > 0 ? groups : 1);
    }
    hipLaunchKernelGGL((shade_kernel<SRC, DST, REF>), dim3(grid), dim3(SH_THREADS), lds, stream, sc, a);
    return hipGetLastError();
}

hipError_t launch_shade(const SceneDev &sc, int src, int dst, const ShadeArgs &a, hipStream_t stream)
{
    if (sc.variant == 1) {
        if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE) return launch_shade_t<SH_SRC_QUEUE, SH_DST_QUEUE, true>(sc, a, stream);
        if (src == SH_SRC_XYZ && dst == SH_DST_FEAT) return launch_shade_t<SH_SRC_XYZ, SH_DST_FEAT, true>(sc, a, stream);
        if (src == SH_SRC_FEAT && dst == SH_DST_RGB) return launch_shade_t<SH_SRC_FEAT, SH_DST_RGB, true>(sc, a, stream);
        return hipErrorInvalidValue;
    }
#if TVR_PIPE
    if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE) {
        hipError_t rc = hipFuncSetAttribute((const void *)shade_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TVR_MLP_IMAGE_BYTES);
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(shade_pipe_kernel, dim3(256), dim3(64 * PW_WAVES), TVR_MLP_IMAGE_BYTES, stream, sc, a);
        return hipGetLastError();
    }
#endif
    if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE) return launch_shade_t<SH_SRC_QUEUE, SH_DST_QUEUE, false>(sc, a, stream);
    if (src == SH_SRC_XYZ && dst == SH_DST_FEAT) return launch_shade_t<SH_SRC_XYZ, SH_DST_FEAT, false>(sc, a, stream);
    if (src == SH_SRC_FEAT && dst == SH_DST_RGB) return launch_shade_t<SH_SRC_FEAT, SH_DST_RGB, false>(sc, a, stream);
    return hipErrorInvalidValue;
}

// ---- scene packing (reference layout -> channels-last, zero-padded) ----
// in (C,H,W) -> out [H+1][W+1][C]; a line (W == 1) packs to [H+1][C]
__global__ __launch_bounds__(256) void pack_plane_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int H, int W, int Wp)
{
    const long long total = (long long)(H + 1) * Wp * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long t = i / C;
        const int x = (int)(t % Wp), y = (int)(t / Wp);
        out[i] = (x < W && y < H) ? in[((size_t)c * H + y) * W + x] : 0.0f;
    }
}

hipError_t launch_pack_plane(const float *in, float *out, int C, int H, int W, hipStream_t stream)
{
    const int Wp = (W == 1) ? 1 : W + 1;
    const long long total = (long long)(H + 1) * Wp * C;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_plane_kernel, dim3(grid), dim3(256), 0, stream, in, out, C, H, W, Wp);
    return hipGetLastError();
}

// reference input index (tensorBase.py:77-82 concat order) of derived value t of base value c; -1 = zero weight
__device__ __forceinline__ int ref_in_index(int c, int t)
{
    if (c < TVR_APPDIM) return t == 0 ? c : (t == 1 ? 30 + 2 * c : (t == 2 ? 31 + 2 * c : (t == 3 ? 84 + 2 * c : 85 + 2 * c)));
    if (c < TVR_APPDIM + 3) {
        const int d = c - TVR_APPDIM;
        return t == 0 ? 27 + d : (t == 1 ? 138 + 2 * d : (t == 2 ? 139 + 2 * d : (t == 3 ? 144 + 2 * d : 145 + 2 * d)));
    }
    return -1;
}

// MLP weights -> fp16 hi/lo operand images.  One thread per (row, k position).
//  mode 0: W1 LDS image  [128][W1_ROW/2 halfs]: kpos = 16s + 8h + j  <->  derived (i%5) of base acc_row(i/5, h), i = 8s + j;
//          base row 31's plain slot (the constant-1 input) carries b1
//  mode 1: W2 LDS image  [128][W2_ROW/2 halfs]: kpos = 16s + 8h + j  <->  hidden unit 16s + 8(j>>2) + 4h + (j&3)
//  mode 2: basis fragments [9][2][32][hi 8 | lo 8]: row r < 27, k = 16s + 8h + j (natural)
//  mode 3: W3 LDS block    [4][8][2][hi 8 | lo 8]: rows 0..2 of W3 + one zero row, k as mode 1
//  mode 4: mode 0 for MLPRender_Fea_Ref (REFTensoRF.py:19-24: [dot, features, viewdirs, PE(features), PE(viewdirs)], 151 inputs):
//          every index moves up by one and base row 30's plain slot carries input 0 (dot)
__global__ __launch_bounds__(256) void pack_mlp_kernel(const float *__restrict__ W, const float *__restrict__ bias,
                                                       unsigned short *__restrict__ out_hi, unsigned short *__restrict__ out_lo, int mode)
{
    const int nrows = (mode <= 1 || mode == 4) ? 128 : (mode == 2 ? 32 : 4);
    const int K = (mode == 0 || mode == 4) ? 160 : (mode == 2 ? 144 : 128);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * K) return;
    const int row = i / K, kpos = i - row * K;
    const int s = kpos >> 4, hh = (kpos >> 3) & 1, j = kpos & 7;
    float w = 0.0f;
    if (mode == 0) {
        const int ii = 8 * s + j;
        const int c = acc_row(ii / 5, hh), idx = ref_in_index(c, ii % 5);
        if (idx >= 0) w = W[(size_t)row * TVR_NIN + idx];
        if (c == 31 && ii % 5 == 0) w = bias[row];                 // the constant-1 input (base row 31): b1 rides in the weight image
    } else if (mode == 4) {
        const int ii = 8 * s + j, c = acc_row(ii / 5, hh), t = ii % 5;
        const int idx = (c == TVR_APPDIM + 3) ? (t == 0 ? 0 : -1) : (ref_in_index(c, t) >= 0 ? ref_in_index(c, t) + 1 : -1);
        if (idx >= 0) w = W[(size_t)row * TVR_NIN_REF + idx];
        if (c == 31 && t == 0) w = bias[row];
    } else if (mode == 1) {
        w = W[(size_t)row * TVR_FEATC + (16 * s + 8 * (j >> 2) + 4 * hh + (j & 3))];
    } else if (mode == 2) {
        if (row < TVR_APPDIM) w = W[(size_t)row * TVR_KAPP + kpos];
    } else {
        if (row < 3) w = W[(size_t)row * TVR_FEATC + (16 * s + 8 * (j >> 2) + 4 * hh + (j & 3))];
    }
    unsigned hi, lo;
    split2(w, 0.0f, hi, lo);
    if (mode == 0 || mode == 4) {
        out_hi[row * (TVR_IMG_W1_ROW / 2) + kpos] = (unsigned short)hi;
        out_lo[row * (TVR_IMG_W1_ROW / 2) + kpos] = (unsigned short)lo;
    } else if (mode == 1) {
        out_hi[row * (TVR_IMG_W2_ROW / 2) + kpos] = (unsigned short)hi;
        out_lo[row * (TVR_IMG_W2_ROW / 2) + kpos] = (unsigned short)lo;
    } else if (mode == 2) {
        unsigned short *o = out_hi + ((size_t)((s * 2 + hh) * 32 + row)) * 16;   // [hi 8 | lo 8] per (s, h, row)
        o[j] = (unsigned short)hi;
        o[8 + j] = (unsigned short)lo;
    } else {
        unsigned short *o = out_hi + ((size_t)((row * 8 + s) * 2 + hh)) * 16;    // [hi 8 | lo 8] per (row, s, h)
        o[j] = (unsigned short)hi;
        o[8 + j] = (unsigned short)lo;
    }
}

hipError_t launch_pack_mlp(const float *W, const float *bias, void *out_hi, void *out_lo, int mode, hipStream_t stream)
{
    const int n = ((mode <= 1 || mode == 4) ? 128 : (mode == 2 ? 32 : 4)) * ((mode == 0 || mode == 4) ? 160 : (mode == 2 ? 144 : 128));
    hipLaunchKernelGGL(pack_mlp_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, W, bias, (unsigned short *)out_hi, (unsigned short *)out_lo, mode);
    return hipGetLastError();
}

// REFTensoRF's four 144 -> {3,3,1,1} linears -> the 8 LDS rows (normal 0..2, specular 3, diffuse 4..6, rho 7; k natural as mode 2)
// and the 16 biases in accumulator-row order {n, tint | n, 0 | rgb_d, rho | 0}
struct RefPtrs { const float *W[4], *b[4]; };      // normal, diffuse, specular, rho
__global__ __launch_bounds__(256) void pack_ref_kernel(const RefPtrs p, unsigned short *__restrict__ rows, float *__restrict__ bias)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16) {
        float b = 0.0f;
        if (i < 3) b = p.b[0][i];
        else if (i == 3) b = p.b[2][0];
        else if (i < 7) b = p.b[0][i - 4];
        else if (i >= 8 && i < 11) b = p.b[1][i - 8];
        else if (i == 11) b = p.b[3][0];
        bias[i] = b;
    }
    if (i >= 8 * TVR_KAPP) return;
    const int row = i / TVR_KAPP, kpos = i - row * TVR_KAPP;
    const int s = kpos >> 4, hh = (kpos >> 3) & 1, j = kpos & 7;
    const float *src = row < 3 ? p.W[0] + row * TVR_KAPP : (row == 3 ? p.W[2] : (row < 7 ? p.W[1] + (row - 4) * TVR_KAPP : p.W[3]));
    unsigned hi, lo;
    split2(src[kpos], 0.0f, hi, lo);
    unsigned short *o = rows + (size_t)row * (TVR_IMG_REF_ROW / 2) + (s * 2 + hh) * 16;
    o[j] = (unsigned short)hi;
    o[8 + j] = (unsigned short)lo;
}

hipError_t launch_pack_ref(const float *const W[4], const float *const b[4], void *rows, float *bias, hipStream_t stream)
{
    RefPtrs p;
    for (int i = 0; i < 4; ++i) { p.W[i] = W[i]; p.b[i] = b[i]; }
    hipLaunchKernelGGL(pack_ref_kernel, dim3((8 * TVR_KAPP + 255) / 256), dim3(256), 0, stream, p, (unsigned short *)rows, bias);
    return hipGetLastError();
}
