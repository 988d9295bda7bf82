// tvr_shade_common.h — what the two shade translation units share (tvr_shade.hip: 32x32x16 tiles, every mode; tvr_shade16.hip: the render path on 16x16x32 tiles):
// the appearance gather (tap loads + bilinear x linear interpolation, op for op the order the reference's grid_sample products have), the pinned fp32 forms,
// sin / cos of the positional encoding, sigmoid, the fp16-range helper.  Moved out of tvr_shade.hip unchanged (round 5).
#pragma once
#include "tvr_device.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef TVR_FAST_SIGMOID
#define TVR_FAST_SIGMOID 0    // experiment (round 5): tvr_shade.hip's kernels with tvr_shade16.hip's v_exp_f32 / v_rcp_f32 sigmoid
#endif
#ifndef TVR_FAST_SINCOS
#define TVR_FAST_SINCOS 0     // experiment (round 5): ... and its fract-based sin / cos argument
#endif
#if TVR_FAST_SIGMOID
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
#else
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif
// (the packed forms v_pk_fma_f32 / v_pk_mul_f32: half the instructions and 2 % SLOWER — beside the partner wave's MFMA stream a packed fp32 op takes
// 52.7 cycles instead of 4.6, scripts/hwprobe/valu_rate.hip)
// two plain v_fma_f32 / v_mul_f32, each pinned by an empty asm (without the pins, and with the SLP vectoriser off: 12.92 vs 12.75 ms)
#ifndef TVR_PIN_PK
#define TVR_PIN_PK 2      // 2: every fp32 op of the interpolation / layer 3 is pinned by an empty asm (keeps the SLP vectoriser from pairing them, and the ops where they
#endif                    // are written); 1: one pin per interpolated channel pair (build with -fno-slp-vectorize); 0: none.  Each pin costs an s_nop 0 — hipcc guards
                          // an inline asm that reads a just-written VGPR — 175 per tile at level 2; level 0 lets hipcc hoist the loads' consumers apart: 256 VGPRs + spills.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c)
{
    float x = __builtin_fmaf(a.x, b.x, c.x), y = __builtin_fmaf(a.y, b.y, c.y);
#if TVR_PIN_PK >= 1
    asm volatile("" : "+v"(x)); asm volatile("" : "+v"(y));
#endif
    return f32x2{x, y};
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b)
{
    float x = a.x * b.x, y = a.y * b.y;
#if TVR_PIN_PK >= 1
    asm volatile("" : "+v"(x)); asm volatile("" : "+v"(y));
#endif
    return f32x2{x, y};
}

// the interpolation's own forms: pinned per op at TVR_PIN_PK 2, per channel pair at 1
__device__ __forceinline__ f32x2 ip_fma(f32x2 a, f32x2 b, f32x2 c)
{
#if TVR_PIN_PK >= 2
    return pk_fma(a, b, c);
#else
    return f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
#endif
}
__device__ __forceinline__ f32x2 ip_mul(f32x2 a, f32x2 b)
{
#if TVR_PIN_PK >= 2
    return pk_mul(a, b);
#else
    return f32x2{a.x * b.x, a.y * b.y};
#endif
}
// the 6 taps (4 plane texels, 2 line texels) x 8 channels of one entry for one k-step
struct Taps {
    float4 t[4][2], lv[2][2];
};

// first float4 of the two this lane fetches from a texel (12 float4 = 48 channels) in k-step t of a plane.  (Measured alternative: lane half 0
// takes float4s 0..5 and half 1 float4s 6..11 of the texel, so that a load instruction touches every 64-B segment once instead of from both
// halves of the wave — 12.77 vs 12.70 ms, no gain: the ray-sorted queue already coalesces.)
#define TVR_Q0(t, h) (4 * (t) + 2 * (h))
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
        // 32-bit texel offsets against the (wave-uniform) plane base: SGPR-base addressing, no 64-bit per-lane address registers (measured:
        // 15.4 vs 15.8 ms with 64-bit per-lane addresses)
        const unsigned o0 = ((unsigned)y0 * (unsigned)Wp + (unsigned)x0) * 12u + (unsigned)q0, o1 = o0 + (unsigned)Wp * 12u;
        const float4 *p = P + o0, *p2 = P + o1;
        T.t[0][0] = p[0]; T.t[0][1] = p[1];
        T.t[1][0] = p[12]; T.t[1][1] = p[13];
        T.t[2][0] = p2[0]; T.t[2][1] = p2[1];
        T.t[3][0] = p2[12]; T.t[3][1] = p2[13];
        const float4 *q = Ln + ((unsigned)l0 * 12u + (unsigned)q0);
        T.lv[0][0] = q[0]; T.lv[0][1] = q[1];
        T.lv[1][0] = q[12]; T.lv[1][1] = q[13];
    } else {
        // arbitrary coordinates (the API's lookups): every tap is fetched from a CLAMPED cell with the same 32-bit offsets, and taps_eval<true> gives the taps
        // that lie outside the grid the weight zero — grid_sample's zeros padding without a select per fetched value.  (Rounds 1-3 selected the 12 float4 of
        // every k-step against zero behind 64-bit per-tap addresses: 256 registers, spills, 25 us per tile.)
        const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
        const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
        const int lc0 = min(max(l0, 0), L - 1), lc1 = min(max(l0 + 1, 0), L - 1);
        const unsigned r0 = (unsigned)yc0 * (unsigned)Wp, r1 = (unsigned)yc1 * (unsigned)Wp;
        const float4 *p00 = P + ((r0 + (unsigned)xc0) * 12u + (unsigned)q0), *p01 = P + ((r0 + (unsigned)xc1) * 12u + (unsigned)q0);
        const float4 *p10 = P + ((r1 + (unsigned)xc0) * 12u + (unsigned)q0), *p11 = P + ((r1 + (unsigned)xc1) * 12u + (unsigned)q0);
        T.t[0][0] = p00[0]; T.t[0][1] = p00[1];
        T.t[1][0] = p01[0]; T.t[1][1] = p01[1];
        T.t[2][0] = p10[0]; T.t[2][1] = p10[1];
        T.t[3][0] = p11[0]; T.t[3][1] = p11[1];
        const float4 *q0p = Ln + ((unsigned)lc0 * 12u + (unsigned)q0), *q1p = Ln + ((unsigned)lc1 * 12u + (unsigned)q0);
        T.lv[0][0] = q0p[0]; T.lv[0][1] = q0p[1];
        T.lv[1][0] = q1p[0]; T.lv[1][1] = q1p[1];
    }
}

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
    const float wx = fx - x0f, wy = fy - y0f;
    float wlf = fl - l0f, ulf = 1.0f - wlf;
    const float ux = 1.0f - wx, uy = 1.0f - wy;
    float a00 = ux * uy, a01 = wx * uy, a10 = ux * wy, a11 = wx * wy;
    if (CHECK) {                                                    // taps outside the grid: weight zero (load_taps<true> fetched a clamped cell for them)
        const int x0 = (int)x0f, y0 = (int)y0f, l0 = (int)l0f;
        const bool xi0 = (x0 >= 0) && (x0 < W), xi1 = (x0 + 1 >= 0) && (x0 + 1 < W);
        const bool yi0 = (y0 >= 0) && (y0 < H), yi1 = (y0 + 1 >= 0) && (y0 + 1 < H);
        a00 = (xi0 && yi0) ? a00 : 0.0f; a01 = (xi1 && yi0) ? a01 : 0.0f;
        a10 = (xi0 && yi1) ? a10 : 0.0f; a11 = (xi1 && yi1) ? a11 : 0.0f;
        ulf = ((l0 >= 0) && (l0 < L)) ? ulf : 0.0f;
        wlf = ((l0 + 1 >= 0) && (l0 + 1 < L)) ? wlf : 0.0f;
    }
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
            f32x2 p = ip_mul(w00, t0);
            p = ip_fma(w01, t1, p);
            p = ip_fma(w10, t2, p);
            p = ip_fma(w11, t3, p);
            f32x2 q = ip_mul(ul, l0);
            q = ip_fma(wl, l1, q);
            f32x2 r = ip_mul(p, q);
#if TVR_PIN_PK == 1
            asm volatile("" : "+v"(r.x), "+v"(r.y));
#endif
            out[g * 4 + hh * 2] = r.x;
            out[g * 4 + hh * 2 + 1] = r.y;
        }
    }
}

// sine / cosine for the positional encoding: v_sin_f32 / v_cos_f32 (they take revolutions) behind a two-term Cody-Waite reduction
// x - k*2pi, 7 instructions per pair and no branch (a branch per value kept hipcc from interleaving layer 1's VALU work with its MFMAs).
// The reduction is exact to an ulp of the remainder for |x| < ~1e4 (k has <= 11 bits), so the error is the hardware's (~1e-6 abs),
// the same as round 1's fract() form had for small |x| (scripts/accuracy_report.py), and smaller than that form's for |x| > 100.
__device__ __forceinline__ void sincos_pe(float x, float &s, float &c)
{
#if TVR_FAST_SINCOS
    const float tf = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
    s = __builtin_amdgcn_sinf(tf);
    c = __builtin_amdgcn_cosf(tf);
    return;
#endif
    const float k = rintf(x * 0.15915494309189535f);
    float r = __builtin_fmaf(k, -6.2831854820251465f, x);
    r = __builtin_fmaf(k, 1.7484555e-7f, r);                   // 2pi = 6.2831854820251465 - 1.7484555e-7
    const float t = r * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}

#define TVR_F16_MAX 65504.0f
__device__ __forceinline__ float absmax2(float a, float b, float m) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), m); }      // one v_max3_f32 |a|, |b|, m

