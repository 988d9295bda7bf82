// tvr_mfma.h — fp16 hi/lo split MFMA idioms shared by the fused-MLP kernels (tvr_ngp.hip, tvr_bg.hip).
// Arithmetic: v_mfma_f32_32x32x16_f16 with every fp32 operand split into fp16 hi + lo and three products per k-step (hi*lo, lo*hi,
// hi*hi): error ~2^-22 relative, i.e. fp32-grade.  Operand layout (lane l, col = l%32, h = l/32): A fragment = 8 halves of row col,
// k = 8h..8h+7 of the step; B fragment = 8 halves of column col, same k; accumulator register i = row (i/4)*8 + 4h + i%4, column col.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)

// relu as ONE VALU op: integer max on the bit pattern (negative floats, -0.0 included, are negative integers)
__device__ __forceinline__ float relu_f(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// fp32 pair -> packed fp16 hi and lo words (x = hi + lo up to ~2^-22 |x|; round-toward-zero never overflows to inf).  x - hi is one
// v_fma_mix_f32 reading the packed half in place (exact in fp32), then one v_cvt_pkrtz per pair: 4 VALU per pair.  (Measured alternative:
// v_fma_mixlo_f16 + v_fma_mixhi_f16 write the rounded residuals straight into the two halves of the lo word, 3 VALU per pair, -108 VALU per
// shade tile — and 1 % SLOWER, 12.89 -> 13.0 ms: the second op depends on the first through the destination register.  Round 3: the same with the
// four low halves first and the four high halves behind them in one asm block, no op following the one whose destination it completes: 13.20 vs 12.72 ms —
// the mixlo / mixhi forms are half rate, scripts/hwprobe/valu_rate.hip.)
// Round 4: the v_fma_mix_f32 is the COMPILER's (rounds 1-3 wrote it as inline asm, which the scheduler cannot classify: in the shade kernel's pipelined matrix
// phase those 8 instructions per fragment drifted to the end of their k-step).  hipcc selects it for fma(fpext(half), m, x) as long as it cannot fold the
// multiplier: m = -1.0 sits in an SGPR behind an empty asm (a literal -1 turns the fma into a subtraction with a separate v_cvt_f32_f16).
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    float m1 = -1.0f;
    asm("" : "+s"(m1));
    const fp16x2 hb = __builtin_amdgcn_cvt_pkrtz(a, b);
    const float ra = __builtin_fmaf((float)hb.x, m1, a), rb = __builtin_fmaf((float)hb.y, m1, b);
    hi = __builtin_bit_cast(unsigned, hb);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
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
// split8 with the four pairs' operations BATCHED — four cvt_pkrtz, eight v_fma_mix_f32, four cvt_pkrtz — instead of pair by pair: written pair by pair, hipcc emits
// `v_fma_mix_f32, v_fma_mix_f32, s_nop 0, v_cvt_pkrtz` for every pair (the packed conversion reads the second residual one instruction after it is written: a wait state),
// 60 - 100 s_nop per shade tile in a kernel whose vector-issue port is full (round 6).  The same operations on the same values: bit-identical fragments.
__device__ __forceinline__ Frag split8b(const float v[8])
{
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    float m1 = -1.0f;
    asm("" : "+s"(m1));
    fp16x2 hb[4];
    float r[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) hb[p] = __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        r[2 * p] = __builtin_fmaf((float)hb[p].x, m1, v[2 * p]);
        r[2 * p + 1] = __builtin_fmaf((float)hb[p].y, m1, v[2 * p + 1]);
    }
    Frag f;
    f.hi = make_uint4(__builtin_bit_cast(unsigned, hb[0]), __builtin_bit_cast(unsigned, hb[1]), __builtin_bit_cast(unsigned, hb[2]), __builtin_bit_cast(unsigned, hb[3]));
    f.lo = make_uint4(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r[0], r[1])), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r[2], r[3])),
                      __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r[4], r[5])), __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r[6], r[7])));
    return f;
}
// the one-part form of the reduced-product arithmetics (tvr_shade.hip, AR < 3): round to nearest even, one v_cvt_pk_f16_f32 per pair; no lo part.
// (|x| > 65504 becomes inf here, not 65504: the range check / the host's proof keeps such values out, include/tvr.h)
__device__ __forceinline__ Frag round8(const float v[8])
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    Frag f;
    f.hi.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[0], v[1]}, f16x2));
    f.hi.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[2], v[3]}, f16x2));
    f.hi.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[4], v[5]}, f16x2));
    f.hi.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v[6], v[7]}, f16x2));
    f.lo = make_uint4(0u, 0u, 0u, 0u);
    return f;
}
template <int AR>
__device__ __forceinline__ Frag frag8(const float v[8])
{
    if constexpr (AR >= 3) return split8(v);
    else return round8(v);
}
