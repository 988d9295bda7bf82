// tvr_shade.hip — kernel 2 of the render path: appearance VM lookup + basis + positional encoding + 3-layer MLP.
//
// Work it replaces in the reference (paths relative to /root/reference/tensorf-myc/):
//   models/tensoRF.py:228-244 compute_appfeature (6 grid_samples + Linear(144->27, no bias)),
//   models/tensorBase.py:9-15 positional_encoding, :76-86 MLPRender_Fea.execute (150->128->128->3, sigmoid).
//
// Design (CDNA4): every wave is an independent pipeline over 32 queue entries, entry = MFMA column (lane & 31), and the
// whole chain stays in registers — no LDS round trip, no barrier in the tile loop.  Per tile a wave runs two phases:
//   GATHER phase (global loads + VALU + LDS, no MFMA)
//     gather   the queue entry arrives one tile ahead; lane (e, h) fetches 8 channels (2 x float4) of each tap of entry e per k-step; the
//              interpolated plane*line products ARE the B fragments of the basis product (9 k-steps, kept as fp16 hi/lo in 72 registers);
//              the phase ends with s_waitcnt vmcnt(0), then the wave takes its SIMD's matrix token (see TVR_MTOKEN)
//   MATRIX phase (MFMA + LDS + VALU, no global load)
//     basis    F^T[32 x 32e]   = Bas[32 x 144] · h^T          A (basis) fragments: hi parts in LDS, lo parts fetched at the end of the gather phase
//     L1       H^T[128 x 32e]  = W1[128 x 160] · X^T          A (weights) from LDS (resident for the workgroup's lifetime); the B
//              fragments [v, sin v, sin 2v, cos v, cos 2v] of the 16 base values a lane owns are derived step by step between the
//              MFMAs (W1's columns are permuted to this order at pack time; b1 is the column of a constant-1 input)
//     L2       H2^T = W2 · relu(H^T): B = the layer-1 accumulators converted in place ("accumulator tile as the next MFMA's
//              operand": the k order inside a step is a fixed permutation, folded into the packed weight columns), kept as eight pre-split
//              fragments; the layer runs ROW BLOCK BY ROW BLOCK, and layer 3 of row block rb - 1 (fp32 FMAs, W3 as fp32 in LDS) runs
//              under the MFMAs of row block rb; token handed over behind the last MFMA
//   finish     layer 3 of the last row block + sigmoid + store, beside the partner's matrix phase
//   The matrix phase is an explicit software pipeline (fragment ring + sched_group_barrier windows, see below); since round 4 the kernel is measured to
//   be POWER-bound on the full chip (1.67 GHz on 256 CUs, 2.34 GHz on 128).
// Arithmetic: v_mfma_f32_32x32x16_f16 with every fp32 operand split into fp16 hi + lo and three products per step
// (hi·hi + hi·lo + lo·hi, fp32 accumulate): ~2^-22 relative error per product — fp32-class accuracy at 16/3 the rate of
// the fp32-input MFMA.  Range: operands pass through fp16, so |x| <= 65504 (cvt_pkrtz saturates) and parts below 6e-8 flush;
// features up to |v| ~ 1e3 and the reference's 1e-4-scale initialisation are tested (tests/test_gpu_parity.py); leaving the range is never silent (template parameter RC, include/tvr.h).
// Template parameter AR (tvr_scene_set_arith): the render and mlp_render kernels of TensorVMSplit also exist with TWO products in layers 1 and 2 (their inputs
// rounded to fp16, weights hi + lo; the basis product keeps three) and with ONE product everywhere — opt-in trades inside the 1e-3 RGB bar for a kernel that is power-bound.
//
// Phase rule: a round-1 build that issued global loads between the MFMAs of a tile returned wrong 16-lane groups on
// some boxes.  The mechanism was not established (round 2: the candidate mechanisms are excluded by probes, and that build renders clean
// today), so the kernel keeps global LOADS and MFMAs in separate phases and scripts/isa_check.py verifies that on the shipped ISA
// (tests/test_isa_rules.py): no VMEM load between the first and the last MFMA of the tile body, no load into a register an MFMA issued
// within the previous 16 instructions reads, and an independent recount of every s_waitcnt vmcnt.
#include <cstdlib>
#include "tvr_device.h"
#include "tvr_kernels.h"
#include "tvr_mfma.h"       // split2 / split8 / Frag / relu_f: the fp16 hi/lo split idioms shared with tvr_bg.hip, tvr_ngp.hip, tvr_mlp_train.hip
#include "tvr_shade_common.h"   // Taps / load_taps / taps_eval, pk_fma / pk_mul, sincos_pe, sigmoid_f, absmax2 (shared with tvr_shade16.hip)


#define TVR_SB __builtin_amdgcn_sched_barrier(0)
// the split is plain arithmetic: without a use in the gather phase hipcc sinks all nine of them (144 VALU) behind the matrix token
// (round-4 experiment, measured and not kept: only k-step 0's fragment split in the gather phase, k-steps 1..8 under the basis product's MFMAs, whose windows are
// empty — 128 VALU instructions leave the gather phase, -700 cycles per tile, and the token is held 250 cycles longer: 12.73 ms against 12.62, profiles/r04_shade_schedule_ab.txt)
#define TVR_GATHER_KEEP(s_)                                                                            \
    do {                                                                                               \
        if (RC) { _Pragma("unroll") for (int j_ = 0; j_ < 8; j_ += 2) rmax = absmax2(hvv[s_][j_], hvv[s_][j_ + 1], rmax); asm volatile("" : "+v"(rmax)); } \
        hf[s_] = frag8<ARB>(hvv[s_]);                                                                  \
        TVR_PIN_FRAG(hf[s_]);                                                                          \
    } while (0)
#define TVR_PIN_FRAG(f)                                                                                \
    do {                                                                                               \
        if constexpr (ARB >= 3) asm volatile("" : "+v"((f).hi.x), "+v"((f).hi.y), "+v"((f).hi.z), "+v"((f).hi.w), "+v"((f).lo.x), "+v"((f).lo.y), "+v"((f).lo.z), "+v"((f).lo.w)); \
        else asm volatile("" : "+v"((f).hi.x), "+v"((f).hi.y), "+v"((f).hi.z), "+v"((f).hi.w));       /* one-product mode: the lo part has no consumer and is never computed */ \
    } while (0)
#ifndef TVR_PF
#define TVR_PF 1          // gather prefetch distance in k-steps (ring of TVR_PF + 1 tap sets) ...
#endif
#ifndef TVR_PF_SHALLOW
#define TVR_PF_SHALLOW 5  // ... for k-steps below this one; 1 from here on (see the gather loop).  Round 4 measured 2 / {3, 4, 5, 6} against 1: 12.42-12.51 ms
#endif                    // against 12.41 (gpurun_out/r4e) — no gain: with two waves per SIMD the gather phase hides under the partner's matrix phase
#define TVR_CHK (SRC != SH_SRC_QUEUE)
#ifndef TVR_TIMING
#define TVR_TIMING 0      // diagnostic build: per-phase s_memtime sums into stats[8..14] (scripts/phase_timing.py passes 16 slots)
#endif
#if TVR_TIMING
#define TVR_STAMP(x) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define TVR_STAMP(x)
#endif
#ifndef TVR_TICKET
#define TVR_TICKET 4      // tiles per ticket of the render path's dynamic tile hand-out (0: static stride)
#endif
#ifndef TVR_TICKET_MIN
#define TVR_TICKET_MIN 256  // tiles per wave from which the tickets are used
#endif
#ifndef TVR_PHASE_FREE
#define TVR_PHASE_FREE 0  // experiment (round 5): 1 = the render kernel fetches the NEXT tile's k-step-0 taps (12 global loads per lane) between layer 1 and
#endif                    // layer 2 of the current tile, i.e. INSIDE the matrix phase — the build that settles whether the phase rule protects anything (scripts/phase_rule_test.sh)
#ifndef SH_WAVES
#define SH_WAVES 8        // two waves per SIMD
#endif
#define SH_THREADS (64 * SH_WAVES)
#define SH_MINW (SH_WAVES / 4)
#define SH_TILE 32

#ifndef TVR_PRIO_F
#define TVR_PRIO_F 0      // s_setprio of finish_tile (layer 3 + store), see the call site
#endif
#ifndef TVR_PRIO_G
#define TVR_PRIO_G 0      // s_setprio while a wave is in its gather phase / its matrix phase.  Rounds 1-3 ran the gather ABOVE the matrix phase (2 / 0: hipcc's
#endif                    // schedule of the matrix phase left its own VALU work outside the MFMAs' shadow anyway, and the gather's dependent load chains gained
#ifndef TVR_PRIO_M        // 6 %).  With the matrix phase as an explicit pipeline (round 4) every issue slot it loses to the partner is matrix-pipe idle time:
#define TVR_PRIO_M 2      // matrix 2 / gather 0 takes 12.17 ms against 12.48 for 0 / 2 (gpurun_out/r4d, two interleaved rounds; 2 / 1: 12.28, 3 / 0: 12.18;
#endif                    // round 3's schedule at 2 / 0: 12.81 against its own 12.62 at 0 / 2).

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

// one k-step of a 128-row layer: the three hi/lo products interleaved across the row blocks (no MFMA directly follows an MFMA it depends on)
__device__ __forceinline__ void mfma3x4(const AFrag4 &A, const Frag &b, f32x16 acc[4])
{
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.l[rb]), __builtin_bit_cast(h8, b.hi), acc[rb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.h[rb]), __builtin_bit_cast(h8, b.lo), acc[rb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.h[rb]), __builtin_bit_cast(h8, b.hi), acc[rb], 0, 0, 0);
}

// ---- round 4: the matrix phase as an explicit software pipeline (round 4) ------------------------------------------------------------
// hipcc's own schedule of the hidden layers (round 3; scripts/isa_trace.py shows it) issued a k-step's A-fragment reads right in front of the MFMAs that
// consume them (ds_read x4, s_waitcnt, MFMA: 70 exposed LDS round trips per tile) and clumped MFMAs (8 back to back) apart from the VALU work of the next
// fragment.  What a wave's own stream can put under its MFMAs was measured (scripts/hwprobe/mfma_issue.hip, mfma_issue2.hip -> profiles/r04_mfma_issue_probe.txt,
// one wave per SIMD, one asm block per iteration): an MFMA holds the pipe for 32 cycles and the wave's issue for 8; up to FIVE independent VALU instructions
// behind every MFMA are free (32.0 - 32.3 cycles per MFMA, one accumulation chain or four, arch or Acc VGPRs alike), the sixth is not (34.0); a ds_read_b128
// costs a window about as much as two or three of them (3 VALU per MFMA + 2 reads per 3 MFMAs: 36.1; 2 VALU: 33.5; reads alone: 32.0).  Round 3's statement
// that a wave's VALU work does not run under its own MFMAs described hipcc's clumps, not the hardware.  Now:
//   * the MFMAs of a k-step go row block by row block (l*hi, h*lo, h*hi on one accumulator: a dependent chain issues back to back at 32 cycles), so an A
//     fragment lives for three MFMAs and a ring of TVR_PD + 2 {hi, lo} pairs, read TVR_PD row blocks ahead, replaces the two whole-k-step fragment sets
//     (64 -> 32 registers; the kernel went from 246 to 210 VGPRs);
//   * __builtin_amdgcn_sched_group_barrier pins the order inside a k-step (one scheduling region): per row block {read} M V0 {read} M V1 M V2, the VALU
//     instructions being the NEXT k-step's B fragment (positional encoding / relu + fp16 split); the windows that carry a read carry fewer of them;
//   * the fp16 split's v_fma_mix_f32 is the compiler's own instruction now (tvr_mfma.h), so that the scheduler can classify it;
//   * layer 1's last k-step carries layer 2's prologue, the basis product is one accumulation chain with its A fragments two k-steps ahead.
// Per accumulator the order of the additions in the hidden layers is the one mfma3x4 has: their results are bit-identical to round 3's.
// What it bought, honestly (profiles/r04_shade_schedule_ab.txt, interleaved rounds on one box each): the token is held 9.4 k cycles instead of 10.9 k, the tile
// takes 23.0 k cycles per wave instead of 24.8 k, the kernel 12.2 - 12.6 ms instead of 12.6 - 13.0 (-3 %): with two in-order waves per SIMD the gather phase, the
// token wait and layer 3 are on the same critical chain as the matrix phase, and a lone wave's hidden layers run at 39 cycles per MFMA whatever the order of their
// instructions (32.5 without the fragment derivation, 32.1 without the fragment reads, 38.8 with both: gpurun_out/r4f) — issue-bound, not pipe-bound.
#ifndef TVR_DIAG
#define TVR_DIAG 0        // diagnostic builds only (wrong pictures): 1 = no fragment derivation in the hidden layers, 2 = no A-fragment reads in layer 1, 4 = layer 1 reads k-step 0's fragments in every k-step (the same LDS traffic, constant operands)
#endif
#define TVR_NV1 3         // VALU (+ transcendental) instructions behind each MFMA of layer 1's last k-step (layer 2's first fragment)
// The three MFMA windows of a row block: TVR_PIPE 1 puts one fragment read in each of the first two windows and V0 / V1 / V2 VALU instructions behind the three
// MFMAs (layer 1: 1 / 3 / 5, layer 2: 0 / 2 / 4; 2 / 3 / 4 and 1 / 4 / 4 measured 5 % slower, 0 / 3 / 6 and 0 / 4 / 5 equal); TVR_PIPE 0: both reads in front of
// the row block and 3 / 3 / 3 (2 / 2 / 2): 3 % slower.
// Round 4, second step: layer 2 runs ROW BLOCK BY ROW BLOCK over eight pre-split fragments of relu(layer 1) (the 64 registers the layer-1 accumulators
// leave), and layer 3 of row block rb - 1 — 16 relu, 48 FMA, 12 weight reads — runs under the MFMAs of row block rb: only the last row block's share of layer 3 is
// left behind the token hand-over.  Layer 3 was 3.0 k cycles of every wave's chain (gather -> token -> matrix -> layer 3) with nothing beside it.
#ifndef TVR_PIPE
#define TVR_PIPE 1
#endif
#ifndef TVR_PD
#define TVR_PD 2          // A fragments are read this many row blocks ahead of their MFMAs, into a ring of TVR_PD + 2 {hi, lo} pairs
#endif
#define TVR_RN (TVR_PD + 2)
#if TVR_PIPE == 0
#define TVR_L1_V0 3
#define TVR_L1_V1 3
#define TVR_L1_V2 3
#define TVR_L2_V0 2
#define TVR_L2_V1 2
#define TVR_L2_V2 2
#define TVR_PIPE_RB(has_read, v0, v1, v2)                                                              \
    do {                                                                                               \
        if (has_read) TVR_SG_DSR(2);                                                                   \
        TVR_SG_MFMA(1); if (v0) TVR_SG_VALU(v0);                                                       \
        TVR_SG_MFMA(1); if (v1) TVR_SG_VALU(v1);                                                       \
        TVR_SG_MFMA(1); if (v2) TVR_SG_VALU(v2);                                                       \
    } while (0)
#else
#ifndef TVR_L1_V0
#define TVR_L1_V0 1
#define TVR_L1_V1 3
#define TVR_L1_V2 5
#endif
#ifndef TVR_L2_V0
#define TVR_L2_V0 0
#define TVR_L2_V1 2
#define TVR_L2_V2 4
#endif
#define TVR_PIPE_RB(has_read, v0, v1, v2)                                                              \
    do {                                                                                               \
        if (has_read) TVR_SG_DSR(1);                                                                   \
        TVR_SG_MFMA(1); if (v0) TVR_SG_VALU(v0);                                                       \
        if (has_read) TVR_SG_DSR(1);                                                                   \
        TVR_SG_MFMA(1); if (v1) TVR_SG_VALU(v1);                                                       \
        TVR_SG_MFMA(1); if (v2) TVR_SG_VALU(v2);                                                       \
    } while (0)
#endif
#ifndef TVR_L1A2_V0
#define TVR_L1A2_V0 2     // layer-1 windows of the two-product arithmetic (two MFMAs per row block) ...
#define TVR_L1A2_V1 4
#endif
#ifndef TVR_L1A1_V0
#define TVR_L1A1_V0 6     // ... and of the one-product arithmetic
#endif
// the same windows for the reduced-product arithmetics (AR 2: two MFMAs and two reads per row block; AR 1: one and one)
#define TVR_PIPE_RB2(has_read, v0, v1)                                                                 \
    do {                                                                                               \
        if (has_read) TVR_SG_DSR(1);                                                                   \
        TVR_SG_MFMA(1); if (v0) TVR_SG_VALU(v0);                                                       \
        if (has_read) TVR_SG_DSR(1);                                                                   \
        TVR_SG_MFMA(1); if (v1) TVR_SG_VALU(v1);                                                       \
    } while (0)
#define TVR_PIPE_RB1(has_read, v0)                                                                     \
    do {                                                                                               \
        if (has_read) TVR_SG_DSR(1);                                                                   \
        TVR_SG_MFMA(1); if (v0) TVR_SG_VALU(v0);                                                       \
    } while (0)
#if TVR_DIAG & 16
#define TVR_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, 2 * (n), 0)
#else
#define TVR_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#endif
#define TVR_SG_VALU(n) __builtin_amdgcn_sched_group_barrier(0x402, (n), 0)      // VALU | TRANS
#define TVR_SG_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
struct AF { uint4 h, l; };
// AR = the products a k-step takes (tvr_scene_set_arith): 3 = Wlo*xhi + Whi*xlo + Whi*xhi (fp32-class, the default); 2 = Wlo*xhi + Whi*xhi (weights keep their
// 22 bits, activations are rounded to fp16); 1 = Whi*xhi (plain fp16 operands).  Always fp32 accumulation.  What a mode does not multiply is neither read nor derived.
template <int AR = 3>
__device__ __forceinline__ void load_af(AF &A, const unsigned char *WH, const unsigned char *WL, int off)
{
    A.h = *(const uint4 *)(WH + off);
    if constexpr (AR >= 2) A.l = *(const uint4 *)(WL + off);
}
template <int AR = 3>
__device__ __forceinline__ void mfma3(const AF &A, const Frag &b, f32x16 &acc)
{
#if TVR_DIAG & 16
    // timing experiment (round 5, wrong pictures): what the 16x16x32 shape would cost / save IN this kernel — every 32x32x16 MFMA becomes two v_mfma_f32_16x16x32_f16 (the
    // same cycles, FLOPs, operand registers and LDS bytes: one {hi, lo} A pair per six MFMAs, as two 16-column B tiles sharing each A fragment would issue them)
    if constexpr (AR == 3) {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        f32x4_ q0 = __builtin_shufflevector(acc, acc, 0, 1, 2, 3), q1 = __builtin_shufflevector(acc, acc, 4, 5, 6, 7);
        const h8 al = __builtin_bit_cast(h8, A.l), ah = __builtin_bit_cast(h8, A.h), bh = __builtin_bit_cast(h8, b.hi), bl = __builtin_bit_cast(h8, b.lo);
        q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, q1, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, q1, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, q1, 0, 0, 0);
        acc = __builtin_shufflevector(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(acc, acc, 8, 9, 10, 11, 12, 13, 14, 15), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        return;
    }
#endif
    if constexpr (AR >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.l), __builtin_bit_cast(h8, b.hi), acc, 0, 0, 0);
    if constexpr (AR >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.h), __builtin_bit_cast(h8, b.lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A.h), __builtin_bit_cast(h8, b.hi), acc, 0, 0, 0);
}
// byte offset of the A fragment of row block q & 3 of k-step q >> 2 inside a weight image
#define TVR_AOFF(q) (((q) >> 2) * TVR_IMG_STEP + ((q) & 3) * TVR_IMG_RB)
// an LDS pointer held in ONE register the compiler cannot see through (so that it addresses base + immediate instead of folding the base away)
#define TVR_LDS_BASE(name, expr)                                                                                    \
    unsigned name##_a = (unsigned)(size_t)(expr);                                                                   \
    asm volatile("" : "+v"(name##_a));                                                                              \
    const unsigned char *name = (const unsigned char *)(const void __attribute__((address_space(3))) *)(size_t)name##_a



// ---- the gather of the one-product arithmetic (AR == 1, queue source): the appearance factors as fp16 (SceneDev::aplane16 / aline16, 96 B per texel) — one 16-B load
// per tap and lane instead of two, half the bytes through the L1 return path that bounds this mode; the interpolation itself stays fp32 (v_fma_mix_f32 reads the
// halves in place: no conversion instructions).  Queue entries lie inside the grid: no bounds handling here.
struct Taps16 {
    uint4 t[4], lv[2];
};
__device__ __forceinline__ void load_taps16(Taps16 &T, const uint4 *__restrict__ P, const uint4 *__restrict__ Ln, int W, float fx, float fy, float fl, int q0)
{
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy), l0 = (int)floorf(fl);
    const unsigned Wp = (unsigned)W + 1u;
    const unsigned o0 = ((unsigned)y0 * Wp + (unsigned)x0) * 6u + (unsigned)q0, o1 = o0 + Wp * 6u;       // a texel = 48 halves = 6 uint4
    const uint4 *p = P + o0, *p2 = P + o1;
    T.t[0] = p[0]; T.t[1] = p[6];
    T.t[2] = p2[0]; T.t[3] = p2[6];
    const uint4 *q = Ln + ((unsigned)l0 * 6u + (unsigned)q0);
    T.lv[0] = q[0]; T.lv[1] = q[6];
}
__device__ __forceinline__ void taps_eval16(const Taps16 &T, float fx, float fy, float fl, float out[8])
{
    const float x0f = floorf(fx), y0f = floorf(fy), l0f = floorf(fl);
    const float wx = fx - x0f, wy = fy - y0f, wl = fl - l0f, ul = 1.0f - wl;
    const float ux = 1.0f - wx, uy = 1.0f - wy;
    const float a00 = ux * uy, a01 = wx * uy, a10 = ux * wy, a11 = wx * wy;
    const h8 t0 = __builtin_bit_cast(h8, T.t[0]), t1 = __builtin_bit_cast(h8, T.t[1]), t2 = __builtin_bit_cast(h8, T.t[2]), t3 = __builtin_bit_cast(h8, T.t[3]);
    const h8 l0 = __builtin_bit_cast(h8, T.lv[0]), l1 = __builtin_bit_cast(h8, T.lv[1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                  // the same order of operations as taps_eval
        float p = __builtin_fmaf((float)t0[j], a00, 0.0f);
        p = __builtin_fmaf((float)t1[j], a01, p);
        p = __builtin_fmaf((float)t2[j], a10, p);
        p = __builtin_fmaf((float)t3[j], a11, p);
        float q = __builtin_fmaf((float)l0[j], ul, 0.0f);
        q = __builtin_fmaf((float)l1[j], wl, q);
        float r = p * q;
        asm volatile("" : "+v"(r));
        out[j] = r;
    }
}

// what a tile hands from its matrix phase to finish_tile (layer 3 + epilogue)
struct Carry {
    f32x16 acc2;                 // layer-2 accumulators of row block 3 (b2 included), hidden unit 96 + acc_row(r, h) in acc2[r]
    long long ent;
    float wq;
    float g[4];                  // REFTensoRF: specular tint and rgb_d
    float rmax;                  // RC: max |x| over this lane's share of the entry's fp16-split operands
    f32x2 s0, s1, s2;            // layer 3's partial sums over row blocks 0..2, taken under layer 2's MFMAs (two chains per output); acc2[3] is what finish_tile still has to do
    bool live;
};

// layer 3 (tensorBase.py:83-84: Linear(128 -> 3) on relu(h2), sigmoid) as fp32 FMAs: W3 [3][128] fp32 and b3 sit in the LDS image.
// Lane (e, h) holds 64 of entry e's 128 hidden units; the two halves are added through the LDS crossbar.  Fixed summation order.
template <int DST, bool REF, bool HAVE_G, bool RC>
__device__ __forceinline__ void finish_tile(const Carry &c, const unsigned char *smem, const ShadeArgs &a, int h)
{
    f32x2 s0 = c.s0, s1 = c.s1, s2 = c.s2;
    // ONE address register (16-bit immediate offsets cover W3's 1.5 KB; the image offset itself does not fit an immediate, and without
    // the opaque base hipcc materialises 48 loop-invariant address registers and spills them)
    unsigned w3a = (unsigned)(size_t)(smem + TVR_IMG_W3) + 16u * (unsigned)h;
    asm volatile("" : "+v"(w3a));
    const float *W3 = (const float *)(const void __attribute__((address_space(3))) *)(size_t)w3a;
    // row block 3 (hidden units 96..127; row blocks 0..2 were taken under layer 2's MFMAs, same order of additions): its 12 weight quads in one LDS round trip
    float4 w[12];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) w[q * 3 + c3] = *(const float4 *)(W3 + c3 * 128 + 32 * 3 + 8 * q);      // hidden units u + 4h .. + 3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w0 = w[q * 3], w1 = w[q * 3 + 1], w2 = w[q * 3 + 2];
        const f32x2 xa = {relu_f(c.acc2[4 * q]), relu_f(c.acc2[4 * q + 1])};
        const f32x2 xb = {relu_f(c.acc2[4 * q + 2]), relu_f(c.acc2[4 * q + 3])};
        if (DST == SH_DST_TRAIN && c.live)               // relu(layer 2), hidden units 96 + 8 q + 4 h .. + 3: the backward's mask and dW3's operand
            *(float4 *)(a.t_h2 + c.ent * TVR_FEATC + 96 + 8 * q + 4 * h) = make_float4(xa.x, xa.y, xb.x, xb.y);
        s0 = pk_fma(xa, f32x2{w0.x, w0.y}, s0); s0 = pk_fma(xb, f32x2{w0.z, w0.w}, s0);
        s1 = pk_fma(xa, f32x2{w1.x, w1.y}, s1); s1 = pk_fma(xb, f32x2{w1.z, w1.w}, s1);
        s2 = pk_fma(xa, f32x2{w2.x, w2.y}, s2); s2 = pk_fma(xb, f32x2{w2.z, w2.w}, s2);
    }
    float r0 = s0.x + s0.y, r1 = s1.x + s1.y, r2 = s2.x + s2.y;
    r0 += __shfl_xor(r0, 32); r1 += __shfl_xor(r1, 32); r2 += __shfl_xor(r2, 32);
    const float4 b3 = *(const float4 *)(smem + TVR_IMG_B3);
    r0 = sigmoid_f(r0 + b3.x); r1 = sigmoid_f(r1 + b3.y); r2 = sigmoid_f(r2 + b3.z);
    if (REF && DST == SH_DST_TRAIN && c.live && h == 0) {        // the MLP's own output: the backward needs sigmoid' and d rgb / d tint
        a.t_rgbs[c.ent * 3] = r0; a.t_rgbs[c.ent * 3 + 1] = r1; a.t_rgbs[c.ent * 3 + 2] = r2;
    }
    if (REF && HAVE_G) {                                         // REFTensoRF.py:232  specular_tint * clamp(rgb_s, 0) + rgb_d
        const float tint = fmaxf(c.g[0], 0.0f);
        r0 = tint * fmaxf(r0, 0.0f) + c.g[1]; r1 = tint * fmaxf(r1, 0.0f) + c.g[2]; r2 = tint * fmaxf(r2, 0.0f) + c.g[3];
    }
    if (RC) {                                                    // an operand of this entry left fp16's range: the colour is NaN, not a clipped product
        const float m = fmaxf(c.rmax, __shfl_xor(c.rmax, 32));
        if (!(m < TVR_F16_MAX)) r0 = r1 = r2 = __builtin_nanf("");
    }
    if (c.live && h == 0) {
        if (DST == SH_DST_QUEUE) {
            a.q_out[c.ent] = make_float4(r0, r1, r2, c.wq);
        } else {
            a.out[c.ent * 3] = r0; a.out[c.ent * 3 + 1] = r1; a.out[c.ent * 3 + 2] = r2;
        }
    }
}

// REF = REFTensoRF (models/REFTensoRF.py:107-133, 174-256): a second basis row block gives normal / diffuse / specular / rho from the
// same h, the view direction is replaced by the reflection about the normalised normal, layer 1 takes one more input (-dot) and the
// colour is  specular_tint * rgb_s + rgb_d.
// The two waves of a SIMD (w and w + 4) run the same program.  Left alone they CONVOY: while both are in their matrix phase they share the MFMA
// pipe, finish together, gather together (matrix pipe idle) and meet again at the next matrix phase — a stable state, whatever the start offset.
// TVR_MTOKEN 1: a per-SIMD token in LDS makes the matrix phase mutually exclusive, which locks the pair in anti-phase (one gathers while the other
// multiplies): 13.9 -> 12.55 ms on one box, interleaved rounds.  Unequal matrix-phase priorities for the two waves instead: 14.0, no effect;
// two tokens (basis + layer 1 | layer 2 as a two-stage pipeline): 12.98; spin back-off s_sleep 1 vs 8: equal; the token taken only at layer 1
// (basis product outside it): 12.88; that plus all sin / cos in front of the token: 13.5 (12.97 same box); matrix-phase priority above
// the gather's: 12.95; without the one-step-ahead LDS prefetch of the weight fragments: 13.2; the hidden layers' VALU / LDS work forced
// between their MFMAs one group per MFMA (sched_group_barrier: M vvv d M vvv d ... instead of hipcc's MMMM vvvvvvvvvvvvvv): 13.01 vs 12.95.
// Every path takes the token after its last global load has landed and gives it back before the next tile's first load, at most once per tile:
// a wave never waits for the token while holding it, so the spin always ends.
#ifndef TVR_MTOKEN
#define TVR_MTOKEN 1
#endif
#ifndef TVR_NLO_LDS
#define TVR_NLO_LDS 6     // k-steps whose basis LO fragments live in LDS (TensorVMSplit kernels; 6 x 864 B fit behind the image, 9 would not)
#endif
#ifndef TVR_MSLEEP
#define TVR_MSLEEP 2
#endif
#if TVR_MTOKEN
// The token only shapes the schedule: results do not depend on it.  The wait is therefore BOUNDED — after TVR_MTOKEN_SPINS polls (a legitimate wait
// is the partner's matrix phase, ~6 us = ~50 polls) the wave goes ahead WITHOUT the token and gives nothing back (`have_tok`), so a lost or
// stuck token can cost speed but never hang the grid or change a pixel.
#ifndef TVR_MTOKEN_SPINS
#define TVR_MTOKEN_SPINS 4096
#endif
#define TVR_TOKEN_TAKE(t)                                                                   \
    do {                                                                                    \
        int got_, n_ = 0;                                                                   \
        do {                                                                                \
            int r_ = 1;                                                                     \
            if (lane == 0) r_ = atomicCAS((t), 0, 1);                                       \
            got_ = __builtin_amdgcn_readfirstlane(r_);                                      \
            if (got_) __builtin_amdgcn_s_sleep(TVR_MSLEEP);                                 \
        } while (got_ && ++n_ < TVR_MTOKEN_SPINS);                                          \
        have_tok = !got_;                                                                   \
    } while (0)
#define TVR_TOKEN_GIVE(t) do { if (have_tok && lane == 0) atomicExch((t), 0); } while (0)
#ifndef TVR_TOKEN_PHASE
#define TVR_TOKEN_PHASE 1     // which phase the per-SIMD token makes mutually exclusive: 1 the matrix phase (shipped); 2 the GATHER phase (experiment: the two
#endif                        // waves' matrix phases may then overlap — one's splits / sin / cos under the other's MFMAs — and "both gathering, pipe idle" cannot happen)
#if TVR_TOKEN_PHASE == 1
#ifndef TVR_GEN_PF
#define TVR_GEN_PF 2       // k-steps between a thread's fetch of its uint4 of the streamed layer-1 image and the LDS store that stages it
#endif
// (GEN, the lockstep kernel of scenes with more than two encoding frequencies: no token — its waves meet at a barrier per layer-1 k-step anyway)
#define TVR_ENTER_MATRIX() do { if (!GEN) TVR_TOKEN_TAKE(mtok); __builtin_amdgcn_s_setprio(TVR_PRIO_M); } while (0)
#define TVR_LEAVE_MATRIX() do { if (!GEN) TVR_TOKEN_GIVE(mtok); __builtin_amdgcn_s_setprio(TVR_PRIO_G); } while (0)
#define TVR_ENTER_GATHER()
#else
#define TVR_ENTER_MATRIX() do { TVR_TOKEN_GIVE(mtok); __builtin_amdgcn_s_setprio(TVR_PRIO_M); } while (0)
#define TVR_LEAVE_MATRIX() __builtin_amdgcn_s_setprio(TVR_PRIO_G)
#define TVR_ENTER_GATHER() TVR_TOKEN_TAKE(mtok)
#endif
#else
#define TVR_ENTER_MATRIX() __builtin_amdgcn_s_setprio(TVR_PRIO_M)
#define TVR_LEAVE_MATRIX() __builtin_amdgcn_s_setprio(TVR_PRIO_G)
#define TVR_ENTER_GATHER()
#endif
// REFTensoRF's second row block on the same h fragments (REFTensoRF.py:126-132): A from the LDS image (8 weight rows; lanes 4..6 re-read the
// normal rows so that both lane halves hold the normal, every other lane reads the zero row), biases as the initial accumulator.
// Uses hf[], accA/B/C, G[], e, h, smem of the enclosing scope.
#define TVR_REF_HEADS()                                                                                                              \
    do {                                                                                                                            \
        const int rr = e < 4 ? e : (e < 7 ? e - 4 : ((e >= 8 && e < 12) ? e - 4 : -1));                                             \
        const unsigned char *rowp = rr >= 0 ? smem + TVR_IMG_REFW + rr * TVR_IMG_REF_ROW : smem + TVR_IMG_REF_ZROW;                 \
        const float4 g0 = *(const float4 *)(smem + TVR_IMG_REFB + 16 * h), g1 = *(const float4 *)(smem + TVR_IMG_REFB + 32 + 16 * h); \
        accA = f32x16{0}; accB = f32x16{0}; accC = f32x16{0};                                                                       \
        accA[0] = g0.x; accA[1] = g0.y; accA[2] = g0.z; accA[3] = g0.w;                                                             \
        accA[4] = g1.x; accA[5] = g1.y; accA[6] = g1.z; accA[7] = g1.w;                                                             \
        _Pragma("unroll") for (int s = 0; s < 9; ++s) {                                                                             \
            const uint4 *ap = (const uint4 *)(rowp + (s * 2 + h) * 32);                                                             \
            const h8 Ah = __builtin_bit_cast(h8, ap[0]), Al = __builtin_bit_cast(h8, ap[1]);                                        \
            if constexpr (ARB >= 2) {                               /* (one-product mode: accA keeps the biases, accB stays zero) */   \
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(h8, hf[s].hi), accA, 0, 0, 0);                 \
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, hf[s].lo), accB, 0, 0, 0);                 \
            }                                                                                                                       \
            accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(h8, hf[s].hi), accC, 0, 0, 0);                     \
        }                                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) G[r] = (accA[r] + accB[r]) + accC[r];                                         \
    } while (0)

// The phase boundary and the basis product  F^T[32 x 32e] = Bas[32 x 144] . h^T  (27 MFMAs).  Uses hf[], bal[], F[], bashp, baslp, NLO, have_tok, mtok, lane of
// the enclosing scope.  Phase boundary: every global load of this tile has landed before the first MFMA issues, and the compiler may not move loads
// below it.  Round 4: ONE accumulation chain (l*hi, h*lo, h*hi per k-step, as the hidden layers do — round 3 summed three chains with 32 VALU adds behind
// the last MFMA, pipe idle), the A fragments in a ring of four read two k-steps ahead of their use, the first two BEFORE the wave waits for its loads and
// for the matrix token.
#define TVR_BASIS_BLOCK()                                                                                                            \
    do {                                                                                                                            \
        constexpr int BST_ = 2 * TVR_IMG_BASH_ROWS * 16;                                                                            \
        AF br_[4];                                                                                                                  \
        auto bld_ = [&](int s_) {                                                                                                   \
            br_[s_ & 3].h = *(const uint4 *)(bashp + s_ * BST_);                                                                    \
            if constexpr (ARB >= 2) {                                                                                                \
                if (s_ < NLO) br_[s_ & 3].l = *(const uint4 *)(baslp + s_ * BST_);                                                  \
                else br_[s_ & 3].l = bal[s_];                                                                                       \
            }                                                                                                                       \
        };                                                                                                                          \
        bld_(0); bld_(1);                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                            \
        TVR_SB;                                                                                                                     \
        TVR_STAMP(tg1);                                                                                                             \
        TVR_ENTER_MATRIX();                                                                                                         \
        TVR_STAMP(tgW);                                                                                                             \
        f32x16 accF_ = f32x16{0};                                                                                                   \
        _Pragma("unroll") for (int s_ = 0; s_ < 9; ++s_) {          /* one scheduling region per k-step: reads issued here are consumed two regions on */ \
            if (s_ + 2 < 9) bld_(s_ + 2);                                                                                           \
            mfma3<ARB>(br_[s_ & 3], hf[s_], accF_);                                                                                  \
            if (s_ + 2 < 9) { if (s_ + 2 < NLO && ARB >= 2) TVR_SG_DSR(2); else TVR_SG_DSR(1); }                                     \
            TVR_SG_MFMA(ARB);                                                                                                        \
            TVR_SB;                                                                                                                 \
        }                                                                                                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) F[r_] = accF_[r_];                                                        \
        TVR_SB;                                                                                                                     \
        if (REF) { f32x16 accA, accB, accC; TVR_REF_HEADS(); }                                                                      \
    } while (0)

// RC (range check, tvr_scene_set_range_check; default ON for the inference entry points): every value that enters an MFMA through the fp16 hi / lo split — the
// interpolated appearance features, the basis outputs and the layer-1 inputs made of them, relu(layer 1) — feeds a running max|x| (one v_max3_f32 per two
// values, +4 % VALU); an entry whose maximum reaches fp16's largest finite value (cvt_pkrtz saturates there, silently) gets NaN as its colour / features, so the
// pixel it belongs to comes out NaN instead of wrong.  Weights are checked by the host (field.py::_fp16_range_proven), which also switches RC off for scenes
// whose interval bounds prove that nothing can leave the range.
template <int SRC, int DST, bool REF, bool RC = false, bool GEN = false, int AR = 3>
__global__ __launch_bounds__(SH_THREADS, SH_MINW) void shade_kernel(const SceneDev sc, const ShadeArgs a)
{
    static_assert(AR == 3 || (!GEN && DST != SH_DST_TRAIN && DST != SH_DST_FEAT), "the reduced-product arithmetics exist for the render and mlp_render paths of scenes with at most two encoding frequencies");
    // the basis product F = Bas . h keeps its three products in the two-product mode: F feeds sin / cos (d sin(2F) / dF = 2, an error of F is AMPLIFIED by |F|'s
    // scale), whereas a rounded input of a linear layer is not (27 of the 243 MFMAs)
    constexpr int ARB = AR == 1 ? 1 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 31, h = lane >> 5;
    // MLP weights -> LDS once per workgroup (the xyz -> features kernel of TensorVMSplit needs only the basis fragments' hi parts)
    constexpr bool BAS_ONLY = DST == SH_DST_FEAT && !REF;
    constexpr int BASH = BAS_ONLY ? 0 : TVR_IMG_BASH;
    // round 3: the LO parts of the basis fragments of the first NLO k-steps also live in LDS (27-row packing like the hi parts), in the 5.5 KB the
    // TensorVMSplit image leaves free: 6 of the 9 per-tile loads of those fragments (L1 hits, but the L1 instruction rate is what binds the gather)
    // and 24 of the 36 registers they held across the phase boundary are gone.  REFTensoRF's image fills the LDS: NLO = 0 there.
    constexpr int NLO = REF ? 0 : (BAS_ONLY ? 9 : TVR_NLO_LDS);
    constexpr int IMG_END = (BAS_ONLY ? (TVR_MLP_IMAGE_BYTES - TVR_IMG_BASH) : (REF ? TVR_MLP_IMAGE_BYTES_REF : TVR_MLP_IMAGE_BYTES));
    constexpr int BASL = IMG_END + 16;                 // behind the four matrix-token words
    {
        const uint4 *src = (const uint4 *)((const unsigned char *)sc.mlp_image + (BAS_ONLY ? TVR_IMG_BASH : 0));
        constexpr int nb = BAS_ONLY ? (TVR_MLP_IMAGE_BYTES - TVR_IMG_BASH) : (REF ? TVR_MLP_IMAGE_BYTES_REF : TVR_MLP_IMAGE_BYTES);
        for (int i = tid; i < nb / 16; i += SH_THREADS) ((uint4 *)smem)[i] = src[i];
#if TVR_MTOKEN
        if (tid < 4) ((int *)(smem + nb))[tid] = 0;
#endif
        for (int i = tid; i < NLO * 2 * TVR_IMG_BASH_ROWS; i += SH_THREADS) {            // global [s][half][32 rows] -> LDS [s][half][27 rows]
            const int sh = i / TVR_IMG_BASH_ROWS, row = i - sh * TVR_IMG_BASH_ROWS;
            ((uint4 *)(smem + BASL))[i] = ((const uint4 *)sc.basis_frag)[sh * 32 + row];
        }
        __syncthreads();
    }
#ifdef TVR_DEBUG_SIMD
    // debug build (tests/test_gpu_faults.py): the token pairs waves w and w + 4, assuming they sit on one SIMD.  HW_REG_HW_ID bits [5:4] = SIMD id.
    // Every wave publishes its id through the (not yet used) q_out tail slot of stats: stats[16 + wave] of workgroup 0, and every workgroup
    // counts its mismatching pairs into stats[15].
    {
        __shared__ int simd_of[SH_WAVES];
        const int hw = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) /* HW_REG_HW_ID (4), offset 4, size 2: simm16 = id | offset << 6 | (size - 1) << 11 */;
        if (lane == 0) simd_of[wave] = hw;
        __syncthreads();
        if (a.stats && tid < SH_WAVES / 2 && simd_of[tid] != simd_of[tid + SH_WAVES / 2]) atomicAdd((unsigned long long *)&a.stats[15], 1ull);
        if (a.stats && blockIdx.x == 0 && tid < SH_WAVES) a.stats[16 + tid] = (unsigned long long)simd_of[tid];
        __syncthreads();
    }
#endif
#if TVR_MTOKEN
    bool have_tok = false;
    int *mtok = (int *)(smem + IMG_END) + (wave & 3);
#endif
    // basis hi fragment of lane (e, h) at k-step s: rows 27..31 of the 32-row tile do not exist (their outputs are never used): clamp
    const unsigned char *bashp = smem + BASH + (h * TVR_IMG_BASH_ROWS + (e < TVR_IMG_BASH_ROWS ? e : TVR_IMG_BASH_ROWS - 1)) * 16;
    const unsigned char *baslp = bashp + (BASL - BASH);  // lo parts, same addressing, k-steps 0 .. NLO-1
    // SRC_QUEUE: the count lives on the device.  SRC_H (training forward): optionally too — a.n is then the capacity of the buffers
    long long n_total = (SRC == SH_SRC_QUEUE) ? (long long)(*a.counter) : a.n;
    if (SRC == SH_SRC_H && a.counter) n_total = (long long)(*a.counter) < a.n ? (long long)(*a.counter) : a.n;
    const long long n_tiles = (n_total + SH_TILE - 1) / SH_TILE;
    unsigned long long clk0 = 0ull, ref0 = 0ull;              // clock probe (stats only), as in the march kernel
    if (a.stats && SRC == SH_SRC_QUEUE && tid == 0) { clk0 = __builtin_amdgcn_s_memtime(); ref0 = __builtin_amdgcn_s_memrealtime(); }
#if TVR_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr bool HAVE_G = REF && SRC != SH_SRC_FEAT;
    const long long tile_stride = (long long)gridDim.x * SH_WAVES;
    // workgroups b, b + 8, ... share an XCD (and its 4 MB L2): give each XCD a contiguous eighth of every window of tiles, so that its L2
    // holds one stretch of the image row instead of all of it (speed only; correctness does not depend on the placement)
    const unsigned lblk = xcd_remap(blockIdx.x, gridDim.x);       // (measured neutral against the identity: 15.4 ms both ways)
    float4 qe_next = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned qray_next = 0;
    // Round 5: the render path's tiles are handed out in tickets of TVR_TICKET consecutive tiles (the first one static, later ones by an atomicAdd on word 16 of the
    // scratch header, zeroed per call with the queue counter beside it) — the XCDs do not run at one speed under these kernels and equal static shares leave the fast ones
    // idle at the end (tvr_shade16.hip, profiles/r05_shade_tail.txt).  Only tvr_render(_z) launches the queue-to-queue mode, so the header is there; GEN keeps the static
    // stride (its layer 1 has workgroup barriers: every wave makes the same passes).  Which wave shades an entry does not matter to the entry.
    // Launches with fewer than TVR_TICKET_MIN tiles per wave keep the static stride: a ticket is a 4-tile quantum (tvr_shade16.hip).
    constexpr bool TICKETS = TVR_TICKET > 0 && SRC == SH_SRC_QUEUE && DST == SH_DST_QUEUE && !GEN;
    constexpr int TKN = TICKETS ? TVR_TICKET : 1;
    const bool dyn = TICKETS && n_tiles >= (long long)gridDim.x * SH_WAVES * TVR_TICKET_MIN;
    unsigned *const tk = TICKETS ? const_cast<unsigned *>(a.counter) + 16 : nullptr;
    const long long tick0 = (long long)gridDim.x * SH_WAVES * TKN;
    const long long tile_first = ((long long)lblk * SH_WAVES + wave) * (dyn ? TKN : 1);
    unsigned tk_pending = 0;
    int tk_sub = 0;
    if constexpr (TICKETS) { if (dyn && lane == 0) tk_pending = atomicAdd(tk, (unsigned)TKN); }
    const int tk_len = dyn ? TKN : 0x7fffffff;
    const long long tk_step = dyn ? 1 : tile_stride;
    if (SRC == SH_SRC_QUEUE && n_total > 0) {
        const long long e0 = tile_first * SH_TILE + e;
        const long long le = e0 < n_total ? e0 : n_total - 1;
        qe_next = a.q_pos[le];
        qray_next = a.q_ray[le];
    }
#if TVR_PHASE_FREE
    constexpr bool PFREE = SRC == SH_SRC_QUEUE && !GEN && AR == 3;
    Taps Tpre;
    auto prefetch_next = [&]() {                                // taps of k-step 0 (plane 0, channels 8h..8h+7) of the entry in qe_next
        float fcn[3];
        const float pnn[3] = {qe_next.x, qe_next.y, qe_next.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) fcn[k] = unnorm(pnn[k], sc.gm1[k]);
        load_taps<false>(Tpre, sc.aplane[0], sc.aline[0], sc.grid[0], sc.grid[1], sc.grid[2], fcn[0], fcn[1], fcn[2], TVR_Q0(0, h));
    };
    if constexpr (PFREE) { if (n_total > 0) prefetch_next(); }
#endif
    // GEN: every wave of the workgroup makes the same number of passes (its layer 1 has workgroup barriers); a pass beyond the last tile works on dead lanes
    const long long tile_end = GEN ? ((n_tiles - (long long)lblk * SH_WAVES + tile_stride - 1) / tile_stride) * tile_stride + (long long)lblk * SH_WAVES + wave : n_tiles;
    long long tile_next = 0;
    for (long long tile = tile_first; tile < tile_end; tile = tile_next) {                // (the advance sits in the for statement: DST_FEAT leaves the body by `continue`)
        tile_next = tile + tile_stride;
        if constexpr (TICKETS) {
            if (++tk_sub < tk_len) tile_next = tile + tk_step;            // (static mode: one endless "ticket" whose tiles lie a grid stride apart)
            else {
                tile_next = tick0 + (long long)__builtin_amdgcn_readfirstlane(tk_pending);
                tk_sub = 0;
                if (lane == 0 && tile_next < n_tiles) tk_pending = atomicAdd(tk, (unsigned)TKN);
            }
        }
        const long long ent = tile * SH_TILE + e;
        const bool live = ent < n_total;
        float F[16];                               // base values: row c = acc_row(r, h) of the feature tile, column = entry
        float dir[3] = {0.f, 0.f, 0.f}, wq = 0.f, dotin = 0.f;
        float G[8];                                // REF: rows acc_row(r, h) of the second block (h=0: normal, tint, rgb_d, rho; h=1: normal)
        float rmax = 0.0f;                         // RC: running max |x| of this lane's fp16-split operands
#pragma unroll
        for (int r = 0; r < 8; ++r) G[r] = 0.f;
#if TVR_TIMING
        unsigned long long tg0 = 0, tgD = 0, tgF = 0, tg1 = 0, tgW = 0, tg2 = 0, tg3 = 0, tg4 = 0, tgL = 0;
#endif
        TVR_STAMP(tg0);
        TVR_ENTER_GATHER();
        // ---------------------------------------------------------------- GATHER phase: global loads + VALU + LDS, no MFMA ----
        TVR_STAMP(tgD);
        // queue entry: fetched one tile ahead (the loads are issued in the previous tile's gather phase and have landed by its phase
        // boundary), unconditionally (a dead lane of the last tile re-reads the last entry; nothing of it is stored)
        float4 qe = qe_next;
        unsigned qray = qray_next;
        if (SRC == SH_SRC_QUEUE) {
            const long long en = tile_next * SH_TILE + e;
            const long long le = en < n_total ? en : n_total - 1;
            qe_next = a.q_pos[le];
            qray_next = a.q_ray[le];
        }
        TVR_STAMP(tgF);
        if (SRC == SH_SRC_H) {
            // training forward: h [n,144] comes from tvr_app_h_forward; this lane's 8 channels of each k-step are 32 contiguous bytes
            Frag hf[9];
            float hvv[9][8];                           // the fragments' fp32 values: k-step 0 is split in the gather phase, 1..8 under the basis MFMAs
            uint4 bal[9];
            if (live) {                                  // view direction: [n,3] given, or that of the entry's ray (the fused training step)
                const float *dp = a.q_ray ? a.rays + (size_t)a.q_ray[ent] * 6 + 3 : a.viewdirs + ent * 3;
                dir[0] = dp[0]; dir[1] = dp[1]; dir[2] = dp[2];
            }
            {
                const long long le = live ? ent : n_total - 1;
                const float4 *hp = (const float4 *)(a.h_in + le * TVR_KAPP) + 2 * h;
                float4 hv4[9][2];
#pragma unroll
                for (int s = 0; s < 9; ++s) { hv4[s][0] = hp[4 * s]; hv4[s][1] = hp[4 * s + 1]; }
                unsigned boff = (unsigned)((h * 32 + e) * 16);
                asm volatile("" : "+v"(boff));
#pragma unroll
                for (int s3 = NLO; s3 < 9; ++s3) if (ARB >= 2) bal[s3] = *(const uint4 *)((const unsigned char *)sc.basis_frag + (boff + (unsigned)(s3 * 1024)));
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    const float hv[8] = {hv4[s][0].x, hv4[s][0].y, hv4[s][0].z, hv4[s][0].w, hv4[s][1].x, hv4[s][1].y, hv4[s][1].z, hv4[s][1].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) hvv[s][j] = hv[j];
                    TVR_GATHER_KEEP(s);
                }
            }
            TVR_BASIS_BLOCK();
        } else if (SRC != SH_SRC_FEAT) {
            float pn[3] = {0.f, 0.f, 0.f};
            if (SRC == SH_SRC_QUEUE) {
                pn[0] = qe.x; pn[1] = qe.y; pn[2] = qe.z; wq = qe.w;
                const float *rp = a.rays + (size_t)qray * 6 + 3;
                dir[0] = rp[0]; dir[1] = rp[1]; dir[2] = rp[2];
            } else if (live) {
                pn[0] = a.xyz[ent * 3]; pn[1] = a.xyz[ent * 3 + 1]; pn[2] = a.xyz[ent * 3 + 2];
            }
            float fc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = unnorm(pn[k], sc.gm1[k]);
            // 9 k-steps of 16 channels (plane p = s/3, channels 16(s%3) + 8h .. +7 of this lane); the B fragments (plane*line products,
            // fp16 hi/lo) stay in registers
            Frag hf[9];
            float hvv[9][8];                           // the fragments' fp32 values: k-step 0 is split in the gather phase, 1..8 under the basis MFMAs
            uint4 bal[9];
            {
                // ring: before k-step s is evaluated, the taps of k-steps up to s + depth(s) are in flight.  depth = TVR_PF (2) while the finished
                // fragments hf[] are few, 1 from k-step TVR_PF_SHALLOW on: registers = 8 s (hf) + 48 per tap set, and a wave has 256.  Round 3 ran depth 1
                // throughout: a tap set then has one evaluation (~80 VALU, ~400 cycles) to arrive, an L2 / MALL hit takes 500-900 cycles, and the phase
                // timing of a lone wave showed half of the gather phase's 8.7 k cycles to be that wait.
                constexpr int PFD = TVR_PF;
                constexpr bool H16 = (AR == 1) && (SRC == SH_SRC_QUEUE);        // the one-product arithmetic gathers the fp16 images
                Taps T[H16 ? 1 : PFD + 1];
                Taps16 T16[H16 ? PFD + 1 : 1];
                auto tgt = [](int s) { const int d = s < TVR_PF_SHALLOW ? TVR_PF : 1; return s + d < 8 ? s + d : 8; };   // last k-step issued before k-step s is evaluated
                auto issue = [&](int s2) {
                    const int p = s2 / 3;
                    const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;      // matMode / vecMode
                    if constexpr (H16) load_taps16(T16[s2 % (PFD + 1)], sc.aplane16[p], sc.aline16[p], sc.grid[ax], fc[ax], fc[bx], fc[vx], 2 * (s2 % 3) + h);
                    else load_taps<TVR_CHK>(T[s2 % (PFD + 1)], sc.aplane[p], sc.aline[p], sc.grid[ax], sc.grid[bx], sc.grid[vx], fc[ax], fc[bx], fc[vx], TVR_Q0(s2 % 3, h));
                };
#if TVR_PHASE_FREE
                if constexpr (PFREE) T[0] = Tpre;                // issued in the previous tile's matrix phase (or in front of the loop)
#endif
#pragma unroll
                for (int s = 0; s < 9; ++s) {
#pragma unroll
                    for (int s2 = (s == 0 ? 0 : tgt(s - 1) + 1); s2 <= tgt(s); ++s2) {
#if TVR_PHASE_FREE
                        if (PFREE && s2 == 0) continue;
#endif
                        issue(s2);
                    }
                    if (!REF && s == 7) {                              // (REFTensoRF: behind the loop — its extra live values leave no room earlier)
                        // the basis A fragments (the tile's last global loads) ride behind the last taps
                        // (lo parts; the hi parts are in LDS.  Byte offsets against the uniform base, opaque per tile: hoisted per-step 64-bit
                        // addresses would spill)
                        unsigned boff = (unsigned)((h * 32 + e) * 16);
                        asm volatile("" : "+v"(boff));
#pragma unroll
                        for (int s3 = NLO; s3 < 9; ++s3) if (ARB >= 2) bal[s3] = *(const uint4 *)((const unsigned char *)sc.basis_frag + (boff + (unsigned)(s3 * 1024)));
                    }
                    const int p = s / 3;
                    const int ax = (p == 2) ? 1 : 0, bx = (p == 0) ? 1 : 2, vx = 2 - p;
                    if constexpr (H16) taps_eval16(T16[s % (PFD + 1)], fc[ax], fc[bx], fc[vx], hvv[s]);
                    else taps_eval<TVR_CHK>(T[s % (PFD + 1)], sc.grid[ax], sc.grid[bx], sc.grid[vx], fc[ax], fc[bx], fc[vx], hvv[s]);
                    TVR_GATHER_KEEP(s);
                    TVR_SB;
                }
            }
            if (REF) {
                unsigned boff = (unsigned)((h * 32 + e) * 16);
                asm volatile("" : "+v"(boff));
#pragma unroll
                for (int s3 = NLO; s3 < 9; ++s3) if (ARB >= 2) bal[s3] = *(const uint4 *)((const unsigned char *)sc.basis_frag + (boff + (unsigned)(s3 * 1024)));
            }
            TVR_BASIS_BLOCK();
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = acc_row(r, h);
                F[r] = (live && c < TVR_APPDIM) ? a.feats[ent * TVR_APPDIM + c] : 0.0f;
            }
            if (live) {
                dir[0] = a.viewdirs[ent * 3]; dir[1] = a.viewdirs[ent * 3 + 1]; dir[2] = a.viewdirs[ent * 3 + 2];
                if (REF) dotin = a.dots[ent];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // phase boundary: nothing of this tile's loads outlives the gather phase
            TVR_STAMP(tg1);
            TVR_ENTER_MATRIX();
            TVR_STAMP(tgW);
        }

        if (DST == SH_DST_FEAT) {
            TVR_LEAVE_MATRIX();
            if (RC) {                                  // (the features are fp32 outputs; what is checked is what went IN: the interpolated h)
                const float m = fmaxf(rmax, __shfl_xor(rmax, 32));
                if (!(m < TVR_F16_MAX)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) F[r] = __builtin_nanf("");
                }
            }
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = acc_row(r, h);
                    if (c < TVR_APPDIM) a.out[ent * TVR_APPDIM + c] = F[r];
                }
                if (REF && h == 0 && a.out2) {       // REFTensoRF.compute_appfeature :126-133 {normal, rgb_d, relu(tint), relu(rho)}
                    float *o = a.out2 + ent * 8;
                    o[0] = G[0]; o[1] = G[1]; o[2] = G[2];
                    o[3] = G[4]; o[4] = G[5]; o[5] = G[6];
                    o[6] = fmaxf(G[3], 0.0f); o[7] = fmaxf(G[7], 0.0f);
                }
            }
            continue;
        }

        TVR_STAMP(tg2);
        if (DST == SH_DST_TRAIN && !REF && live) {   // features [n,32]: rows 27..31 (h=0: r=15; h=1: r=12..15) are written as zero
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4 *)(a.t_feats + ent * 32 + 8 * q + 4 * h) = q < 3 ? make_float4(F[4 * q], F[4 * q + 1], F[4 * q + 2], F[4 * q + 3])
                                                                    : (h == 0 ? make_float4(F[12], F[13], F[14], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        // view direction occupies base rows 27 (h=0, r=15), 28, 29 (h=1, r=12, 13); row 30 = REF's -dot, row 31 = the constant 1 (b1's column)
        if (REF && SRC != SH_SRC_FEAT) {
            // REFTensoRF.execute :215-227: normalise the normal, d = -view, dot = d.n, reflection = 2 dot n - d; the MLP
            // takes the reflection as its direction and -dot as input 0 (row 30: only its t=0 slot has a weight)
            const float nrm = sqrtf(fmaxf((G[0] * G[0] + G[1] * G[1]) + G[2] * G[2], 1e-30f));
            const float nx = G[0] / nrm, ny = G[1] / nrm, nz = G[2] / nrm;
            const float dx = -dir[0], dy = -dir[1], dz = -dir[2];
            const float dot = (dx * nx + dy * ny) + dz * nz;
            dir[0] = 2.0f * dot * nx - dx; dir[1] = 2.0f * dot * ny - dy; dir[2] = 2.0f * dot * nz - dz;
            dotin = -dot;
        }
        if (h == 0) F[15] = dir[0];
        else { F[12] = dir[1]; F[13] = dir[2]; F[14] = dotin; F[15] = 1.0f; }
        if (RC) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) rmax = absmax2(F[r], F[r + 1], rmax);
            asm volatile("" : "+v"(rmax));             // (pure arithmetic: without a pin hipcc sinks the whole max chain, and the values it reads, to the tile's end)
        }
        if (DST == SH_DST_TRAIN && REF && live) {
            // REFTensoRF training forward: the 31 base values of layer 1 — 27 features, the reflection direction (rows 27..29) and -dot (row 30),
            // both functions of h through the normal head — are what the backward differentiates through; row 31 (the constant) is stored as 0.
            // And the raw outputs of the four heads {normal 3, tint, rgb_d 3, rho} (h = 0 lanes hold all eight).
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4 *)(a.t_feats + ent * 32 + 8 * q + 4 * h) = (q < 3 || h == 0) ? make_float4(F[4 * q], F[4 * q + 1], F[4 * q + 2], F[4 * q + 3])
                                                                                      : make_float4(F[12], F[13], F[14], 0.f);
            if (h == 0) {
                *(float4 *)(a.t_g8 + ent * 8) = make_float4(G[0], G[1], G[2], G[3]);
                *(float4 *)(a.t_g8 + ent * 8 + 4) = make_float4(G[4], G[5], G[6], G[7]);
            }
        }

        // ---- layers 1 and 2 as ONE software pipeline (see "round 4: the matrix phase as an explicit software pipeline" above) ----
        // layer 1: 10 k-steps; slot i = 8s + j of this lane is derived value (i % 5) of base value i / 5; sin / cos of a base value are taken in the step
        // that first needs them (2-3 per step).  Layer 2: the B fragments are the relu'd layer-1 accumulators, 8 registers per k-step; b2 is the initial
        // accumulator.  Layer 1's last k-step already carries layer 2's prologue: once row block 0's last MFMA has issued, b2 and W2's first fragments are
        // read and relu(acc[0]) is split under the MFMAs of row blocks 1..3 (round 3 paid an LDS round trip and 24 VALU ops there with the pipe idle).
        f32x16 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x16{0};
        Carry cur;
        {
            float S1[16], C1[16];
            const int rowoff = (h * 128 + e) * 16;
            auto l1_frag = [&](int s, Frag &b) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * s + j, r = i / 5, t = i % 5;
                    // first slot of base value r in program order is i = 5r (t = 0): take its sin / cos there
                    if (t == 0) sincos_pe(F[r], S1[r], C1[r]);
                    v[j] = t == 0 ? F[r] : (t == 1 ? S1[r] : (t == 2 ? 2.0f * S1[r] * C1[r]                 // sin 2v
                                  : (t == 3 ? C1[r] : __builtin_fmaf(-2.0f * S1[r], S1[r], 1.0f))));          // cos 2v
                }
                b = frag8<AR>(v);
            };
            auto relu_frag = [&](int s, Frag &b) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu_f(acc[s >> 1][8 * (s & 1) + j]);
                if (RC) { _Pragma("unroll") for (int j = 0; j < 8; j += 2) rmax = fmaxf(fmaxf(v[j], v[j + 1]), rmax); asm volatile("" : "+v"(rmax)); }
                if (DST == SH_DST_TRAIN && live) {   // relu(layer 1): element j is hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3)
                    *(float4 *)(a.t_h1 + ent * TVR_FEATC + 16 * s + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4 *)(a.t_h1 + ent * TVR_FEATC + 16 * s + 8 + 4 * h) = make_float4(v[4], v[5], v[6], v[7]);
                }
                b = frag8<AR>(v);
            };
            Frag fr[8];                                         // relu(layer 1) as layer 2's eight B fragments (the 64 registers `acc` leaves)
            f32x16 a2cur, a2nxt;                                // layer-2 accumulators of the row block in flight / its successor's b2
            auto b2_init = [&](int rb2, f32x16 &dst) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 bv = *(const float4 *)(smem + TVR_IMG_B2 + (32 * rb2 + 8 * q4 + 4 * h) * 4);
                    dst[4 * q4] = bv.x; dst[4 * q4 + 1] = bv.y; dst[4 * q4 + 2] = bv.z; dst[4 * q4 + 3] = bv.w;
                }
            };
            // ONE opaque address register per image: every fragment read is base + 16-bit immediate (left to itself hipcc materialises an address
            // register per read, and those VALU adds take the slots the pipeline means for the fragment derivation)
            TVR_LDS_BASE(W1Hb, smem + TVR_IMG_W1H + rowoff);
            TVR_LDS_BASE(W1Lb, smem + TVR_IMG_W1L + rowoff);
            TVR_LDS_BASE(W2Hb, smem + TVR_IMG_W2H + rowoff);        // W2's lo image starts 32 KB behind its hi image: one base, immediates < 64 KB
            const unsigned char *W2Lb = W2Hb + (TVR_IMG_W2L - TVR_IMG_W2H);
            Frag bcur, bnxt;
            AF ring[TVR_RN], ring2[TVR_RN];
            if constexpr (GEN) {
                // ---- layer 1 with up to six encoding frequencies (TVR_GEN_*, tvr_device.h): 26 k-steps in LOCKSTEP.  Slot i = 8s + j of this lane is derived value
                // i % 13 of base value i / 13: v, sin(2^f v) f < 6, cos(2^f v) f < 6 — v_sin / v_cos of the once-reduced argument (revolutions) times 2^f, exact
                // scalings.  One k-step of W1's fragments (8 KB: hi | lo) sits in an LDS slot; every thread fetches one uint4 of the NEXT k-step at the top of a
                // step and stores it into the other slot at the bottom, in front of the step's barrier.  (The fetch is a global load between MFMAs: this
                // instantiation is outside the phase rule of the two-frequency kernels — scripts/isa_check.py exempts it by name; the rule's probes came back
                // clean in round 2, and this is the general path, not the measured one.)
                float TR[16];
                auto gen_frag = [&](int s, Frag &b) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * s + j, r = i / TVR_GEN_T, t = i % TVR_GEN_T;
                        if (t == 0) {                                   // Cody-Waite reduction as in sincos_pe; the remainder in revolutions, |TR| <= 0.5
                            const float k = rintf(F[r] * 0.15915494309189535f);
                            float rr = __builtin_fmaf(k, -6.2831854820251465f, F[r]);
                            rr = __builtin_fmaf(k, 1.7484555e-7f, rr);
                            TR[r] = rr * 0.15915494309189535f;
                        }
                        v[j] = t == 0 ? F[r] : (t <= TVR_GEN_PE ? __builtin_amdgcn_sinf(TR[r] * (float)(1 << (t > 0 && t <= TVR_GEN_PE ? t - 1 : 0)))
                                                                : __builtin_amdgcn_cosf(TR[r] * (float)(1 << (t > TVR_GEN_PE ? t - 1 - TVR_GEN_PE : 0))));
                    }
                    b = split8(v);
                };
                const uint4 *gsrc = (const uint4 *)sc.w1gen + tid;                       // k-step s: uint4 s * 512 + tid of the image (hi 256 | lo 256 uint4)
                uint4 *slot = (uint4 *)(smem + TVR_IMG_W1H) + tid;                        // THREE 8 KB staging slots at the head of the (otherwise unused) W1 region
                // Round 6 — what a k-step of round 4's form cost and why (scripts/phase_timing.py with TVR_PE=6: 1 455 cycles for the 768 cycles of its two waves' MFMAs):
                // every wave reads the whole 8 KB slot (64 KB per k-step and CU: ~512 cycles of the LDS) and, in lockstep, every wave did so right behind the barrier and
                // multiplied afterwards — reads and MFMAs in series.  Not the barrier count (groups of 2 / 4 / 5 k-steps per barrier: slower), not the order of the fragment
                // derivation (pinned under the MFMAs: equal), not the fetch distance (1 .. 6 k-steps ahead: equal).  Now the image is staged TWO k-steps ahead (three slots), so
                // that the fragments of row blocks 2, 3 are read under the MFMAs of row blocks 0, 1 and the NEXT k-step's fragments of row blocks 0, 1 under those of row blocks
                // 2, 3.  Per accumulator the products come in mfma3x4's order (lo*hi, hi*lo, hi*hi): bit-identical results.
                constexpr int PFG = TVR_GEN_PF;
                uint4 wq[PFG];                                                            // wq[i]: k-step s + 2 + i, in flight
                slot[0] = gsrc[0];                                                       // (the last readers of these slots passed a barrier since)
                slot[512] = gsrc[512];
#pragma unroll
                for (int i = 0; i < PFG; ++i) wq[i] = (2 + i < TVR_GEN_KS) ? gsrc[(2 + i) * 512] : make_uint4(0u, 0u, 0u, 0u);
                gen_frag(0, bcur);
                __syncthreads();
                uint4 Ah[4], Al[4];
                auto rd2 = [&](int s_, int rb0) {                                        // the fragments of row blocks rb0, rb0 + 1 of k-step s_
                    const unsigned char *sb = smem + TVR_IMG_W1H + (s_ % 3) * 8192 + rowoff;
#pragma unroll
                    for (int rb = rb0; rb < rb0 + 2; ++rb) { Ah[rb] = *(const uint4 *)(sb + rb * TVR_IMG_RB); Al[rb] = *(const uint4 *)(sb + 4096 + rb * TVR_IMG_RB); }
                };
                auto mm2 = [&](int rb0) {
#pragma unroll
                    for (int rb = rb0; rb < rb0 + 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, Al[rb]), __builtin_bit_cast(h8, bcur.hi), acc[rb], 0, 0, 0);
#pragma unroll
                    for (int rb = rb0; rb < rb0 + 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, Ah[rb]), __builtin_bit_cast(h8, bcur.lo), acc[rb], 0, 0, 0);
#pragma unroll
                    for (int rb = rb0; rb < rb0 + 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, Ah[rb]), __builtin_bit_cast(h8, bcur.hi), acc[rb], 0, 0, 0);
                };
                rd2(0, 0);
#pragma unroll
                for (int s = 0; s < TVR_GEN_KS; ++s) {
                    const uint4 wn = wq[0];                                              // k-step s + 2: requested PFG steps ago
#pragma unroll
                    for (int i = 0; i + 1 < PFG; ++i) wq[i] = wq[i + 1];
                    wq[PFG - 1] = (s + 2 + PFG < TVR_GEN_KS) ? gsrc[(s + 2 + PFG) * 512] : make_uint4(0u, 0u, 0u, 0u);
                    rd2(s, 2);
                    if (s + 1 < TVR_GEN_KS) gen_frag(s + 1, bnxt);
                    mm2(0);
                    TVR_SB;
                    if (s + 1 < TVR_GEN_KS) rd2(s + 1, 0);                               // (slot (s + 1) % 3 was complete at the last barrier)
                    mm2(2);
                    if (s + 2 < TVR_GEN_KS) slot[((s + 2) % 3) * 512] = wn;               // (slot (s + 2) % 3 = (s - 1) % 3: its last readers passed the last barrier)
                    if (s + 1 < TVR_GEN_KS) bcur = bnxt;
                    __syncthreads();
                }
                // layer 2's prologue: b2 -> the initial accumulators, W2's first fragment pairs, relu(layer 1) of k-step 0
#pragma unroll
                for (int q0 = 0; q0 < TVR_PD; ++q0) load_af(ring2[q0], W2Hb, W2Lb, TVR_AOFF(4 * q0));
                b2_init(0, a2cur);
                relu_frag(0, fr[0]);
                relu_frag(1, fr[1]);
                TVR_SB;
            } else {
#pragma unroll
            for (int q0 = 0; q0 < TVR_PD; ++q0) load_af<AR>(ring[q0], W1Hb, W1Lb, TVR_AOFF(q0));
            l1_frag(0, bcur);
            TVR_SB;
#pragma unroll
            for (int s = 0; s < 10; ++s) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const int q = 4 * s + rb;
                    if (q + TVR_PD < 40 && !(TVR_DIAG & 2)) load_af<AR>(ring[(q + TVR_PD) % TVR_RN], W1Hb, W1Lb, TVR_AOFF((TVR_DIAG & 8) ? 0 : ((TVR_DIAG & 4) ? ((q + TVR_PD) & 3) : q + TVR_PD)));
                    mfma3<AR>(ring[(TVR_DIAG & 2) ? (q & 1) : (q % TVR_RN)], bcur, acc[rb]);
                    if (s == 9 && rb == 0) {
                        // layer 2's prologue (acc[0] is complete): b2 -> the initial accumulators, W2's first two fragment pairs, relu(acc[0]) split
#pragma unroll
                        for (int q0 = 0; q0 < TVR_PD; ++q0) load_af<AR>(ring2[q0], W2Hb, W2Lb, TVR_AOFF(4 * q0));
                        b2_init(0, a2cur);
                    }
                }
                if (TVR_DIAG & 1) bnxt = bcur;
                else if (s + 1 < 10) l1_frag(s + 1, bnxt);
                else { relu_frag(0, fr[0]); relu_frag(1, fr[1]); }         // acc[0] holds both k-steps' 16 hidden units
                if (s < 9) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        if constexpr (AR == 3) TVR_PIPE_RB(4 * s + rb + TVR_PD < 40, TVR_L1_V0, TVR_L1_V1, TVR_L1_V2);
                        else if constexpr (AR == 2) TVR_PIPE_RB2(4 * s + rb + TVR_PD < 40, TVR_L1A2_V0, TVR_L1A2_V1);      // (the fragment is 4 cvt_pk instead of 16 split ops)
                        else TVR_PIPE_RB1(4 * s + rb + TVR_PD < 40, TVR_L1A1_V0);
                    }
                } else {
                    TVR_SG_MFMA(AR);                // row block 0
                    TVR_SG_DSR(4 + (AR >= 2 ? 2 : 1) * TVR_PD);     // W2's first fragments, b2 of row block 0
                    TVR_SG_MFMA(AR);                // row block 1: relu(acc[0]) may be read 3 MFMAs after its last write
#pragma unroll
                    for (int m = 0; m < 2 * AR; ++m) { TVR_SG_MFMA(1); if constexpr (AR == 3) TVR_SG_VALU(5); else if constexpr (AR == 2) TVR_SG_VALU(6); else TVR_SG_VALU(12); }
                }
                bcur = bnxt;
                TVR_SB;
            }
            }           // !GEN
#if TVR_PHASE_FREE
            if constexpr (PFREE) { TVR_SB; prefetch_next(); TVR_SB; }      // 12 global loads per lane between the MFMAs of layer 1 and those of layer 2
#endif
            TVR_STAMP(tg3);
            {
                f32x16 a2prev = f32x16{0};                          // the finished row block layer 3 is working through
                f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                TVR_LDS_BASE(W3b, smem + TVR_IMG_W3 + 16 * h);
                const float *W3 = (const float *)W3b;
                float4 w3[3];
                // layer 3 over the 4 hidden units 32 rb3 + 8 qq + 4 h .. + 3 of this lane (tensorBase.py:83-84: Linear(128 -> 3) on relu(h2)); the same order of
                // additions as finish_tile's
                auto l3_quad = [&](const f32x16 &x, int rb3, int qq) {
                    const f32x2 xa = {relu_f(x[4 * qq]), relu_f(x[4 * qq + 1])};
                    const f32x2 xb = {relu_f(x[4 * qq + 2]), relu_f(x[4 * qq + 3])};
                    if (DST == SH_DST_TRAIN && live)
                        *(float4 *)(a.t_h2 + ent * TVR_FEATC + 32 * rb3 + 8 * qq + 4 * h) = make_float4(xa.x, xa.y, xb.x, xb.y);
                    s0 = pk_fma(xa, f32x2{w3[0].x, w3[0].y}, s0); s0 = pk_fma(xb, f32x2{w3[0].z, w3[0].w}, s0);
                    s1 = pk_fma(xa, f32x2{w3[1].x, w3[1].y}, s1); s1 = pk_fma(xb, f32x2{w3[1].z, w3[1].w}, s1);
                    s2 = pk_fma(xa, f32x2{w3[2].x, w3[2].y}, s2); s2 = pk_fma(xb, f32x2{w3[2].z, w3[2].w}, s2);
                };
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int q = 8 * rb + ks;                  // position in the (row block, k-step) sequence; its fragment: TVR_AOFF(4 ks + rb)
                        if (q + TVR_PD < 32) load_af<AR>(ring2[(q + TVR_PD) % TVR_RN], W2Hb, W2Lb, TVR_AOFF(4 * ((q + TVR_PD) & 7) + ((q + TVR_PD) >> 3)));
                        if (ks == 6 && rb < 3) b2_init(rb + 1, a2nxt);
                        if (rb > 0 && !(ks & 1)) {
#pragma unroll
                            for (int c3 = 0; c3 < 3; ++c3) w3[c3] = *(const float4 *)(W3 + c3 * 128 + 32 * (rb - 1) + 8 * (ks >> 1));
                        }
                        mfma3<AR>(ring2[q % TVR_RN], fr[ks], a2cur);
                        if (rb == 0) { if (ks + 2 < 8) relu_frag(ks + 2, fr[ks + 2]); }
                        else if (ks & 1) l3_quad(a2prev, rb - 1, ks >> 1);
                        if (q + TVR_PD < 32) TVR_SG_DSR(AR >= 2 ? 2 : 1);
                        if (ks == 6 && rb < 3) TVR_SG_DSR(4);
                        if (rb > 0 && !(ks & 1)) TVR_SG_DSR(3);
                        if constexpr (AR == 3) {
                            if (rb == 0 && ks + 2 < 8) { TVR_SG_MFMA(1); TVR_SG_VALU(8); TVR_SG_MFMA(1); TVR_SG_VALU(8); TVR_SG_MFMA(1); TVR_SG_VALU(8); }
                            else if (rb > 0 && (ks & 1)) { TVR_SG_MFMA(1); TVR_SG_VALU(5); TVR_SG_MFMA(1); TVR_SG_VALU(5); TVR_SG_MFMA(1); TVR_SG_VALU(6); }
                            else TVR_SG_MFMA(3);
                        } else if constexpr (AR == 2) {       // relu_frag is 8 max + 4 cvt_pk here; layer 3's quad 4 max + 12 FMA as ever
                            if (rb == 0 && ks + 2 < 8) { TVR_SG_MFMA(1); TVR_SG_VALU(6); TVR_SG_MFMA(1); TVR_SG_VALU(6); }
                            else if (rb > 0 && (ks & 1)) { TVR_SG_MFMA(1); TVR_SG_VALU(8); TVR_SG_MFMA(1); TVR_SG_VALU(8); }
                            else TVR_SG_MFMA(2);
                        } else {
                            if (rb == 0 && ks + 2 < 8) { TVR_SG_MFMA(1); TVR_SG_VALU(12); }
                            else if (rb > 0 && (ks & 1)) { TVR_SG_MFMA(1); TVR_SG_VALU(16); }
                            else TVR_SG_MFMA(1);
                        }
                        TVR_SB;
                    }
                    a2prev = a2cur;
                    if (rb < 3) a2cur = a2nxt;
                }
                cur.acc2 = a2prev;
                cur.s0 = s0; cur.s1 = s1; cur.s2 = s2;
            }
        }
        TVR_STAMP(tgL);
#if TVR_DIAG & 1
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) cur.acc2 += acc[rb];              // keeps layer 1 alive in the build that derives no fragments
#endif
        cur.ent = ent; cur.live = live; cur.wq = wq;
        cur.g[0] = G[3]; cur.g[1] = G[4]; cur.g[2] = G[5]; cur.g[3] = G[6];
        TVR_LEAVE_MATRIX();
        // layer 3 + store run at the LOWEST priority: their 320 VALU ops need no particular moment, the partner's matrix phase needs every issue slot it
        // can get (12.66 vs 12.75 ms against running them at the gather's priority, two interleaved rounds; priorities 1 and 3: 12.73 / 12.75)
        __builtin_amdgcn_s_setprio(TVR_PRIO_F);
        cur.rmax = rmax;
        finish_tile<DST, REF, HAVE_G, RC>(cur, smem, a, h);
        __builtin_amdgcn_s_setprio(TVR_PRIO_G);
        TVR_STAMP(tg4);
#if TVR_TIMING
        tsum[0] += tgF - tgD; tsum[5] += tgD - tg0; tsum[1] += tg1 - tgF; tsum[6] += tgW - tg1; tsum[2] += tg2 - tgW; tsum[3] += tg3 - tg2; tsum[4] += tgL - tg3; tsum[7] += tg4 - tgL;
#endif
    }
#if TVR_TIMING
    if (a.stats && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long *)&a.stats[8 + i], tsum[i]);     // queue fetch, gather, basis, L1 (+PE), L2, -, wait for the matrix token, hand-over + layer 3 + store
#endif
    if (a.stats && SRC == SH_SRC_QUEUE && tid == 0) {
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_SHADE_CLK], __builtin_amdgcn_s_memtime() - clk0);
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_SHADE_REF], __builtin_amdgcn_s_memrealtime() - ref0);
    }
    if (a.stats && SRC == SH_SRC_QUEUE && blockIdx.x == 0 && tid == 0)
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_APP], (unsigned long long)n_total);
}

template <int SRC, int DST, bool REF, bool RC, bool GEN = false, int AR = 3>
static hipError_t launch_shade_t(const SceneDev &sc, const ShadeArgs &a, hipStream_t stream)
{
    constexpr bool BAS_ONLY = DST == SH_DST_FEAT && !REF;
    constexpr int NLO = REF ? 0 : (BAS_ONLY ? 9 : TVR_NLO_LDS);
    const int lds = (REF ? TVR_MLP_IMAGE_BYTES_REF : (BAS_ONLY ? (TVR_MLP_IMAGE_BYTES - TVR_IMG_BASH) : TVR_MLP_IMAGE_BYTES)) + 16 + NLO * 2 * TVR_IMG_BASH_ROWS * 16;
    static_assert((REF ? TVR_MLP_IMAGE_BYTES_REF : TVR_MLP_IMAGE_BYTES) + 16 + (REF ? 0 : TVR_NLO_LDS) * 2 * TVR_IMG_BASH_ROWS * 16 <= 160 * 1024, "LDS image + tokens + basis lo parts must fit 160 KB");
    hipError_t rc = hipFuncSetAttribute((const void *)shade_kernel<SRC, DST, REF, RC, GEN, AR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (rc != hipSuccess) return rc;
    unsigned grid = 256;       // one workgroup per CU (LDS holds the MLP weights), persistent over SH_TILE-entry tiles
#ifdef TVR_EXP_GRID                                             // scripts/overlap_experiment.py only: a build_variant.sh -DTVR_EXP_GRID library
    if (const char *g = getenv("TVR_EXP_GRID_SHADE")) { const long long v = atoll(g); if (v > 0 && v < 256) grid = (unsigned)v; }
#endif
    if (SRC != SH_SRC_QUEUE) {
        const long long groups = (a.n + SH_TILE * SH_WAVES - 1) / (SH_TILE * SH_WAVES);
        if (groups < grid) grid = (unsigned)(groups > 0 ? groups : 1);
    }
    hipLaunchKernelGGL((shade_kernel<SRC, DST, REF, RC, GEN, AR>), dim3(grid), dim3(SH_THREADS), lds, stream, sc, a);
    return hipGetLastError();
}

// the render / mlp_render kernels in the scene's arithmetic (tvr_scene_set_arith); every other path computes with three products whatever the mode
template <int SRC, int DST, bool REF, bool RC>
static hipError_t launch_shade_ar(const SceneDev &sc, const ShadeArgs &a, hipStream_t stream)
{
    if (sc.arith == TVR_ARITH_F16ACT) return launch_shade_t<SRC, DST, REF, RC, false, 2>(sc, a, stream);
    if (sc.arith == TVR_ARITH_F16) return launch_shade_t<SRC, DST, REF, RC, false, 1>(sc, a, stream);
    return launch_shade_t<SRC, DST, REF, RC, false, 3>(sc, a, stream);
}

template <bool REF>
static hipError_t launch_shade_v(const SceneDev &sc, int src, int dst, const ShadeArgs &a, hipStream_t stream)
{
    const bool rc = sc.range_check != 0;        // the inference entry points; the training forward has its own saturation flag (tvr_mlp_train.hip)
    if constexpr (!REF) {
        if (sc.gen) {                              // more than two encoding frequencies: the lockstep layer 1 (TensorVMSplit scenes only; check_desc refuses the rest)
            if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE) return rc ? launch_shade_t<SH_SRC_QUEUE, SH_DST_QUEUE, false, true, true>(sc, a, stream) : launch_shade_t<SH_SRC_QUEUE, SH_DST_QUEUE, false, false, true>(sc, a, stream);
            if (src == SH_SRC_FEAT && dst == SH_DST_RGB) return rc ? launch_shade_t<SH_SRC_FEAT, SH_DST_RGB, false, true, true>(sc, a, stream) : launch_shade_t<SH_SRC_FEAT, SH_DST_RGB, false, false, true>(sc, a, stream);
            if (src == SH_SRC_H && dst == SH_DST_TRAIN) return launch_shade_t<SH_SRC_H, SH_DST_TRAIN, false, false, true>(sc, a, stream);     // round 6: the fused training step of such scenes
        }
    }
#ifndef TVR_SHADE16
#define TVR_SHADE16 1      // 0: A/B builds — the render path stays on this file's 32x32x16 kernel
#endif
#if TVR_SHADE16
    // round 5: the render path of TensorVMSplit scenes in the default arithmetic runs on 16x16x32 tiles (tvr_shade16.hip); every other mode stays here
    // round 6: REFTensoRF's render path too (TVR_SHADE16_REF=0: A/B builds keep it on this file's kernel)
#ifndef TVR_SHADE16_REF
#define TVR_SHADE16_REF 1
#endif
    if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE && sc.arith == TVR_ARITH_F32 && sc.img16 && (!REF || (TVR_SHADE16_REF && sc.refg16))) return launch_shade16(sc, a, stream);
#endif
    if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE) return rc ? launch_shade_ar<SH_SRC_QUEUE, SH_DST_QUEUE, REF, true>(sc, a, stream) : launch_shade_ar<SH_SRC_QUEUE, SH_DST_QUEUE, REF, false>(sc, a, stream);
    if (src == SH_SRC_XYZ && dst == SH_DST_FEAT) return rc ? launch_shade_t<SH_SRC_XYZ, SH_DST_FEAT, REF, true>(sc, a, stream) : launch_shade_t<SH_SRC_XYZ, SH_DST_FEAT, REF, false>(sc, a, stream);
    if (src == SH_SRC_FEAT && dst == SH_DST_RGB) return rc ? launch_shade_ar<SH_SRC_FEAT, SH_DST_RGB, REF, true>(sc, a, stream) : launch_shade_ar<SH_SRC_FEAT, SH_DST_RGB, REF, false>(sc, a, stream);
    if (src == SH_SRC_H && dst == SH_DST_TRAIN) return launch_shade_t<SH_SRC_H, SH_DST_TRAIN, REF, false>(sc, a, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_shade(const SceneDev &sc, int src, int dst, const ShadeArgs &a, hipStream_t stream)
{
    return sc.variant == 1 ? launch_shade_v<true>(sc, src, dst, a, stream) : launch_shade_v<false>(sc, src, dst, a, stream);
}

// ---- scene packing (reference layout -> channels-last, zero-padded) ----
// in (Cin,H,W) -> out [H+1][W+1][C]; a line (W == 1) packs to [H+1][C].  Cin <= C: a scene with fewer components than the kernels are built
// for (TensorBase's defaults are 8 / 24) is packed with zero channels, which contribute exact zeros to every sum
__global__ __launch_bounds__(256) void pack_plane_kernel(const float *__restrict__ in, float *__restrict__ out, int Cin, int C, int H, int W, int Wp)
{
    const long long total = (long long)(H + 1) * Wp * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long t = i / C;
        const int x = (int)(t % Wp), y = (int)(t / Wp);
        out[i] = (x < W && y < H && c < Cin) ? in[((size_t)c * H + y) * W + x] : 0.0f;
    }
}

// planes: one workgroup per (row y, 64 columns): the reads run along x of each channel, the writes along the packed row — both coalesced through an LDS tile
// (the element-per-thread kernel above reads with a stride of H*W floats: 9.6 us per plane, 12 planes per training step)
#define PK_TILE 64
__global__ __launch_bounds__(256) void pack_plane_tiled_kernel(const float *__restrict__ in, float *__restrict__ out, int Cin, int C, int H, int W, int Wp)
{
    __shared__ float tile[PK_TILE][TVR_CA + 1];
    const int y = blockIdx.y, x0 = blockIdx.x * PK_TILE;
    for (int k = threadIdx.x; k < C * PK_TILE; k += 256) {
        const int c = k / PK_TILE, xx = k - c * PK_TILE, x = x0 + xx;
        tile[xx][c] = (y < H && x < W && c < Cin) ? in[((size_t)c * H + y) * W + x] : 0.0f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C * PK_TILE; k += 256) {
        const int xx = k / C, c = k - xx * C, x = x0 + xx;
        if (x < Wp) out[((size_t)y * Wp + x) * C + c] = tile[xx][c];
    }
}

hipError_t launch_pack_plane(const float *in, float *out, int Cin, int C, int H, int W, hipStream_t stream)
{
    const int Wp = (W == 1) ? 1 : W + 1;
    if (W > 1 && C <= TVR_CA) {
        hipLaunchKernelGGL(pack_plane_tiled_kernel, dim3((unsigned)((Wp + PK_TILE - 1) / PK_TILE), (unsigned)(H + 1)), dim3(256), 0, stream, in, out, Cin, C, H, W, Wp);
        return hipGetLastError();
    }
    const long long total = (long long)(H + 1) * Wp * C;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_plane_kernel, dim3(grid), dim3(256), 0, stream, in, out, Cin, C, H, W, Wp);
    return hipGetLastError();
}

// MLP weights -> fp16 hi/lo operand images.  One thread per (row, k position).
//  mode 0: W1 LDS image  [s][h][128][8 halfs]: kpos = 16s + 8h + j  <->  derived (i%5) of base acc_row(i/5, h), i = 8s + j;
//          base row 31's plain slot (the constant-1 input) carries b1
//  mode 1: W2 LDS image  [s][h][128][8 halfs]: kpos = 16s + 8h + j  <->  hidden unit 16s + 8(j>>2) + 4h + (j&3)
//  mode 2: basis fragments, hi [9][2][27][8] (LDS image) and lo [9][2][32][8] (global): row r < 27, k = 16s + 8h + j (natural)
//  (layer 3 runs as fp32 FMAs: W3 [3][128] fp32 and b3 are copied into the LDS image as they are, tvr_api.hip)
//  mode 4: mode 0 for MLPRender_Fea_Ref (REFTensoRF.py:19-24: [dot, features, viewdirs, PE(features), PE(viewdirs)], 151 inputs):
//          every index moves up by one and base row 30's plain slot carries input 0 (dot)
// Shapes smaller than the kernels' (hidden width < 128, fea_pe / view_pe < 2, fewer appearance components) are packed with zero weights:
// a hidden unit that does not exist outputs relu(0) = 0 and feeds zero columns, an input that does not exist has a zero column.
__global__ __launch_bounds__(256) void pack_mlp_kernel(const float *__restrict__ W, const float *__restrict__ bias,
                                                       unsigned short *__restrict__ out_hi, unsigned short *__restrict__ out_lo, int mode, const MlpShape sh)
{
    const int nrows = (mode <= 1 || mode >= 4) ? 128 : (mode == 2 ? 32 : 4);
    const int K = (mode == 0 || mode == 4) ? 160 : (mode == 2 ? 144 : (mode == 5 ? 16 * TVR_GEN_KS : 128));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * K) return;
    const int row = i / K, kpos = i - row * K;
    const int s = kpos >> 4, hh = (kpos >> 3) & 1, j = kpos & 7;
    float w = 0.0f;
    if (mode == 0) {
        const int ii = 8 * s + j;
        const int c = acc_row(ii / 5, hh), idx = ref_in_index(c, ii % 5, sh.fea_pe, sh.view_pe);
        if (idx >= 0 && row < sh.featureC) w = W[(size_t)row * sh.n_in + idx];
        if (c == 31 && ii % 5 == 0 && row < sh.featureC) w = bias[row];      // the constant-1 input (base row 31): b1 rides in the weight image
    } else if (mode == 4) {
        const int ii = 8 * s + j, c = acc_row(ii / 5, hh), t = ii % 5;
        const int idx = (c == TVR_APPDIM + 3) ? (t == 0 ? 0 : -1) : (ref_in_index(c, t) >= 0 ? ref_in_index(c, t) + 1 : -1);
        if (idx >= 0) w = W[(size_t)row * TVR_NIN_REF + idx];
        if (c == 31 && t == 0) w = bias[row];
    } else if (mode == 5) {                 // general frequencies: slot i = 8s + j of lane half hh is derived value i % 13 of base value acc_row(i / 13, hh)
        const int ii = 8 * s + j, c = acc_row(ii / TVR_GEN_T, hh), t = ii % TVR_GEN_T, idx = gen_in_index(c, t, sh.fea_pe, sh.view_pe);
        if (idx >= 0 && row < sh.featureC) w = W[(size_t)row * sh.n_in + idx];
        if (c == 31 && t == 0 && row < sh.featureC) w = bias[row];
    } else if (mode == 1) {
        const int u = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
        if (row < sh.featureC && u < sh.featureC) w = W[(size_t)row * sh.featureC + u];
    } else if (mode == 2) {
        const int pl = kpos / TVR_CA, c = kpos - pl * TVR_CA;              // the kernels' k = 48 * plane + channel; basis_mat's column = its plane's offset + channel
        if (row < TVR_APPDIM && c < sh.app_n_comp[pl]) w = W[(size_t)row * sh.k_app + sh.app_off[pl] + c];
    } else {
        if (row < 3) w = W[(size_t)row * TVR_FEATC + (16 * s + 8 * (j >> 2) + 4 * hh + (j & 3))];
    }
    unsigned hi, lo;
    split2(w, 0.0f, hi, lo);
    if (mode == 5) {                                                             // streamed image: per k-step [hi: [h][row 128][8] | lo: the same], 8 KB
        out_hi[((s * 2 + 0) * 2 + hh) * 1024 + row * 8 + j] = (unsigned short)hi;
        out_hi[((s * 2 + 1) * 2 + hh) * 1024 + row * 8 + j] = (unsigned short)lo;
    } else if (mode == 0 || mode == 1 || mode == 4) {                            // k-step major weight image: [s][h][row 128][8]
        out_hi[((s * 2 + hh) * 128 + row) * 8 + j] = (unsigned short)hi;
        out_lo[((s * 2 + hh) * 128 + row) * 8 + j] = (unsigned short)lo;
    } else if (mode == 2) {                                                      // basis: hi parts -> the LDS image [s][h][27 rows][8], lo parts -> global [s][h][32 rows][8]
        if (row < TVR_IMG_BASH_ROWS) out_hi[((s * 2 + hh) * TVR_IMG_BASH_ROWS + row) * 8 + j] = (unsigned short)hi;
        out_lo[((s * 2 + hh) * 32 + row) * 8 + j] = (unsigned short)lo;
    } else {
        unsigned short *o = out_hi + ((size_t)((row * 8 + s) * 2 + hh)) * 16;    // [hi 8 | lo 8] per (row, s, h)
        o[j] = (unsigned short)hi;
        o[8 + j] = (unsigned short)lo;
    }
}

hipError_t launch_pack_mlp(const float *W, const float *bias, void *out_hi, void *out_lo, int mode, const MlpShape &sh, hipStream_t stream)
{
    const int n = ((mode <= 1 || mode >= 4) ? 128 : (mode == 2 ? 32 : 4)) * ((mode == 0 || mode == 4) ? 160 : (mode == 2 ? 144 : (mode == 5 ? 16 * TVR_GEN_KS : 128)));
    hipLaunchKernelGGL(pack_mlp_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, W, bias, (unsigned short *)out_hi, (unsigned short *)out_lo, mode, sh);
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
