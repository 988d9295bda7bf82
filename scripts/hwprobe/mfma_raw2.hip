// probe (run on the GPU box): how many wait states does "VALU writes a VGPR -> MFMA reads it as A/B" need on gfx950, and does an
// already-satisfied `s_waitcnt` count as one?  hipcc's hazard recognizer pads VALU-write -> MFMA-read to 2 wait states and counts ANY
// instruction in between as one, s_waitcnt included: the round-1 build that corrupted (commit 2db8f31) contains
//     v_cvt_pkrtz_f16_f32 v29, v29, v42 ; s_waitcnt vmcnt(12) ; s_nop 0 ; v_mfma_f32_32x32x16_f16 v[6:21], v[114:117], v[26:29], v[6:21]
// Each iteration sets the LAST dword of the A (or B) operand to 2.0 (fp16 x2) right in front of an MFMA, then back to 1.0 in front of the next.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_raw2.hip -o mfma_raw2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ITER 2000

#define GAP0 ""
#define GAP1 "s_nop 0\n"
#define GAP2 "s_nop 1\n"
#define GAP3 "s_nop 2\n"
#define GAP4 "s_nop 3\n"
#define GAPW "s_waitcnt vmcnt(0)\n"                       /* nothing outstanding: satisfied at once */
#define GAPW1 "s_waitcnt vmcnt(0)\n s_nop 0\n"            /* what hipcc emitted in the corrupting build */
#define GAPWW "s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(0)\n"
#define GAPV "v_mov_b32 v90, v91\n"                       /* one independent VALU instruction */
#define GAPVV "v_mov_b32 v90, v91\n v_mov_b32 v92, v91\n"
#define GAPS "s_mov_b32 s90, 0\n"                         /* one SALU instruction */
#define GAPSS "s_mov_b32 s90, 0\n s_mov_b32 s91, 0\n"

// OPB = 0: the written register is the last dword of A (v103); 1: last dword of B (v107).  CVT: produce it by v_cvt_pkrtz instead of v_mov.
#define BODY(GAP)                                                                                                              \
    asm volatile("v_mov_b32 v100, %2\n v_mov_b32 v101, %2\n v_mov_b32 v102, %2\n v_mov_b32 v104, %2\n v_mov_b32 v105, %2\n v_mov_b32 v106, %2\n" \
                 "v_mov_b32 v103, %2\n v_mov_b32 v107, %2\n s_nop 7\n"                                                          \
                 "v_mov_b32 v103, %1\n" GAP "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n s_nop 7\n"                \
                 "v_mov_b32 v103, %2\n s_nop 7\n"                                                                              \
                 "v_mov_b32 v107, %1\n" GAP "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n s_nop 7\n"                \
                 "v_mov_b32 v107, %2\n s_nop 7\n"                                                                              \
                 : "+v"(acc) : "v"(two), "v"(uno) : "v90", "v92", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "s90", "s91")

template <int G>
__global__ void probe(float *__restrict__ out, int *__restrict__ bad)
{
    f32x16 acc = {0};
    unsigned two = 0x40004000u, uno = 0x3C003C00u;
    for (int it = 0; it < ITER; ++it) {
        if (G == 0) BODY(GAP0);
        if (G == 1) BODY(GAP1);
        if (G == 2) BODY(GAP2);
        if (G == 3) BODY(GAP3);
        if (G == 4) BODY(GAP4);
        if (G == 5) BODY(GAPW);
        if (G == 6) BODY(GAPW1);
        if (G == 7) BODY(GAPWW);
        if (G == 8) BODY(GAPV);
        if (G == 9) BODY(GAPVV);
        if (G == 10) BODY(GAPS);
        if (G == 11) BODY(GAPSS);
    }
    // the last dword of a fragment holds 2 of a lane's 8 k-values, in both lane halves: 4 of an element's 16 k-terms are 2.0 * 1.0 when the
    // MFMA reads the FRESH value (12 + 8 = 20 per MFMA, 40 per iteration); a STALE read gives 16 per MFMA
    const float want = (float)ITER * 40.0f;
    int wrong = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) wrong += (acc[r] != want);
    if (wrong) atomicAdd(bad, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = acc[0];
}

template <int G>
static void run(const char *tag, float *out, int *bad)
{
    for (int w : {4, 8}) {
        (void)hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(probe<G>, dim3(256), dim3(64 * w), 0, 0, out, bad);
        int h = 0; float o = 0;
        (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
        printf("VALU write -> [%-28s] -> MFMA read, waves/SIMD %d: lanes that read a stale operand %7d of %d (acc[0] = %.0f; fresh %.0f, stale %.0f)\n", tag, w / 4, h,
               256 * 64 * w, o, (float)ITER * 40.0f, (float)ITER * 32.0f);
    }
}

int main()
{
    float *out; int *bad;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&bad, 4);
    run<0>("nothing", out, bad);
    run<1>("s_nop 0", out, bad);
    run<2>("s_nop 1", out, bad);
    run<3>("s_nop 2", out, bad);
    run<4>("s_nop 3", out, bad);
    run<5>("s_waitcnt vmcnt(0)", out, bad);
    run<6>("s_waitcnt vmcnt(0); s_nop 0", out, bad);
    run<7>("s_waitcnt x2", out, bad);
    run<8>("1 independent v_mov", out, bad);
    run<9>("2 independent v_mov", out, bad);
    run<10>("1 s_mov", out, bad);
    run<11>("2 s_mov", out, bad);
    return 0;
}
