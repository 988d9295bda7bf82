// probe (GPU box): in the power-limited regime (every CU issuing dense f16 MFMAs on random operands), which MFMA shape / tile shape delivers more FLOP/s?
// Round 5 rewrite.  Round 4's 16x16 arm was defective: hipcc renamed its four accumulators into partially overlapping register ranges
// (a[0:3] <- a[2:5]), which the matrix pipe serialises: 23.5 cycles per MFMA where the shape's rate is 16.  Here every MFMA, LDS read and wait is an
// `asm volatile` statement with tied accumulators: the order below IS the instruction stream (checked: llvm-objdump shows no s_nop / v_mov / waitcnt of the
// compiler's inside the loop bodies), and every arm prints its CYCLES PER MFMA — an arm that does not reach ~32 (32x32x16) or ~16 (16x16x32) on half the
// chip measures itself, not the shape.
//
// Arms (one wave per SIMD, 256-thread workgroups, one per CU):
//   bare32 / bare16     MFMAs only, operands in registers, every MFMA with other A and B registers than its predecessor; 4 / 16 independent accumulators
//   lds32               the shipped shade kernel's hidden layer: per 32-row block one {hi, lo} A-fragment pair (2 ds_read_b128, ring of 4 read two blocks
//                       ahead, counted lgkmcnt) feeds 3 MFMAs (l*bhi, h*blo, h*bhi) on the block's accumulator — 32 columns per wave
//   lds16               the proposed form: per 16-row block one pair feeds 6 MFMAs 16x16x32 = two 16-column B tiles x three products — the same LDS bytes per
//                       FLOP, the same 64 accumulator registers per 128 rows x 32 columns
//   lds32x64 / lds16x64 64 columns per wave: a pair feeds 6 (12) MFMAs — half the LDS bytes per FLOP, 128 accumulator registers
// Build: hipcc --offload-arch=gfx950 -O2 mfma_shape_power.hip -o mfma_shape_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define M32(acc, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define M16(acc, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n))

__device__ __forceinline__ unsigned rnd_h2(unsigned &s)
{
    // two random fp16 in (-1, 1): sign, exponent 0x30..0x3b, random mantissa
    unsigned r = 0;
    for (int k = 0; k < 2; ++k) {
        s = s * 1664525u + 1013904223u;
        const unsigned m = (s >> 9) & 0x3ffu, ex = 8u + ((s >> 20) % 7u), sg = (s >> 31);
        r |= ((sg << 15) | (ex << 10) | m) << (16 * k);                 // exponent field 8..14: 2^-7 <= |x| < 1, random sign and mantissa: sums stay finite
    }
    return r;
}
__device__ __forceinline__ u32x4 rnd_frag(unsigned &s) { return u32x4{rnd_h2(s), rnd_h2(s), rnd_h2(s), rnd_h2(s)}; }

enum { BARE32, BARE16, LDS32, LDS16, LDS32X64, LDS16X64 };
struct Out { unsigned long long cyc, ref; };

template <int ARM>
__global__ __launch_bounds__(256) void burn(int iters, float *sink, Out *out)
{
    __shared__ u32x4 lds[2048];                               // 32 KB of random fragments
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = rnd_frag(s);
    __syncthreads();
    const unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    u32x4 B[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) B[i] = rnd_frag(s);
    float total = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (ARM == BARE32) {
        f32x16 c[4] = {{0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) M32(c[j & 3], B[(j * 3) & 7], B[(j * 5 + 1) & 7]);
        }
        total = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (ARM == BARE16) {
        f32x4 d[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) M16(d[j & 15], B[(j * 3) & 7], B[(j * 5 + 1) & 7]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) total += d[i][i & 3];
    } else if constexpr (ARM == LDS32 || ARM == LDS32X64) {
        // a "layer": 128 rows x K = 128 (8 k-steps of 16) x 32 (64) columns; block q = (k-step, row block); per block 2 reads + 3 (6) MFMAs
        constexpr int NT = ARM == LDS32 ? 1 : 2;
        f32x16 c[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) c[i][t] = f32x16{0};
        u32x4 Ah[4], Al[4];
        DSR(Ah[0], la, 0); DSR(Al[0], la, 1024);
        DSR(Ah[1], la, 2048); DSR(Al[1], la, 3072);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {                     // two k-steps per loop trip: ring indices are static
                // ONE asm block per row block (between separate asm statements hipcc's hazard recogniser puts an s_nop 0 after every second MFMA)
#define RD2 "ds_read_b128 %[nh], %[la] offset:%[oh]\n\tds_read_b128 %[nl], %[la] offset:%[ol]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define X32(c, a, b) "v_mfma_f32_32x32x16_f16 %[" #c "], %[" #a "], %[" #b "], %[" #c "]\n\t"
#define X16(c, a, b) "v_mfma_f32_16x16x32_f16 %[" #c "], %[" #a "], %[" #b "], %[" #c "]\n\t"
                if constexpr (NT == 1)
                    asm volatile(RD2 X32(c0, al, b0h) X32(c0, ah, b0l) X32(c0, ah, b0h)
                                 : [nh] "=&v"(Ah[(q + 2) & 3]), [nl] "=&v"(Al[(q + 2) & 3]), [c0] "+v"(c[q & 3][0])
                                 : [la] "v"(la), [ah] "v"(Ah[q & 3]), [al] "v"(Al[q & 3]), [b0h] "v"(B[2 * (q >> 2)]), [b0l] "v"(B[2 * (q >> 2) + 1]),
                                   [oh] "n"(((q + 2) & 15) * 2048), [ol] "n"(((q + 2) & 15) * 2048 + 1024));
                else
                    asm volatile(RD2 X32(c0, al, b0h) X32(c1, al, b1h) X32(c0, ah, b0l) X32(c1, ah, b1l) X32(c0, ah, b0h) X32(c1, ah, b1h)
                                 : [nh] "=&v"(Ah[(q + 2) & 3]), [nl] "=&v"(Al[(q + 2) & 3]), [c0] "+v"(c[q & 3][0]), [c1] "+v"(c[q & 3][NT - 1])
                                 : [la] "v"(la), [ah] "v"(Ah[q & 3]), [al] "v"(Al[q & 3]), [b0h] "v"(B[2 * (q >> 2)]), [b0l] "v"(B[2 * (q >> 2) + 1]),
                                   [b1h] "v"(B[4 + 2 * (q >> 2)]), [b1l] "v"(B[4 + 2 * (q >> 2) + 1]), [oh] "n"(((q + 2) & 15) * 2048), [ol] "n"(((q + 2) & 15) * 2048 + 1024));
            }
        }
        LGKM(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) total += c[i][t][i];
        total += __builtin_bit_cast(float, Ah[0][0] ^ Al[1][1] ^ Ah[2][2] ^ Al[3][3]);
    } else {
        // 16x16x32: block q = 16-row block of one 32-deep k-step; per block 2 reads + 6 (12) MFMAs on the block's NT accumulators
        constexpr int NT = ARM == LDS16 ? 2 : 4;
        f32x4 d[8][NT];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) d[i][t] = f32x4{0, 0, 0, 0};
        u32x4 Ah[4], Al[4];
        DSR(Ah[0], la, 0); DSR(Al[0], la, 1024);
        DSR(Ah[1], la, 2048); DSR(Al[1], la, 3072);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {                     // one k-step of 32 per loop trip
                if constexpr (NT == 2)
                    asm volatile(RD2 X16(c0, al, b0h) X16(c1, al, b1h) X16(c0, ah, b0l) X16(c1, ah, b1l) X16(c0, ah, b0h) X16(c1, ah, b1h)
                                 : [nh] "=&v"(Ah[(q + 2) & 3]), [nl] "=&v"(Al[(q + 2) & 3]), [c0] "+v"(d[q][0]), [c1] "+v"(d[q][1])
                                 : [la] "v"(la), [ah] "v"(Ah[q & 3]), [al] "v"(Al[q & 3]), [b0h] "v"(B[0]), [b0l] "v"(B[1]), [b1h] "v"(B[2]), [b1l] "v"(B[3]),
                                   [oh] "n"(((q + 2) & 15) * 2048), [ol] "n"(((q + 2) & 15) * 2048 + 1024));
                else
                    asm volatile(RD2 X16(c0, al, b0h) X16(c1, al, b1h) X16(c2, al, b2h) X16(c3, al, b3h) X16(c0, ah, b0l) X16(c1, ah, b1l) X16(c2, ah, b2l) X16(c3, ah, b3l)
                                 X16(c0, ah, b0h) X16(c1, ah, b1h) X16(c2, ah, b2h) X16(c3, ah, b3h)
                                 : [nh] "=&v"(Ah[(q + 2) & 3]), [nl] "=&v"(Al[(q + 2) & 3]), [c0] "+v"(d[q][0]), [c1] "+v"(d[q][1]), [c2] "+v"(d[q][NT - 2]), [c3] "+v"(d[q][NT - 1])
                                 : [la] "v"(la), [ah] "v"(Ah[q & 3]), [al] "v"(Al[q & 3]), [b0h] "v"(B[0]), [b0l] "v"(B[1]), [b1h] "v"(B[2]), [b1l] "v"(B[3]),
                                   [b2h] "v"(B[4]), [b2l] "v"(B[5]), [b3h] "v"(B[6]), [b3l] "v"(B[7]), [oh] "n"(((q + 2) & 15) * 2048), [ol] "n"(((q + 2) & 15) * 2048 + 1024));
            }
        }
        LGKM(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) total += d[i][t][i & 3];
        total += __builtin_bit_cast(float, Ah[0][0] ^ Al[1][1] ^ Ah[2][2] ^ Al[3][3]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (total == 12345.678f) sink[0] = total;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out->cyc = t1 - t0; out->ref = r1 - r0; }
}

struct Arm { const char *name; int mfma_per_iter; double flop_per_mfma; };
static const Arm ARMS[6] = {
    {"bare32    MFMA 32x32x16 only                        ", 16, 32768.0},
    {"bare16    MFMA 16x16x32 only                        ", 32, 16384.0},
    {"lds32     32 cols, pair -> 3 MFMA 32x32x16 (shipped)", 24, 32768.0},
    {"lds16     32 cols, pair -> 6 MFMA 16x16x32          ", 48, 16384.0},
    {"lds32x64  64 cols, pair -> 6 MFMA 32x32x16          ", 48, 32768.0},
    {"lds16x64  64 cols, pair -> 12 MFMA 16x16x32         ", 96, 16384.0},
};
template <int ARM>
static void run(int blocks, float *sink, Out *dout, double target_ms)
{
    const Arm &A = ARMS[ARM];
    // iterations for ~target_ms at 2 GHz and the nominal rate of the shape
    const double cyc_per_iter = A.mfma_per_iter * (A.flop_per_mfma == 32768.0 ? 32.0 : 16.0);
    const int iters = (int)(target_ms * 1e-3 * 2.0e9 / cyc_per_iter);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int NW = 6, NT = 6;                                // back-to-back launches: NW to settle the clock, NT timed
    for (int i = 0; i < NW; ++i) hipLaunchKernelGGL((burn<ARM>), dim3(blocks), dim3(256), 0, 0, iters, sink, dout);
    (void)hipEventRecord(e0);
    for (int i = 0; i < NT; ++i) hipLaunchKernelGGL((burn<ARM>), dim3(blocks), dim3(256), 0, 0, iters, sink, dout);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= NT;
    Out o; (void)hipMemcpy(&o, dout, sizeof(o), hipMemcpyDeviceToHost);
    const double n_mfma = (double)iters * A.mfma_per_iter;
    const double flop = (double)blocks * 4 * n_mfma * A.flop_per_mfma;
    printf("%s %3d CUs: %7.3f ms  clock %.3f GHz  %6.2f cycles/MFMA  %6.0f TFLOP/s\n", A.name, blocks, ms, (double)o.cyc / ((double)o.ref * 10.0),
           (double)o.cyc / n_mfma, flop / (ms * 1e-3) / 1e12);
    fflush(stdout);
}
int main(int argc, char **argv)
{
    float *sink; Out *dout;
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&dout, sizeof(Out));
    const double tms = argc > 1 ? atof(argv[1]) : 12.0;       // per-launch length: the shade kernel's own
    for (int rep = 0; rep < 2; ++rep) {
        for (int blocks : {128, 256}) {
            run<BARE32>(blocks, sink, dout, tms); run<BARE16>(blocks, sink, dout, tms);
            run<LDS32>(blocks, sink, dout, tms); run<LDS16>(blocks, sink, dout, tms);
            run<LDS32X64>(blocks, sink, dout, tms); run<LDS16X64>(blocks, sink, dout, tms);
        }
        printf("\n");
    }
    return 0;
}
