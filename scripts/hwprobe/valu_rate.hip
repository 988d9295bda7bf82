// probe (run on the GPU box): issue cost of the VALU instructions the shade kernel is made of, in cycles per wave-level instruction on one SIMD
// (s_memtime around a long unrolled stream of independent instructions; 1 and 2 waves per SIMD; with and without an MFMA stream in the partner wave).
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define REP 64
#define ITER 200

// 8 independent destination registers per op kind, round-robin: no back-to-back dependency
#define OPS8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define RUN(NAME, OP)                                                                                              \
    template <> __device__ void body<NAME>(float &x, unsigned &u) {                                                \
        for (int it = 0; it < ITER; ++it) { asm volatile(OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP)               \
            :: : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"); }   \
    }
enum { K_FMA, K_MUL, K_PKFMA, K_CVTPKRTZ, K_FMAMIX, K_MIXLO, K_MAXI, K_SIN, K_RNDNE, K_CVTPK, K_ADDU, K_LSHLADD64, K_PKMUL, K_MOV, K_FMAC_DEP, K_N };
template <int K> __device__ void body(float &x, unsigned &u);
#define OP_FMA(i) "v_fma_f32 v4" #i ", v30, v31, v32\n"
#define OP_MUL(i) "v_mul_f32 v4" #i ", v30, v31\n"
#define OP_PKFMA(i) "v_pk_fma_f32 v[" PK##i "], v[30:31], v[32:33], v[34:35]\n"
#define PK0 "40:41"
#define PK1 "42:43"
#define PK2 "44:45"
#define PK3 "46:47"
#define PK4 "48:49"
#define PK5 "50:51"
#define PK6 "52:53"
#define PK7 "54:55"
#define OP_CVTPKRTZ(i) "v_cvt_pkrtz_f16_f32 v4" #i ", v30, v31\n"
#define OP_FMAMIX(i) "v_fma_mix_f32 v4" #i ", v30, -1.0, v31 op_sel_hi:[1,0,0]\n"
#define OP_MIXLO(i) "v_fma_mixlo_f16 v4" #i ", v30, -1.0, v31 op_sel_hi:[1,0,0]\n"
#define OP_MAXI(i) "v_max_i32 v4" #i ", 0, v30\n"
#define OP_SIN(i) "v_sin_f32 v4" #i ", v30\n"
#define OP_RNDNE(i) "v_rndne_f32 v4" #i ", v30\n"
#define OP_CVTPK(i) "v_cvt_pk_f16_f32 v4" #i ", v30, v31\n"
#define OP_ADDU(i) "v_add_u32 v4" #i ", v30, v31\n"
#define OP_LSHLADD64(i) "v_lshl_add_u64 v[" PK##i "], v[30:31], 4, v[32:33]\n"
#define OP_PKMUL(i) "v_pk_mul_f32 v[" PK##i "], v[30:31], v[32:33]\n"
#define OP_MOV(i) "v_mov_b32 v4" #i ", v30\n"
#define OP_FMAC_DEP(i) "v_fmac_f32 v40, v30, v31\n"
RUN(K_FMA, OP_FMA) RUN(K_MUL, OP_MUL) RUN(K_PKFMA, OP_PKFMA) RUN(K_CVTPKRTZ, OP_CVTPKRTZ) RUN(K_FMAMIX, OP_FMAMIX) RUN(K_MIXLO, OP_MIXLO)
RUN(K_MAXI, OP_MAXI) RUN(K_SIN, OP_SIN) RUN(K_RNDNE, OP_RNDNE) RUN(K_CVTPK, OP_CVTPK) RUN(K_ADDU, OP_ADDU) RUN(K_LSHLADD64, OP_LSHLADD64)
RUN(K_PKMUL, OP_PKMUL) RUN(K_MOV, OP_MOV) RUN(K_FMAC_DEP, OP_FMAC_DEP)

// MODE 0: every wave runs the VALU stream.  MODE 1: waves 4..7 (the SIMD partners of 0..3) run an MFMA stream instead (same duration or longer).
template <int K, int MODE>
__global__ void probe(unsigned long long *out, float *sink)
{
    const int wave = threadIdx.x >> 6;
    float x = threadIdx.x;
    unsigned u = threadIdx.x;
    asm volatile("v_mov_b32 v30, 1.0\n v_mov_b32 v31, 0x3c003c00\n v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n"
                 ::: "v30", "v31", "v32", "v33", "v34", "v35");
    if (MODE >= 1 && wave >= 4) {
        f32x16 acc = {0};
        h8 a = {1, 1, 1, 1, 1, 1, 1, 1};
        f32x16 acc2 = {0};
        const int n = MODE == 1 ? ITER * 6 : ITER;              // MODE 1: longer than any VALU stream measured; MODE >= 2: the MFMA stream is the one timed
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n; ++it) {                        // 2 x 8 MFMAs x 32 cycles per iteration
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc2, 0, 0, 0);
            }
        }
        if (acc[0] + acc2[0] == 12345.f) sink[0] = acc[1];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
        return;
    }
    __builtin_amdgcn_s_setprio(MODE == 1 || MODE == 2 ? 2 : 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < (MODE >= 2 ? 12 : 1); ++rep) body<K>(x, u);       // MODE >= 2: outlasts the MFMA stream
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int K>
static void run(const char *name, unsigned long long *out, float *sink)
{
    double res[3];
    int cfg = 0;
    for (int waves : {4, 8}) {
        hipLaunchKernelGGL((probe<K, 0>), dim3(256), dim3(64 * waves), 0, 0, out, sink);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        res[cfg++] = (double)h[0] / (ITER * REP);
    }
    hipLaunchKernelGGL((probe<K, 1>), dim3(256), dim3(64 * 8), 0, 0, out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    res[2] = (double)h[0] / (ITER * REP);
    double mf[2];
    for (int mode : {2, 3}) {
        (void)hipMemset(out, 0, 256 * 8 * 8);
        if (mode == 2) hipLaunchKernelGGL((probe<K, 2>), dim3(256), dim3(64 * 8), 0, 0, out, sink);
        else hipLaunchKernelGGL((probe<K, 3>), dim3(256), dim3(64 * 8), 0, 0, out, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        mf[mode - 2] = (double)h[4] / (ITER * 16);
    }
    printf("%-22s cycles per instruction: 1 wave/SIMD %6.2f   2 waves/SIMD (both this stream) %6.2f   beside a partner wave of back-to-back MFMAs %6.2f"
           "   | the partner's cycles per MFMA beside this stream: stream at s_setprio 2 %6.2f, at 0 %6.2f\n", name, res[0], res[1], res[2], mf[0], mf[1]);
}

int main()
{
    unsigned long long *out; float *sink;
    (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&sink, 64);
    run<K_MOV>("v_mov_b32", out, sink);
    run<K_FMA>("v_fma_f32", out, sink);
    run<K_MUL>("v_mul_f32", out, sink);
    run<K_FMAC_DEP>("v_fmac_f32 (dependent)", out, sink);
    run<K_PKFMA>("v_pk_fma_f32", out, sink);
    run<K_PKMUL>("v_pk_mul_f32", out, sink);
    run<K_CVTPKRTZ>("v_cvt_pkrtz_f16_f32", out, sink);
    run<K_CVTPK>("v_cvt_pk_f16_f32", out, sink);
    run<K_FMAMIX>("v_fma_mix_f32", out, sink);
    run<K_MIXLO>("v_fma_mixlo_f16", out, sink);
    run<K_MAXI>("v_max_i32", out, sink);
    run<K_SIN>("v_sin_f32", out, sink);
    run<K_RNDNE>("v_rndne_f32", out, sink);
    run<K_ADDU>("v_add_u32", out, sink);
    run<K_LSHLADD64>("v_lshl_add_u64", out, sink);
    return 0;
}
