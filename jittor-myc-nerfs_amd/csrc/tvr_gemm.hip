// tvr_gemm.hip — C[Ka x Kb] = A^T[Ka x M] * B[M x Kb] for tall-skinny operands (M = the ~3.5e5 appearance samples of a training batch,
// Ka, Kb <= 160): the weight gradients dW = dY^T X of the training step's Linears (tensorBase.py:69-71, tensoRF.py:150, train.py:258).
// The library GEMM picks 32x32 macro tiles with stream-K for these shapes and runs at ~15 TFLOP/s (0.9 ms per gradient, three per step);
// here every workgroup reduces its own slab of rows with fp32-input MFMAs (v_mfma_f32_32x32x2_f32, fp32 semantics, no splitting needed:
// no atomics) and writes its Ka x Kb partial to a scratch slab; a second kernel sums the slabs in a fixed order, so the result is
// bit-reproducible.  (Adding the partials into C with atomics instead costs 0.3 ms: 256 workgroups hit the same 19 k addresses at once.)
//
// Two workgroups per CU (two waves per SIMD; the requested LDS caps it there).  The MFMA operands come straight from global loads, so the
// load-behind-MFMA hazard described in tvr_shade.hip is designed out the same way: within an iteration the loads (into the NEXT buffers)
// are issued before the MFMAs (which read the CURRENT buffers, live until their last use), and every iteration ends with a
// compiler-visible read of the last accumulator, which drains the wave's MFMAs before the next iteration's loads can reuse a register.
#include <hip/hip_runtime.h>
#include "tvr_kernels.h"
#include "tvr_mfma.h"

#define TG_WAVES 4
#define TG_MAXT 5                 // 32x32 output tiles per wave (20 per workgroup: 128 x 160)
#define TG_UNROLL 4               // 2-row steps in flight

__global__ __launch_bounds__(64 * TG_WAVES) void gemm_tn_kernel(const float *__restrict__ A, const int lda, const int Ka,
                                                                 const float *__restrict__ B, const int ldb, const int Kb,
                                                                 const long long M_cap, float *__restrict__ P, const long long rpb_host, const unsigned *__restrict__ m_dev)
{
    // m_dev: the row count lives on the device (M_cap = capacity, the grid is sized for it); the slabs are then cut from the true count here
    const long long M = m_dev ? ((long long)*m_dev < M_cap ? (long long)*m_dev : M_cap) : M_cap;
    long long rows_per_block = rpb_host;
    if (m_dev) {
        rows_per_block = ((M + (long long)gridDim.x - 1) / (long long)gridDim.x + 15) / 16 * 16;
        if (rows_per_block < 64) rows_per_block = 64;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, k = lane >> 5;
    const int nrb = (Ka + 31) >> 5, ncb = (Kb + 31) >> 5, ntiles = nrb * ncb;
    // this wave's tiles t = wave, wave + 4, ...: row offsets into A / B for this lane, or -1 past the edge
    int offA[TG_MAXT], offB[TG_MAXT];
    f32x16 acc[TG_MAXT];
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        const int t = wave + q * TG_WAVES;
        const int rb = t / ncb, cb = t - rb * ncb;
        offA[q] = (t < ntiles && rb * 32 + i < Ka) ? rb * 32 + i : -1;
        offB[q] = (t < ntiles && cb * 32 + i < Kb) ? cb * 32 + i : -1;
        acc[q] = f32x16{0};
    }
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = m0 >= M ? m0 : (m0 + rows_per_block < M ? m0 + rows_per_block : M);
    // software pipeline: the operands of the next 8 rows are in flight while the MFMAs of the current 8 run
    float a[TG_UNROLL][TG_MAXT], b[TG_UNROLL][TG_MAXT];
    auto fetch = [&](long long m, float (&fa)[TG_UNROLL][TG_MAXT], float (&fb)[TG_UNROLL][TG_MAXT]) {
#pragma unroll
        for (int u = 0; u < TG_UNROLL; ++u) {
            const long long row = m + 2 * u + k;
            const bool in = row < m1;
            const long long rr = in ? row : m1 - 1;            // every load is unconditional (clamped address, value selected afterwards):
#pragma unroll                                                   // a predicated load is a branch, and its join waits for the data
            for (int q = 0; q < TG_MAXT; ++q) {
                const float va = A[rr * lda + (offA[q] >= 0 ? offA[q] : 0)];
                const float vb = B[rr * ldb + (offB[q] >= 0 ? offB[q] : 0)];
                fa[u][q] = (in && offA[q] >= 0) ? va : 0.0f;
                fb[u][q] = (in && offB[q] >= 0) ? vb : 0.0f;
            }
        }
    };
    fetch(m0, a, b);
    float drain = 0.0f;
    for (long long m = m0; m < m1; m += 2 * TG_UNROLL) {
        float an[TG_UNROLL][TG_MAXT], bn[TG_UNROLL][TG_MAXT];
        fetch(m + 2 * TG_UNROLL, an, bn);                      // rows >= m1 read as zero
#pragma unroll
        for (int u = 0; u < TG_UNROLL; ++u)
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], b[u][q], acc[q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        drain += acc[TG_MAXT - 1][15];                          // written by the last MFMA issued: all of them have completed
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < TG_UNROLL; ++u)
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) { a[u][q] = an[u][q]; b[u][q] = bn[u][q]; }
    }
    if (drain == 1.2345e-30f && P == nullptr) P[0] = drain;     // keeps `drain` alive; never true
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        const int t = wave + q * TG_WAVES;
        if (t >= ntiles) continue;
        const int rb = t / ncb, cb = t - rb * ncb;
        const int col = cb * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
            if (row < Ka && col < Kb) P[((size_t)blockIdx.x * Ka + row) * Kb + col] = acc[q][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ LDS-staged fp32 kernel (the default since round 2)
// The kernel above fetches every MFMA operand with its own dword load — a wave-level load costs the CU's L1 path ~18 cycles whatever its
// width (scripts/hwprobe/l1_width.hip) and a block of A is fetched once per column block of B: 40 wave-level loads per 2 rows and workgroup.
// Here a workgroup copies TL_ROWS rows of A and of B into LDS once per chunk (16-B loads when the rows are contiguous and aligned, dword
// loads otherwise), double-buffered; the waves take their fragments from there with ds_read_b32, one 2-row step ahead of the MFMAs; tiles
// a wave does not have cost no MFMA.  Global loads of the next chunk are issued before the chunk's MFMAs and written to LDS after the
// MFMAs have drained: no load is issued while an MFMA is in flight.
// Measured (scripts/gemm_bench.py, one box, M = 356 123): 128 x 150: 0.283 -> 0.241 ms; 128 x 128: 0.263 -> 0.193; 3 x 128: 0.224 -> 0.095;
// 27 x 144: 0.229 -> 0.111; 2.1e6 x 128 x 128: 1.36 -> 0.92.  What is left at the full shapes is the fp32 matrix pipe (0.107 ms for 20 tiles)
// PLUS ~0.09 ms of memory wait that two waves per SIMD (139 VGPRs + 80 accumulators) do not overlap; chunk sizes 8 / 32 rows and a second
// register set (prefetch distance 2) measured equal / slower.
#ifndef TL_ROWS
#define TL_ROWS 16
#endif
#ifndef TVR_GEMM_SPLIT16
#define TVR_GEMM_SPLIT16 1        // 0: the fused step's weight gradients keep the fp32 products (A/B switch)
#endif
#ifndef TG_DIAG
#define TG_DIAG 0                 // timing experiments only: 1 no MFMAs (memory + staging alone), 2 no global fetch after the first chunk (matrix pipe alone)
#endif
#define TL_PF ((TL_ROWS * 320 + 255) / 256)          // floats (or float4s / 4) per thread and chunk, at most: 20 tiles = 320 columns

// MODE 0: dword staging (any strides); 1: float4 staging of contiguous rows (lda == Ka, ldb == Kb: a chunk is one flat run of floats, any K);
// 2: float4 staging of strided rows (Ka, Kb, lda, ldb multiples of 4 — operands that are column blocks of wider matrices)
// F16: the products run on v_mfma_f32_32x32x16_f16 with both operands split into fp16 hi + lo (three products, fp32-grade; tvr_mfma.h) and A multiplied by
// the power of two *scale first — for operands whose range is KNOWN to fit fp16 at that scale: the fused training step's gradients, which its backward kernels
// already carried through fp16 at the same scale (and flag if they saturate).  A 16-row chunk is then ONE k-step: 3 MFMAs of 32 cycles per tile instead of 8 of
// 64, and the kernel runs at the staging path's rate (measured by leaving the MFMAs out: 4.9 TB/s of operand reads against 2.3 TB/s with the fp32 products).
// Wave w owns row block w of the result (Ka <= 128) and walks the column blocks: per chunk one A fragment and <= 5 B fragments are read from LDS and split.
template <int MODE, bool F16>
__global__ __launch_bounds__(64 * TG_WAVES) void gemm_tn_lds_kernel(const float *__restrict__ A, const int lda, const int Ka,
                                                                     const float *__restrict__ B, const int ldb, const int Kb,
                                                                     const long long M_cap, float *__restrict__ P, const long long rpb_host, const unsigned *__restrict__ m_dev,
                                                                     const int ones, const float *__restrict__ scale)
{
    // m_dev: the row count lives on the device (M_cap = capacity, the grid is sized for it); the slabs are then cut from the true count here
    // ones = 1: B has one more, VIRTUAL, column of ones (never read from memory): column Kb of the result is A^T 1 = the column sums of A — the bias
    // gradient of the Linear whose weight gradient this product is, without a second pass over A
    const long long M = m_dev ? ((long long)*m_dev < M_cap ? (long long)*m_dev : M_cap) : M_cap;
    long long rows_per_block = rpb_host;
    if (m_dev) {
        rows_per_block = ((M + (long long)gridDim.x - 1) / (long long)gridDim.x + 15) / 16 * 16;
        if (rows_per_block < 64) rows_per_block = 64;
    }
    constexpr bool VEC4 = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, k = lane >> 5;
    const int Kc = Kb + ones;                                           // columns of the result
    const int nrb = (Ka + 31) >> 5, ncb = (Kc + 31) >> 5, ntiles = nrb * ncb;
    const int na = TL_ROWS * Ka, nab = TL_ROWS * (Ka + Kb);             // floats of A / of A and B in one chunk (both multiples of 4)
    int offA[TG_MAXT], offB[TG_MAXT];
    f32x16 acc[TG_MAXT];
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        const int t = wave + q * TG_WAVES;
        const int rb = t / ncb, cb = t - rb * ncb;
        offA[q] = (t < ntiles && rb * 32 + i < Ka) ? rb * 32 + i : -1;
        offB[q] = (t < ntiles && cb * 32 + i < Kb) ? cb * 32 + i : ((t < ntiles && cb * 32 + i < Kc) ? -2 : -1);      // -2: the virtual ones column
        acc[q] = f32x16{0};
    }
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = m0 >= M ? m0 : (m0 + rows_per_block < M ? m0 + rows_per_block : M);
    // chunk [m, m + TL_ROWS) -> registers: element e of the chunk's A floats (row-major, Ka per row) followed by its B floats; rows >= m1 read as zero
    constexpr int NPRE = VEC4 ? 4 * ((TL_PF + 3) / 4) : TL_PF;
    float preA[NPRE];                                                  // (prefetch distance 2 with a second register set measured slower: 0.32 vs 0.23 ms)
    // dword staging: element e = tid + 256 j of a chunk is (row erow[j], column c) of A or B whatever the chunk — the division is done once
    int eoff[VEC4 ? 1 : TL_PF], erow[VEC4 ? 1 : TL_PF];
    // strided float4 staging: float4 j of a thread is (row vrow[j], columns voff[j] .. + 3) of A or B — the division is done once here too
    constexpr int NV = (TL_PF + 3) / 4;
    int voff[MODE == 2 ? NV : 1], vrow[MODE == 2 ? NV : 1];
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = 4 * (tid + 256 * j);                          // a float4 never straddles rows or A | B: Ka, Kb are multiples of 4
            const bool isa = e < na;
            const int el = isa ? e : e - na, K = isa ? Ka : Kb;
            const int r = el / K, c = el - r * K;
            vrow[j] = e < nab ? r : TL_ROWS;
            voff[j] = r * (isa ? lda : ldb) + c;
        }
    }
    if (!VEC4) {
#pragma unroll
        for (int j = 0; j < TL_PF; ++j) {
            const int e = tid + 256 * j;
            const bool isa = e < na;
            const int el = isa ? e : e - na, K = isa ? Ka : Kb;
            const int r = el / K, c = el - r * K;
            erow[j] = e < nab ? r : TL_ROWS;                           // TL_ROWS: never inside a chunk
            eoff[j] = r * (isa ? lda : ldb) + c;
        }
    }
    auto fetch = [&](long long m, float (&pre)[NPRE]) {
        const long long rows = m1 - m < TL_ROWS ? m1 - m : TL_ROWS;    // may be <= 0 past the end
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const bool in = vrow[j] < rows;
                const bool isa = 4 * (tid + 256 * j) < na;
                const float *base = isa ? A : B;
                const float4 v = *(const float4 *)(base + (in ? m * (long long)(isa ? lda : ldb) + voff[j] : 0));      // unconditional load, clamped address
                pre[4 * j] = in ? v.x : 0.0f; pre[4 * j + 1] = in ? v.y : 0.0f; pre[4 * j + 2] = in ? v.z : 0.0f; pre[4 * j + 3] = in ? v.w : 0.0f;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < (TL_PF + 3) / 4; ++j) {
                const int e = 4 * (tid + 256 * j);                      // a float4 never straddles A | B: na is a multiple of 4
                const bool isa = e < na;
                const int el = isa ? e : e - na, K = isa ? Ka : Kb;
                const long long valid = rows > 0 ? rows * K : 0;       // floats of this operand's part of the chunk that exist
                const float *src = (isa ? A + m * (long long)Ka : B + m * (long long)Kb) + el;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < nab) {
                    if (el + 3 < valid) v = *(const float4 *)src;
                    else {                                              // the ragged end of the last chunk
                        if (el < valid) v.x = src[0];
                        if (el + 1 < valid) v.y = src[1];
                        if (el + 2 < valid) v.z = src[2];
                    }
                }
                pre[4 * j] = v.x; pre[4 * j + 1] = v.y; pre[4 * j + 2] = v.z; pre[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < TL_PF; ++j) {
                const bool in = erow[j] < rows;
                const float *base = (tid + 256 * j < na) ? A : B;
                const float v = base[in ? m * (long long)(base == A ? lda : ldb) + eoff[j] : 0];     // unconditional load, clamped address
                pre[j] = in ? v : 0.0f;
            }
        }
    };
    auto stash = [&](float *buf, const float (&pre)[NPRE]) {
        if (VEC4) {
#pragma unroll
            for (int j = 0; j < (TL_PF + 3) / 4; ++j) {
                const int e = 4 * (tid + 256 * j);
                if (e < nab) *(float4 *)(buf + e) = make_float4(pre[4 * j], pre[4 * j + 1], pre[4 * j + 2], pre[4 * j + 3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TL_PF; ++j) {
                const int e = tid + 256 * j;
                if (e < nab) buf[e] = pre[j];
            }
        }
    };
    const int bufsz = (nab + 3) & ~3;
    fetch(m0, preA);
    stash(sm, preA);
    __syncthreads();
    if constexpr (F16) {
        static_assert(TL_ROWS == 16, "one fp16 k-step per chunk");
        const float sc = scale ? *scale : 1.0f;
        const bool has = wave < nrb;                                    // (wave-uniform)
        const int colA = wave * 32 + i;
        const bool okA = has && colA < Ka;
        f32x16 acc16[TG_MAXT];
#pragma unroll
        for (int q = 0; q < TG_MAXT; ++q) acc16[q] = f32x16{0};
        float drain16 = 0.0f;
        int cur16 = 0;
        for (long long m = m0; m < m1; m += TL_ROWS) {
            const bool more = m + TL_ROWS < m1;
            if (more) fetch(m + TL_ROWS, preA);
            __builtin_amdgcn_sched_barrier(0);
            const float *c_s = sm + cur16 * bufsz;
            if (has) {
                float a8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = okA ? c_s[(8 * k + j) * Ka + colA] * sc : 0.0f;       // rows 8 h .. 8 h + 7 of column colA (rows past m1 hold zeros)
                const Frag fa = split8(a8);
#pragma unroll
                for (int q = 0; q < TG_MAXT; ++q) {
                    if (q < ncb) {
                        const int col = q * 32 + i;
                        float b8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) b8[j] = col < Kb ? c_s[na + (8 * k + j) * Kb + col] : ((ones && col == Kb) ? 1.0f : 0.0f);
                        const Frag fb = split8(b8);
                        acc16[q] = MFMAH(fa.lo, fb.hi, acc16[q]);
                        acc16[q] = MFMAH(fa.hi, fb.lo, acc16[q]);
                        acc16[q] = MFMAH(fa.hi, fb.hi, acc16[q]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) drain16 += acc16[q][15];  // every MFMA of the chunk has completed before the next chunk's staging writes
            __builtin_amdgcn_sched_barrier(0);
            if (more) stash(sm + (cur16 ^ 1) * bufsz, preA);
            __syncthreads();
            cur16 ^= 1;
        }
        if (drain16 == 1.2345e-30f && P == nullptr) P[0] = drain16;
        if (has) {
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) {
                if (q >= ncb) continue;
                const int col = q * 32 + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
                    if (row < Ka && col < Kc) P[((size_t)blockIdx.x * Ka + row) * Kc + col] = acc16[q][r];
                }
            }
        }
        return;
    }
    float drain = 0.0f;
    int cur = 0;
    // LDS word offsets of this lane's operands (row k of a 2-row step; 0 and a zero mask for tiles / columns that do not exist)
    const int nq = __builtin_amdgcn_readfirstlane(wave < ntiles ? (ntiles - wave + TG_WAVES - 1) / TG_WAVES : 0);       // tiles of this wave
    int la[TG_MAXT], lb[TG_MAXT];
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        la[q] = k * Ka + (offA[q] >= 0 ? offA[q] : 0);
        lb[q] = na + k * Kb + (offB[q] >= 0 ? offB[q] : 0);
    }
    for (long long m = m0; m < m1; m += TL_ROWS) {
        const bool more = m + TL_ROWS < m1;
        if (more && !((TG_DIAG & 2) && m > m0)) fetch(m + TL_ROWS, preA);   // global loads of the next chunk: in flight during this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const float *c_s = sm + cur * bufsz;
        // the operands of step u + 1 are read from LDS while the MFMAs of step u run (an MFMA that waits for its own ds_read costs the LDS
        // latency every time: 12 k cycles per chunk instead of 2.6 k)
        float va[2][TG_MAXT], vb[2][TG_MAXT];
#pragma unroll
        for (int q = 0; q < TG_MAXT; ++q) { va[0][q] = c_s[la[q]]; vb[0][q] = c_s[lb[q]]; }
#pragma unroll
        for (int u = 0; u < TL_ROWS / 2; ++u) {
            if (u + 1 < TL_ROWS / 2) {
#pragma unroll
                for (int q = 0; q < TG_MAXT; ++q) {
                    va[(u + 1) & 1][q] = c_s[la[q] + 2 * (u + 1) * Ka];
                    vb[(u + 1) & 1][q] = c_s[lb[q] + 2 * (u + 1) * Kb];
                }
            }
            __builtin_amdgcn_sched_barrier(0);                          // (hipcc would sink the reads behind this step's MFMAs and wait for them at once)
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q)
                if (q < nq && !(TG_DIAG & 1))                           // wave-uniform: tiles this wave does not have cost no matrix-pipe time
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(offA[q] >= 0 ? va[u & 1][q] : 0.0f, offB[q] >= 0 ? vb[u & 1][q] : (offB[q] == -2 ? 1.0f : 0.0f),
                                                                  acc[q], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < TG_MAXT; ++q) drain += acc[q][15];          // every MFMA of the chunk has completed
        __builtin_amdgcn_sched_barrier(0);
        if (more) stash(sm + (cur ^ 1) * bufsz, preA);
        __syncthreads();
        cur ^= 1;
    }
    if (drain == 1.2345e-30f && P == nullptr) P[0] = drain;             // keeps `drain` alive; never true
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        const int t = wave + q * TG_WAVES;
        if (t >= ntiles) continue;
        const int rb = t / ncb, cb = t - rb * ncb;
        const int col = cb * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
            if (row < Ka && col < Kc) P[((size_t)blockIdx.x * Ka + row) * Kc + col] = acc[q][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ fp16-split variant (-DTVR_GEMM_F32=0; NOT the default)
// Same reduction on v_mfma_f32_32x32x16_f16 with both operands split into fp16 hi + lo and three products per 16-row step (tvr_mfma.h;
// error ~2^-22 relative, fp32-grade): 96 MFMA cycles per 16 rows and tile instead of 512, which moves the kernel from the fp32 matrix
// pipe onto memory.  A wave owns one 32-column block of A (or of B when A has more than four) and walks the blocks of the other
// operand, so every fragment is loaded and split once per wave: lane (col, h) reads rows m + 8h .. m + 8h + 7 of its column — the 8 k
// values of an MFMA fragment — with 8 row-coalesced dword loads.
// Measured and rejected as the default: 0.227 vs 0.26 ms at M = 3.6e5 and 0.8 vs 1.1 ms at M = 2.1e6 (the kernel is on memory and launch
// latency, not on the fp32 matrix pipe), and fp16's exponent range is wrong for gradients — dY entries below 6e-8 vanish in the split and a
// NerfPlusPlus background gradient came out 6e-3 off (tests/test_gpu_npp.py).  Forward activations are O(1); gradients are not.
// (An fp32 kernel with this wave-owns-a-block fragment sharing was also tried: 0.63 vs 0.26 ms at 356 000 x 128 x 128 — slower, removed.)
#ifndef TVR_GEMM_F32
#define TVR_GEMM_F32 1
#endif
#ifndef TVR_GEMM_LDS
#define TVR_GEMM_LDS 1
#endif

struct Rows8 {
    float v[8];
};
__device__ __forceinline__ Rows8 fetch8(const float *__restrict__ X, int ld, int col, long long m, long long m1, int h)
{
    Rows8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long row = m + 8 * h + j;
        const bool in = row < m1 && col >= 0;
        const float v = X[(in ? row : m1 - 1) * ld + (col >= 0 ? col : 0)];      // unconditional load, clamped address (see above)
        r.v[j] = in ? v : 0.0f;
    }
    return r;
}

__global__ __launch_bounds__(64 * TG_WAVES) void gemm_tn_f16_kernel(const float *__restrict__ A, const int lda, const int Ka,
                                                                     const float *__restrict__ B, const int ldb, const int Kb,
                                                                     const long long M_cap, float *__restrict__ P, const long long rpb_host, const unsigned *__restrict__ m_dev)
{
    // m_dev: the row count lives on the device (M_cap = capacity, the grid is sized for it); the slabs are then cut from the true count here
    const long long M = m_dev ? ((long long)*m_dev < M_cap ? (long long)*m_dev : M_cap) : M_cap;
    long long rows_per_block = rpb_host;
    if (m_dev) {
        rows_per_block = ((M + (long long)gridDim.x - 1) / (long long)gridDim.x + 15) / 16 * 16;
        if (rows_per_block < 64) rows_per_block = 64;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int nrb = (Ka + 31) >> 5, ncb = (Kb + 31) >> 5;
    // own: the operand whose 32-column block this wave keeps; oth: the operand whose blocks it walks (at most TG_MAXT of them)
    const bool own_is_a = nrb <= TG_WAVES;
    const float *__restrict__ Xo = own_is_a ? A : B, *__restrict__ Xw = own_is_a ? B : A;
    const int ldo = own_is_a ? lda : ldb, ldw = own_is_a ? ldb : lda, Ko = own_is_a ? Ka : Kb, Kw = own_is_a ? Kb : Ka;
    const int nown = own_is_a ? nrb : ncb, nwalk = own_is_a ? ncb : nrb;
    // with fewer than four own blocks the waves form groups that share an own block and split the walk: own block = wave % nown, and
    // the wave's q-th walk block is group + q * groups
    const int groups = TG_WAVES / nown, own_blk = wave % nown, group = wave / nown;
    const bool active = group < groups;
    const int col_o = (active && own_blk * 32 + i < Ko) ? own_blk * 32 + i : -1;
    int col_w[TG_MAXT], blk_w[TG_MAXT];
    f32x16 acc[TG_MAXT];
#pragma unroll
    for (int q = 0; q < TG_MAXT; ++q) {
        blk_w[q] = (active && group + q * groups < nwalk) ? group + q * groups : -1;
        col_w[q] = (blk_w[q] >= 0 && blk_w[q] * 32 + i < Kw) ? blk_w[q] * 32 + i : -1;
        acc[q] = f32x16{0};
    }
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = m0 >= M ? m0 : (m0 + rows_per_block < M ? m0 + rows_per_block : M);
    if (active) {
        Rows8 ro = fetch8(Xo, ldo, col_o, m0, m1, h), rw[TG_MAXT];
#pragma unroll
        for (int q = 0; q < TG_MAXT; ++q) rw[q] = fetch8(Xw, ldw, col_w[q], m0, m1, h);
        float drain = 0.0f;
        for (long long m = m0; m < m1; m += 16) {
            // loads of the next 16 rows first (into fresh registers), then the splits and MFMAs of the current ones
            const Rows8 no = fetch8(Xo, ldo, col_o, m + 16, m1, h);
            Rows8 nw[TG_MAXT];
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) nw[q] = fetch8(Xw, ldw, col_w[q], m + 16, m1, h);
            const Frag fo = split8(ro.v);
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) {
                if (blk_w[q] >= 0) {
                    const Frag fw = split8(rw[q].v);
                    // C tile = (A block)^T (B block): the A fragment is the MFMA's first operand whichever operand this wave owns
                    const Frag &fa = own_is_a ? fo : fw, &fb = own_is_a ? fw : fo;
                    acc[q] = MFMAH(fa.lo, fb.hi, acc[q]);
                    acc[q] = MFMAH(fa.hi, fb.lo, acc[q]);
                    acc[q] = MFMAH(fa.hi, fb.hi, acc[q]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            drain += acc[0][15];                                    // drains the wave's MFMAs before the next iteration reuses registers
#pragma unroll
            for (int q = 1; q < TG_MAXT; ++q) drain += acc[q][15];
            __builtin_amdgcn_sched_barrier(0);
            ro = no;
#pragma unroll
            for (int q = 0; q < TG_MAXT; ++q) rw[q] = nw[q];
        }
        if (drain == 1.2345e-30f && P == nullptr) P[0] = drain;     // keeps `drain` alive; never true
#pragma unroll
        for (int q = 0; q < TG_MAXT; ++q) {
            if (blk_w[q] < 0) continue;
            const int rb = own_is_a ? own_blk : blk_w[q], cb = own_is_a ? blk_w[q] : own_blk;
            const int col = cb * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < Ka && col < Kb) P[((size_t)blockIdx.x * Ka + row) * Kb + col] = acc[q][r];
            }
        }
    }
}

// C[e] = sum over the slabs in a fixed order: 16 lanes per element (lane l adds slabs l, l+16, ... in order), then a fixed butterfly
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float *__restrict__ P, const int n_slabs, const int n, float *__restrict__ C,
                                                             const float *__restrict__ scale)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t >> 4, l = t & 15;
    float s = 0.0f;
    if (e < n)
        for (int b = l; b < n_slabs; b += 16) s = s + P[(size_t)b * n + e];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s = s + __shfl_xor(s, off);
    if (e < n && l == 0) C[e] = scale ? s / *scale : s;               // (a power of two: the division is exact)
}

// the same for a [Ka, Kb + 1] product whose last column is the bias gradient: columns < Kb go to C (row stride Kb), column Kb to bias
__global__ __launch_bounds__(256) void gemm_tn_reduce_bias_kernel(const float *__restrict__ P, const int n_slabs, const int Ka, const int Kb, float *__restrict__ C,
                                                                  float *__restrict__ bias, const float *__restrict__ scale)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t >> 4, l = t & 15, n = Ka * (Kb + 1);
    float s = 0.0f;
    if (e < n)
        for (int b = l; b < n_slabs; b += 16) s = s + P[(size_t)b * n + e];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s = s + __shfl_xor(s, off);
    if (scale) s = s / *scale;
    if (e < n && l == 0) {
        const int row = e / (Kb + 1), col = e - row * (Kb + 1);
        if (col < Kb) C[row * Kb + col] = s;
        else bias[row] = s;
    }
}

static int cu_count()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus;
}

static void gemm_tn_shape(long long M, long long &grid, long long &rpb)
{
    grid = 2 * cu_count();
    rpb = ((M + grid - 1) / grid + 15) / 16 * 16;                  // multiple of the 16-row step
    if (rpb < 64) rpb = 64;
    grid = M > 0 ? (M + rpb - 1) / rpb : 0;
}

size_t gemm_tn_scratch_bytes(int Ka, int Kb, long long M)
{
    long long grid, rpb;
    gemm_tn_shape(M, grid, rpb);
    return (size_t)(grid > 0 ? grid : 1) * Ka * Kb * sizeof(float);
}

hipError_t launch_gemm_tn(const float *A, int lda, int Ka, const float *B, int ldb, int Kb, long long M, float *C, float *scratch, hipStream_t stream,
                          const unsigned *m_dev, float *bias_out, const float *scale_f16)
{
    long long grid, rpb;
    gemm_tn_shape(M, grid, rpb);
    const int ones = bias_out ? 1 : 0;
    if (grid == 0) {
        hipError_t rc = hipMemsetAsync(C, 0, (size_t)Ka * Kb * sizeof(float), stream);
        if (rc == hipSuccess && bias_out) rc = hipMemsetAsync(bias_out, 0, (size_t)Ka * sizeof(float), stream);
        return rc;
    }
    if (ones && !(TVR_GEMM_F32 && TVR_GEMM_LDS && TL_ROWS * (Ka + Kb) <= 256 * TL_PF)) return hipErrorInvalidValue;      // (only the LDS-staged kernel has the ones column)
    const int lds = 64 * 1024;                                     // unused; caps the CU at two workgroups (see header)
    hipError_t rc = hipFuncSetAttribute((const void *)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (rc != hipSuccess) return rc;
    if (TVR_GEMM_F32 && TVR_GEMM_LDS && TL_ROWS * (Ka + Kb) <= 256 * TL_PF) {          // (a 20 x 1 tile product, e.g. a column sum, has up to 672 columns: old kernel)
        const int lds2 = 2 * ((TL_ROWS * (Ka + Kb) + 3) & ~3) * (int)sizeof(float);       // <= 40 KB
        // 16-B loads need contiguous rows (a chunk is then one flat run of floats) and 16-B aligned chunk starts (TL_ROWS * K * 4 B is)
        const bool al = ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0;
        const int mode = (al && lda == Ka && ldb == Kb) ? 1 : ((al && !((Ka | Kb | lda | ldb) & 3)) ? 2 : 0);
        const bool f16 = scale_f16 != nullptr && Ka <= 32 * TG_WAVES && TVR_GEMM_SPLIT16;
        const dim3 g((unsigned)grid), b(64 * TG_WAVES);
#define TG_GO(MODE_, F16_) hipLaunchKernelGGL((gemm_tn_lds_kernel<MODE_, F16_>), g, b, lds2, stream, A, lda, Ka, B, ldb, Kb, M, scratch, rpb, m_dev, ones, scale_f16)
        if (f16) { if (mode == 1) TG_GO(1, true); else if (mode == 2) TG_GO(2, true); else TG_GO(0, true); }
        else { if (mode == 1) TG_GO(1, false); else if (mode == 2) TG_GO(2, false); else TG_GO(0, false); }
#undef TG_GO
        if (!f16) scale_f16 = nullptr;
    } else if (TVR_GEMM_F32) {
        scale_f16 = nullptr;                                        // (the direct-load kernels keep the fp32 products)
        hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)grid), dim3(64 * TG_WAVES), lds, stream, A, lda, Ka, B, ldb, Kb, M, scratch, rpb, m_dev);
    } else {
        scale_f16 = nullptr;
        rc = hipFuncSetAttribute((const void *)gemm_tn_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(gemm_tn_f16_kernel, dim3((unsigned)grid), dim3(64 * TG_WAVES), lds, stream, A, lda, Ka, B, ldb, Kb, M, scratch, rpb, m_dev);
    }
    const int n = Ka * (Kb + ones);
    if (ones) hipLaunchKernelGGL(gemm_tn_reduce_bias_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, stream, scratch, (int)grid, Ka, Kb, C, bias_out, scale_f16);
    else hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, stream, scratch, (int)grid, n, C, scale_f16);
    return hipGetLastError();
}
