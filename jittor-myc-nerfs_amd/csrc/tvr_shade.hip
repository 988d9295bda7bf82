// tvr_shade.hip — kernel 2 of the render path: appearance VM lookup + basis + positional encoding + 3-layer MLP.
//
// Work it replaces in the reference (paths relative to /root/reference/tensorf-myc/):
//   models/tensoRF.py:228-244 compute_appfeature (6 grid_samples + Linear(144->27, no bias)),
//   models/tensorBase.py:9-15 positional_encoding, :76-86 MLPRender_Fea.execute (150->128->128->3, sigmoid).
//
// One workgroup (4 waves) processes tiles of 64 queue entries:
//   gather : 12 lanes x float4 = one 192-B appearance texel per tap -> h[144] = plane*line, staged K-major in LDS
//   basis  : f[64x27]  = h[64x144] · basisT            v_mfma_f32_16x16x4_f32
//   PE     : in[150]   = [f, d, sin/cos(f 2^k), sin/cos(d 2^k)] written K-major to LDS straight from the accumulators
//   L1, L2 : 150->128, 128->128 (+bias, relu)          v_mfma_f32_32x32x2_f32, wave w owns output columns 32w..32w+31
//   L3     : 128->3 (+bias, sigmoid)                   v_mfma_f32_16x16x4_f32
// fp32-input MFMA is bit-identical to a k-ordered fmaf chain (MI355X_MICROARCH §Matrix cores), so the MLP carries no
// reduced-precision error against the 1e-3 RGB parity bar.  Activations live K-major ([k][entry], row stride 65) so
// MFMA A-operand reads and accumulator write-backs are bank-conflict-free; weights stream K-major from L2.
#include "tvr_device.h"
#include "tvr_kernels.h"

#define SH_THREADS 256
#define SH_TM 64
#define SH_LD 65
#define SH_X_FLOATS (TVR_KAPP * SH_LD)   // 9360: h (144 rows) / act1 (128 rows)
#define SH_Y_FLOATS (152 * SH_LD)        // 9880: mlp_in (150 rows) / act2 (128 rows)
#define SH_LDS_BYTES ((SH_X_FLOATS + SH_Y_FLOATS + 2 * SH_TM * 4) * 4)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// positional-encoding rows of feature channel c with value v, entry column e  (tensorBase.py:9-15: index c*F+k)
__device__ __forceinline__ void pe_rows_feat(float *Y, int c, int e, float v)
{
    Y[c * SH_LD + e] = v;
    const float v2 = v * 2.0f;
    Y[(30 + 2 * c) * SH_LD + e] = sinf(v);
    Y[(31 + 2 * c) * SH_LD + e] = sinf(v2);
    Y[(84 + 2 * c) * SH_LD + e] = cosf(v);
    Y[(85 + 2 * c) * SH_LD + e] = cosf(v2);
}

// one 128-wide hidden layer on a 64-entry tile: out[n][e] = relu(sum_k in[k][e] * WT[k][n] + b[n]);  wave w -> n in [32w, 32w+32)
template <int K>
__device__ __forceinline__ void hidden_layer(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ WT,
                                             const float *__restrict__ bias, int wave, int lane)
{
    f32x16 c0 = {0}, c1 = {0};
    const int half = lane >> 5, l31 = lane & 31;
    const float *wp = WT + half * TVR_FEATC + wave * 32 + l31;
    const float *ap = in + half * SH_LD + l31;
#pragma unroll 5
    for (int kk = 0; kk < K / 2; ++kk) {
        const float b = wp[(size_t)kk * 2 * TVR_FEATC];
        const float a0 = ap[kk * 2 * SH_LD];
        const float a1 = ap[kk * 2 * SH_LD + 32];
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, c1, 0, 0, 0);
    }
    const int n = wave * 32 + l31;
    const float bb = bias[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        out[n * SH_LD + row] = fmaxf(c0[r] + bb, 0.0f);
        out[n * SH_LD + 32 + row] = fmaxf(c1[r] + bb, 0.0f);
    }
}

template <int SRC, int DST>
__global__ __launch_bounds__(SH_THREADS, 2) void shade_kernel(const SceneDev sc, const ShadeArgs a)
{
    extern __shared__ float lds[];
    float *X = lds;
    float *Y = lds + SH_X_FLOATS;
    float *en = Y + SH_Y_FLOATS;          // [64][4] xyz_norm
    float *ed = en + SH_TM * 4;           // [64][4] view dir
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long n_total = (SRC == SH_SRC_QUEUE) ? (long long)(*a.counter) : a.n;
    const long long n_tiles = (n_total + SH_TM - 1) / SH_TM;

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long e0 = tile * SH_TM;
        // ---- phase 0: entry positions / view directions ----
        if (tid < SH_TM) {
            const long long e = e0 + tid;
            float4 pn = make_float4(0.f, 0.f, 0.f, 0.f), dv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < n_total) {
                if (SRC == SH_SRC_QUEUE) {
                    pn = a.q_pos[e];
                    const float *rp = a.rays + (size_t)a.q_ray[e] * 6 + 3;
                    dv = make_float4(rp[0], rp[1], rp[2], 0.f);
                } else if (SRC == SH_SRC_XYZ) {
                    pn = make_float4(a.xyz[e * 3], a.xyz[e * 3 + 1], a.xyz[e * 3 + 2], 0.f);
                } else {
                    dv = make_float4(a.viewdirs[e * 3], a.viewdirs[e * 3 + 1], a.viewdirs[e * 3 + 2], 0.f);
                }
            }
            *(float4 *)(en + tid * 4) = pn;
            *(float4 *)(ed + tid * 4) = dv;
        }
        __syncthreads();

        if (SRC != SH_SRC_FEAT) {
            // ---- phase 1: gather 64 entries x 3 planes x 12 float4-quads ----
#pragma unroll 3
            for (int it = 0; it < (SH_TM * 36) / SH_THREADS; ++it) {
                const int item = tid + SH_THREADS * it;
                const int ent = item / 36, rem = item - ent * 36;
                const int pl = rem / 12, q = rem - pl * 12;
                const int ax = kMat[pl][0], bx = kMat[pl][1], vx = kVec[pl];
                const float fx = unnorm(en[ent * 4 + ax], sc.gm1[ax]);
                const float fy = unnorm(en[ent * 4 + bx], sc.gm1[bx]);
                const float fl = unnorm(en[ent * 4 + vx], sc.gm1[vx]);
                float4 h;
                if (SRC == SH_SRC_QUEUE) {
                    const float x0 = floorf(fx), y0 = floorf(fy), l0 = floorf(fl);
                    h = vm_term<12, false>(sc.aplane[pl], sc.aline[pl], sc.grid[ax], sc.grid[bx], sc.grid[vx], (int)x0, (int)y0, (int)l0,
                                           fx - x0, fy - y0, fl - l0, q);
                } else {
                    const float x0 = floorf(fminf(fmaxf(fx, -2.0f), sc.gm1[ax] + 2.0f));
                    const float y0 = floorf(fminf(fmaxf(fy, -2.0f), sc.gm1[bx] + 2.0f));
                    const float l0 = floorf(fminf(fmaxf(fl, -2.0f), sc.gm1[vx] + 2.0f));
                    h = vm_term<12, true>(sc.aplane[pl], sc.aline[pl], sc.grid[ax], sc.grid[bx], sc.grid[vx], (int)x0, (int)y0, (int)l0,
                                          fx - x0, fy - y0, fl - l0, q);
                }
                float *xp = X + (pl * TVR_CA + q * 4) * SH_LD + ent;
                xp[0] = h.x;
                xp[SH_LD] = h.y;
                xp[2 * SH_LD] = h.z;
                xp[3 * SH_LD] = h.w;
            }
            __syncthreads();

            // ---- phase 2: basis  f[16 entries of wave w][32] = h · basisT  (two 16x16 N-tiles) ----
            f32x4 f0 = {0}, f1 = {0};
            {
                const int kq = lane >> 4, l15 = lane & 15;
                const float *ap = X + kq * SH_LD + wave * 16 + l15;
                const float *bp = sc.basisT + kq * 32 + l15;
#pragma unroll 6
                for (int kk = 0; kk < TVR_KAPP / 4; ++kk) {
                    const float av = ap[kk * 4 * SH_LD];
                    const float b0 = bp[kk * 4 * 32], b1 = bp[kk * 4 * 32 + 16];
                    f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, f0, 0, 0, 0);
                    f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, f1, 0, 0, 0);
                }
            }
            const int col = lane & 15, rbase = wave * 16 + (lane >> 4) * 4;
            if (DST == SH_DST_FEAT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long e = e0 + rbase + r;
                    if (e < n_total) {
                        a.out[e * TVR_APPDIM + col] = f0[r];
                        if (col + 16 < TVR_APPDIM) a.out[e * TVR_APPDIM + col + 16] = f1[r];
                    }
                }
                __syncthreads();
                continue;
            }
            // ---- phase 3a: PE of the features straight from the accumulators ----
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pe_rows_feat(Y, col, rbase + r, f0[r]);
                if (col + 16 < TVR_APPDIM) pe_rows_feat(Y, col + 16, rbase + r, f1[r]);
            }
        } else {
            for (int idx = tid; idx < SH_TM * TVR_APPDIM; idx += SH_THREADS) {
                const int ent = idx / TVR_APPDIM, c = idx - ent * TVR_APPDIM;
                const long long e = e0 + ent;
                pe_rows_feat(Y, c, ent, e < n_total ? a.feats[e * TVR_APPDIM + c] : 0.0f);
            }
        }
        // ---- phase 3b: view direction rows 27..29 and its PE rows 138..149 ----
        if (tid < SH_TM * 3) {
            const int ent = tid / 3, c = tid - ent * 3;
            const float v = ed[ent * 4 + c], v2 = v * 2.0f;
            Y[(27 + c) * SH_LD + ent] = v;
            Y[(138 + 2 * c) * SH_LD + ent] = sinf(v);
            Y[(139 + 2 * c) * SH_LD + ent] = sinf(v2);
            Y[(144 + 2 * c) * SH_LD + ent] = cosf(v);
            Y[(145 + 2 * c) * SH_LD + ent] = cosf(v2);
        }
        __syncthreads();

        // ---- phase 4/5: hidden layers ----
        hidden_layer<TVR_NIN>(Y, X, sc.W1T, sc.b1, wave, lane);
        __syncthreads();
        hidden_layer<TVR_FEATC>(X, Y, sc.W2T, sc.b2, wave, lane);
        __syncthreads();

        // ---- phase 6: output layer, 16 entries per wave ----
        {
            f32x4 o = {0};
            const int kq = lane >> 4, l15 = lane & 15;
            const float *ap = Y + kq * SH_LD + wave * 16 + l15;
            const float *bp = sc.W3T + kq * 16 + l15;
#pragma unroll 8
            for (int kk = 0; kk < TVR_FEATC / 4; ++kk)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * 4 * SH_LD], bp[kk * 4 * 16], o, 0, 0, 0);
            const int col = l15, rbase = wave * 16 + kq * 4;
            if (col < 3) {
                const float bb = sc.b3[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long e = e0 + rbase + r;
                    if (e < n_total) {
                        const float v = sigmoid_f(o[r] + bb);
                        if (DST == SH_DST_QUEUE) ((float *)a.q_pos)[e * 4 + col] = v;
                        else a.out[e * 3 + col] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (a.stats && SRC == SH_SRC_QUEUE && blockIdx.x == 0 && tid == 0)
        atomicAdd((unsigned long long *)&a.stats[TVR_STAT_APP], (unsigned long long)n_total);
}

template <int SRC, int DST>
static hipError_t launch_shade_t(const SceneDev &sc, const ShadeArgs &a, hipStream_t stream)
{
    (void)hipFuncSetAttribute((const void *)shade_kernel<SRC, DST>, hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES);
    unsigned grid = 512;       // 2 workgroups per CU (LDS-limited), persistent over tiles
    if (SRC != SH_SRC_QUEUE) {
        const long long tiles = (a.n + SH_TM - 1) / SH_TM;
        if (tiles < grid) grid = (unsigned)(tiles > 0 ? tiles : 1);
    }
    hipLaunchKernelGGL((shade_kernel<SRC, DST>), dim3(grid), dim3(SH_THREADS), SH_LDS_BYTES, stream, sc, a);
    return hipGetLastError();
}

hipError_t launch_shade(const SceneDev &sc, int src, int dst, const ShadeArgs &a, hipStream_t stream)
{
    if (src == SH_SRC_QUEUE && dst == SH_DST_QUEUE) return launch_shade_t<SH_SRC_QUEUE, SH_DST_QUEUE>(sc, a, stream);
    if (src == SH_SRC_XYZ && dst == SH_DST_FEAT) return launch_shade_t<SH_SRC_XYZ, SH_DST_FEAT>(sc, a, stream);
    if (src == SH_SRC_FEAT && dst == SH_DST_RGB) return launch_shade_t<SH_SRC_FEAT, SH_DST_RGB>(sc, a, stream);
    return hipErrorInvalidValue;
}

// ---- scene packing (reference layout -> channels-last, zero-padded) ----
// in (C,H,W) -> out [H+1][W+1][C]; a line is the W == 1 case written as [H+1][C] by passing W = 0 pad off.
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

// in [n_in][K] row-major -> out [K][n_out], zero for n >= n_in
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float *__restrict__ in, float *__restrict__ out, int n_in, int K, int n_out)
{
    const int total = K * n_out;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i % n_out, k = i / n_out;
        out[i] = n < n_in ? in[(size_t)n * K + k] : 0.0f;
    }
}

// planes: W>1 -> padded row length W+1; lines (W == 1): packed [H+1][C], no x padding
hipError_t launch_pack_plane(const float *in, float *out, int C, int H, int W, hipStream_t stream)
{
    const int Wp = (W == 1) ? 1 : W + 1;
    const long long total = (long long)(H + 1) * Wp * C;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_plane_kernel, dim3(grid), dim3(256), 0, stream, in, out, C, H, W, Wp);
    return hipGetLastError();
}

hipError_t launch_transpose_pad(const float *in, float *out, int n_in, int K, int n_out, hipStream_t stream)
{
    unsigned grid = (unsigned)((K * n_out + 255) / 256);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(grid), dim3(256), 0, stream, in, out, n_in, K, n_out);
    return hipGetLastError();
}
