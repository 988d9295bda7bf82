// tvr_mlp_train.hip — backward of the appearance network of the training step (tensorf-myc/train.py:225-261 through
// models/tensoRF.py:244 `basis_mat` and models/tensorBase.py:76-86 `MLPRender_Fea.execute`) as register-resident MFMA chains.
//
// The forward of a training step is the inference shade kernel itself (tvr_shade.hip, SRC_H / DST_TRAIN: same instructions, so a training
// forward and an evaluation render agree bit for bit); it leaves  features [m,32], relu(layer 1) [m,128], relu(layer 2) [m,128], rgb [m,3].
// Here, per 32-entry tile and wave (entry = MFMA column, lane (e, h) as in the shade kernel):
//   d_out  = grad_rgb * rgb * (1 - rgb)                                        sigmoid', VALU
//   dH2^T  = mask(relu 2) . W3^T d_out^T                       [128 x 32e]     K = 3: plain fp32 FMAs, W3 as fp32 in LDS
//   dH1^T  = mask(relu 1) . W2^T dH2^T                         [128 x 32e]     MFMA: A = W2^T image in LDS, B = the dH2 registers in place
//   dX^T   = W1^T dH1^T                                        [160 x 32e]     MFMA: row 32 t + c of W1^T's image is derived value t of base value c,
//                                                                              so lane (e, h) ends up with the five slot gradients of its own 16 base values
//   dF     = dX0 + cos v dX1 + 2 cos 2v dX2 - sin v dX3 - 2 sin 2v dX4                 derivative of [v, sin v, sin 2v, cos v, cos 2v]
// and a second, small kernel:  dh^T = Bas^T dF^T [144 x 32e]  (the gradient that tvr_app_h_backward scatters into planes and lines).
// Stored for the weight gradients (tall-skinny reductions, tvr_gemm_tn): d_out [m,4], dH2 [m,128], dH1 [m,128], dF [m,32].
// Arithmetic: fp16 hi/lo split MFMA, 3 products, fp32 accumulate, as in the forward (tvr_mfma.h).  GRADIENT RANGE: operands pass through fp16,
// so |dH| must stay below 65 504 and parts below 6e-8 of the largest... are flushed: the gradients are therefore scaled per call by `gscale`
// (a power of two chosen by the host from max |grad_rgb|) on entry and unscaled on exit — the MSE gradients of a 4096-ray batch are O(1e-4).
// Phase discipline as in tvr_shade.hip: all loads of a tile are issued and waited for before its first MFMA (stores are not loads).
#include "tvr_device.h"
#include "tvr_kernels.h"
#include "tvr_mfma.h"

#define TI_ROW 272                                   // 128 k positions * 2 B + 16 B pad (conflict-free ds_read_b128, as the forward image)
#define TI_W2T_H 0
#define TI_W2T_L (128 * TI_ROW)
#define TI_W1T_H (2 * 128 * TI_ROW)
#define TI_W1T_L (TI_W1T_H + 160 * TI_ROW)
#define TI_W3 (TI_W1T_H + 2 * 160 * TI_ROW)          // W3 [3][128] fp32
#define TI_LDS_BYTES (TI_W3 + 3 * 128 * 4)           // 158 208 B
#define TI_BAST (TI_LDS_BYTES)                       // Bas^T fragments [5 row blocks][KS k-steps][2 halves][32 rows][hi 8 | lo 8] fp16 (not copied to LDS);
                                                     // KS = 2 (27 features, padded to 32); REFTensoRF: KS = 3, k 32..39 = the eight head outputs
#define TI_SCAL (TI_BAST + 5 * 3 * 2 * 32 * 32)      // REFTensoRF: {max |dg8| bits, the heads' own gradient scale} (the normal's gradient carries 1 / |n|: its range is not the network's)
#define TI_BYTES (TI_SCAL + 64)                      // 188 992 B (sized for KS = 3)
// Round 6 — scenes with more than two encoding frequencies (TVR_GEN_*, tvr_device.h; tensorBase.py:141-145): W1^T does not fit the LDS (13 derived values x 32 base
// rows x 128 units, hi + lo: 208 KB).  Its fragments live in global memory behind the image in the order the backward reads them — [slot t (13)][k-step s (8)][hi | lo]
// [lane (64)] uint4, lane (e, h) = row e (base value), k 16 s + 8 h .. + 7 — and every wave streams them from L2, half a slot (8 KB) ahead of its MFMAs.
#define TI_W1G ((TI_BYTES + 255) / 256 * 256)
#define TI_W1G_BYTES (TVR_GEN_T * 8 * 2 * 1024)      // 212 992 B
#define TI_BYTES_ALL (TI_W1G + TI_W1G_BYTES)

#define MT_WAVES 8
#define MT_THREADS (64 * MT_WAVES)

// hidden unit of k position kpos of a hidden-layer k-step (the accumulator-as-operand order, tvr_shade.hip pack mode 1)
__device__ __forceinline__ int unit_of_kpos(int kpos)
{
    const int s = kpos >> 4, hh = (kpos >> 3) & 1, j = kpos & 7;
    return 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
}

// REFTensoRF's four heads on h (REFTensoRF.py:86-96), in the order of the shade kernel's second row block: normal 0..2, specular 3, diffuse 4..6, rho 7
struct HeadPtrs { const float *W[4]; };              // normal [3,144], diffuse [3,144], specular [1,144], rho [1,144]; all NULL: TensorVMSplit
// basis_mat as the reference holds it: [27][k_app] with plane p's n[p] <= 48 components at columns off[p] .. (tensoRF.py:150, 228-244); the kernels' channel 48 p + c
struct AppCols { int n[3], off[3], k_app; };
__device__ __forceinline__ float head_weight(const HeadPtrs &hp, int i, int ch)
{
    return i < 3 ? hp.W[0][i * TVR_KAPP + ch] : (i == 3 ? hp.W[2][ch] : (i < 7 ? hp.W[1][(i - 4) * TVR_KAPP + ch] : hp.W[3][ch]));
}

// one thread per element of the three transposed images.  ref: W1 is MLPRender_Fea_Ref's [128,151] (REFTensoRF.py:9-16: every input index moves up
// by one, base row 30's plain slot is input 0 = -dot) and Bas^T gets a third k-step holding the heads' weights.
__global__ __launch_bounds__(256) void pack_train_image_kernel(const float *__restrict__ W1, const float *__restrict__ W2, const float *__restrict__ W3,
                                                               const float *__restrict__ Bas, const HeadPtrs hp, const int ref, const int gen, const AppCols ac,
                                                               const int fc, const int fea_pe, const int view_pe, unsigned char *__restrict__ img)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *(unsigned *)(img + TI_SCAL) = 0u;
    const int KB = ref ? 48 : 32;
    const int n_w2 = 128 * 128, n_w1 = 160 * 128, n_w3 = 3 * 128, n_b = 160 * KB;
    if (i < n_w2) {
        const int row = i >> 7, kpos = i & 127;                                   // row = layer-1 unit, k = layer-2 unit
        unsigned hi, lo;
        const int u2 = unit_of_kpos(kpos);                                        // (fc < 128: W2 is [fc, fc]; units that do not exist are zero rows and columns)
        split2((u2 < fc && row < fc) ? W2[(size_t)u2 * fc + row] : 0.0f, 0.0f, hi, lo);
        ((unsigned short *)(img + TI_W2T_H + row * TI_ROW))[kpos] = (unsigned short)hi;
        ((unsigned short *)(img + TI_W2T_L + row * TI_ROW))[kpos] = (unsigned short)lo;
    } else if (i < n_w2 + n_w1) {
        const int k = i - n_w2, row = k >> 7, kpos = k & 127;                     // row = 32 t + c: derived value t of base value c; k = layer-1 unit
        const int c = row & 31, t = row >> 5;
        int idx = ref ? ref_in_index(c, t) : ref_in_index(c, t, fea_pe, view_pe);      // (fewer than two frequencies: the slots that do not exist have no column, weight 0)
        if (ref) idx = (c == TVR_APPDIM + 3) ? (t == 0 ? 0 : -1) : (idx >= 0 ? idx + 1 : -1);
        // (gen: W1 is [fc, 30 + 54 fea_pe + 6 view_pe] and goes to the streamed image, pack_train_w1gen_kernel; this region stays zero)
        const int u1 = unit_of_kpos(kpos), nin1 = ref ? TVR_NIN_REF : TVR_APPDIM + 3 + 2 * TVR_APPDIM * fea_pe + 6 * view_pe;
        const float w = (idx >= 0 && !gen && u1 < fc) ? W1[(size_t)u1 * nin1 + idx] : 0.0f;
        unsigned hi, lo;
        split2(w, 0.0f, hi, lo);
        ((unsigned short *)(img + TI_W1T_H + row * TI_ROW))[kpos] = (unsigned short)hi;
        ((unsigned short *)(img + TI_W1T_L + row * TI_ROW))[kpos] = (unsigned short)lo;
    } else if (i < n_w2 + n_w1 + n_w3) {
        const int k = i - n_w2 - n_w1;
        const int c3 = k >> 7, u3 = k & 127;
        ((float *)(img + TI_W3))[k] = u3 < fc ? W3[c3 * fc + u3] : 0.0f;
    } else if (i < n_w2 + n_w1 + n_w3 + n_b) {
        const int k = i - n_w2 - n_w1 - n_w3, row = k / KB, kpos = k - row * KB;  // row = channel (144, padded to 160), k = feature (27, padded to 32) [+ 8 heads, padded to 16]
        const int f = unit_of_kpos(kpos);
        float w = 0.0f;
        if (row < TVR_KAPP) {
            const int pl = row / TVR_CA, c = row - pl * TVR_CA;                   // (a scene with fewer components: zero rows behind plane pl's own, as in the packed scene)
            if (f < TVR_APPDIM) w = c < ac.n[pl] ? Bas[(size_t)f * ac.k_app + ac.off[pl] + c] : 0.0f;
            else if (ref && f >= 32 && f < 40) w = head_weight(hp, f - 32, row);
        }
        unsigned hi, lo;
        split2(w, 0.0f, hi, lo);
        const int s = kpos >> 4, hh = (kpos >> 3) & 1, j = kpos & 7, rb = row >> 5, r = row & 31;
        unsigned short *o = (unsigned short *)(img + TI_BAST) + ((size_t)(((rb * (ref ? 3 : 2) + s) * 2 + hh) * 32 + r)) * 16;
        o[j] = (unsigned short)hi;
        o[8 + j] = (unsigned short)lo;
    }
}

// one thread per (slot t, k-step s, lane): the hi and the lo uint4 of the streamed W1^T image (TI_W1G)
__global__ __launch_bounds__(256) void pack_train_w1gen_kernel(const float *__restrict__ W1, const int fc, const int fea_pe, const int view_pe, unsigned char *__restrict__ img)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TVR_GEN_T * 8 * 64) return;
    const int lane = i & 63, s = (i >> 6) & 7, t = i >> 9;
    const int e = lane & 31, h = lane >> 5;
    const int nin = TVR_APPDIM + 3 + 2 * TVR_APPDIM * fea_pe + 6 * view_pe;
    const int idx = gen_in_index(e, t, fea_pe, view_pe);
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const int u0 = unit_of_kpos(16 * s + 8 * h + j), u1 = unit_of_kpos(16 * s + 8 * h + j + 1);
        split2((idx >= 0 && u0 < fc) ? W1[(size_t)u0 * nin + idx] : 0.0f, (idx >= 0 && u1 < fc) ? W1[(size_t)u1 * nin + idx] : 0.0f, hi[j >> 1], lo[j >> 1]);
    }
    uint4 *o = (uint4 *)(img + TI_W1G) + (size_t)((t * 8 + s) * 2) * 64 + lane;
    o[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    o[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

struct AFragN { uint4 h[5], l[5]; };

template <int NB>
__device__ __forceinline__ void load_afragn(AFragN &A, const unsigned char *WH, const unsigned char *WL, int off0)
{
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
        A.h[rb] = *(const uint4 *)(WH + off0 + rb * 32 * TI_ROW);
        A.l[rb] = *(const uint4 *)(WL + off0 + rb * 32 * TI_ROW);
    }
}

template <int NB>
__device__ __forceinline__ void mfma3xn(const AFragN &A, const Frag &b, f32x16 acc[NB])
{
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) acc[rb] = MFMAH(A.l[rb], b.hi, acc[rb]);
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) acc[rb] = MFMAH(A.h[rb], b.lo, acc[rb]);
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) acc[rb] = MFMAH(A.h[rb], b.hi, acc[rb]);
}

struct MlpBwdArgs {
    const float *grad_rgb, *rgb, *feats, *h1, *h2;   // [m,3], [m,3], [m,32], [m,128], [m,128]
    long long m;                                     // entries, or the buffers' capacity when m_dev is set
    const unsigned *m_dev;                           // optional device-side entry count
    const float *gscale;                             // device scalar: gradients are multiplied by this power of two on entry, by its inverse on exit
    float *d_out, *dh2, *dh1, *dfeats;               // [m,4], [m,128], [m,128], [m,32]
    const unsigned char *image;                      // TI_BYTES
    const float *g8;                                 // REFTensoRF: raw head outputs [m,8] (the tint scales the gradient entering the network)
    unsigned *sat;                                   // optional: set to 1 when a scaled gradient reaches fp16's largest finite value on its way into a product
};

// REF: MLPRender_Fea_Ref inside REFTensoRF.execute (REFTensoRF.py:229-232): the colour is relu(tint) * rgb_s + rgb_d, so the gradient entering the
// network is relu(tint) * grad_rgb and `rgb` is rgb_s; the direction rows 27..29 (the reflection) and row 30 (-dot) of layer 1's input are
// functions of h as well: their gradients come out in dfeats columns 27..30 (ref_heads_backward_kernel takes them to the heads).
// GEN (round 6): layer 1 has 13 derived values per base value (up to six frequencies); dX runs slot by slot over the streamed image TI_W1G, the gradient of the
// positional encoding is taken per slot — d sin(2^f v) = 2^f cos(2^f v), d cos(2^f v) = -2^f sin(2^f v), from the forward's own reduced argument — and summed into dF.
template <bool REF, bool GEN = false>
__global__ __launch_bounds__(MT_THREADS, MT_WAVES / 4) void mlp_train_backward_kernel(const MlpBwdArgs a)
{
    static_assert(!(REF && GEN), "REFTensoRF scenes have two encoding frequencies");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 31, h = lane >> 5;
    {
        const uint4 *src = (const uint4 *)a.image;
        for (int i = tid; i < TI_LDS_BYTES / 16; i += MT_THREADS) ((uint4 *)smem)[i] = src[i];
        __syncthreads();
    }
    const long long a_m = a.m_dev ? ((long long)*a.m_dev < a.m ? (long long)*a.m_dev : a.m) : a.m;
    const long long n_tiles = (a_m + 31) / 32;
    const float gscale = *a.gscale, inv_scale = 1.0f / gscale;
    float amax = 0.0f;                               // largest |operand| this lane split to fp16 (v_cvt_pkrtz saturates at 65 504 silently)
    for (long long tile = (long long)blockIdx.x * MT_WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * MT_WAVES) {
        const long long ent = tile * 32 + e;
        const bool live = ent < a_m;
        const long long le = live ? ent : a_m - 1;
        // ---------------------------------------------------------------- loads (all of this tile's) ----
        float g[3], o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { g[c] = a.grad_rgb[le * 3 + c]; o[c] = a.rgb[le * 3 + c]; }
        if (REF) {
            const float tint = fmaxf(a.g8[le * 8 + 3], 0.0f);
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = g[c] * tint;
        }
        unsigned long long m1 = 0ull, m2 = 0ull;     // bit 16 rb + r: relu(layer 1 / 2) of hidden unit 32 rb + acc_row(r, h) is positive
        const unsigned lrow128 = (unsigned)le * (TVR_FEATC * 4u) + 16u * (unsigned)h;
        {
            float4 t2[16], t1[16];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    t2[rb * 4 + q] = *(const float4 *)((const unsigned char *)a.h2 + (lrow128 + (unsigned)(128 * rb + 32 * q)));
                    t1[rb * 4 + q] = *(const float4 *)((const unsigned char *)a.h1 + (lrow128 + (unsigned)(128 * rb + 32 * q)));
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                m2 |= (unsigned long long)((t2[i].x > 0.f) | ((t2[i].y > 0.f) << 1) | ((t2[i].z > 0.f) << 2) | ((t2[i].w > 0.f) << 3)) << (4 * i);
                m1 |= (unsigned long long)((t1[i].x > 0.f) | ((t1[i].y > 0.f) << 1) | ((t1[i].z > 0.f) << 2) | ((t1[i].w > 0.f) << 3)) << (4 * i);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---------------------------------------------------------------- d_out, dH2 (VALU) ----
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) d[c] = live ? (g[c] * gscale) * (o[c] * (1.0f - o[c])) : 0.0f;
        if (live && h == 0) *(float4 *)(a.d_out + ent * 4) = make_float4(d[0] * inv_scale, d[1] * inv_scale, d[2] * inv_scale, 0.0f);
        f32x16 dh2[4];
        {
            unsigned w3a = (unsigned)(size_t)(smem + TI_W3) + 16u * (unsigned)h;
            asm volatile("" : "+v"(w3a));
            const float *W3 = (const float *)(const void __attribute__((address_space(3))) *)(size_t)w3a;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w0 = *(const float4 *)(W3 + 32 * rb + 8 * q), w1 = *(const float4 *)(W3 + 128 + 32 * rb + 8 * q),
                                 w2 = *(const float4 *)(W3 + 256 + 32 * rb + 8 * q);
                    const float x[4] = {__builtin_fmaf(w2.x, d[2], __builtin_fmaf(w1.x, d[1], w0.x * d[0])), __builtin_fmaf(w2.y, d[2], __builtin_fmaf(w1.y, d[1], w0.y * d[0])),
                                        __builtin_fmaf(w2.z, d[2], __builtin_fmaf(w1.z, d[1], w0.z * d[0])), __builtin_fmaf(w2.w, d[2], __builtin_fmaf(w1.w, d[1], w0.w * d[0]))};
#pragma unroll
                    for (int i = 0; i < 4; ++i) dh2[rb][4 * q + i] = ((m2 >> (16 * rb + 4 * q + i)) & 1ull) ? x[i] : 0.0f;
                }
        }
        const unsigned row128 = (unsigned)ent * (TVR_FEATC * 4u) + 16u * (unsigned)h;       // byte offset of this lane's first quad in a [m,128] matrix
        if (live) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4 *)((unsigned char *)a.dh2 + (row128 + (unsigned)(128 * rb + 32 * q))) =
                        make_float4(dh2[rb][4 * q] * inv_scale, dh2[rb][4 * q + 1] * inv_scale, dh2[rb][4 * q + 2] * inv_scale, dh2[rb][4 * q + 3] * inv_scale);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------------------------------------------------------- dH1 = mask . W2^T dH2 ----
        f32x16 dh1[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) dh1[rb] = f32x16{0};
        {
            const int rowoff = e * TI_ROW + h * 16;
            auto frag = [&](int s, Frag &b) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[j] = dh2[s >> 1][8 * (s & 1) + j]; amax = fmaxf(amax, fabsf(x[j])); }
                b = split8(x);
            };
            Frag bcur, bnxt;
            AFragN acur, anxt;
            frag(0, bcur);
            load_afragn<4>(acur, smem + TI_W2T_H, smem + TI_W2T_L, rowoff);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8) {
                    load_afragn<4>(anxt, smem + TI_W2T_H, smem + TI_W2T_L, rowoff + (s + 1) * 32);
                    frag(s + 1, bnxt);
                }
                mfma3xn<4>(acur, bcur, dh1);
                acur = anxt;
                bcur = bnxt;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh1[rb][r] = ((m1 >> (16 * rb + r)) & 1ull) ? dh1[rb][r] : 0.0f;
        if (live) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4 *)((unsigned char *)a.dh1 + (row128 + (unsigned)(128 * rb + 32 * q))) =
                        make_float4(dh1[rb][4 * q] * inv_scale, dh1[rb][4 * q + 1] * inv_scale, dh1[rb][4 * q + 2] * inv_scale, dh1[rb][4 * q + 3] * inv_scale);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GEN) {
            // ---------------------------------------------------------------- dX slot by slot, dF accumulated ----
            Frag bf[8];                                                        // dH1 as the eight B fragments every slot multiplies
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[j] = dh1[s >> 1][8 * (s & 1) + j]; amax = fmaxf(amax, fabsf(x[j])); }
                bf[s] = split8(x);
                asm volatile("" : "+v"(amax));                                 // taken HERE (left alone, hipcc sinks the max chain to the tile's end and keeps dH1's 64 registers alive for it)
            }
            __builtin_amdgcn_sched_barrier(0);                                 // (every dH1 register has been read: the MFMAs of dH1 have completed)
            float TR[16];                                                      // the forward's reduced arguments, in revolutions (tvr_shade.hip gen_frag)
            {
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 f4 = *(const float4 *)(a.feats + le * 32 + 8 * q + 4 * h);
                    v[4 * q] = f4.x; v[4 * q + 1] = f4.y; v[4 * q + 2] = f4.z; v[4 * q + 3] = f4.w;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float k = rintf(v[r] * 0.15915494309189535f);
                    float rr = __builtin_fmaf(k, -6.2831854820251465f, v[r]);
                    rr = __builtin_fmaf(k, 1.7484555e-7f, rr);
                    TR[r] = rr * 0.15915494309189535f;
                }
            }
            float df[16];
            // half slot c: uint4 (8 c + i) * 64 + lane, i = 2 k + {0: hi, 1: lo} of k-step 4 (c & 1) + k.  Addressing: the image's scalar base + a 32-bit offset per 4 KB (taken where it is
            // used) + an immediate < 4 KB — left to itself hipcc computes the 208 lane addresses once per kernel and spills them (578 VGPRs)
            unsigned loff = (unsigned)lane * 16u;
            asm volatile("" : "+v"(loff));
            auto ld_half = [&](int c, uint4 (&dst)[8]) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned cb = (unsigned)(TI_W1G + c * 8192 + q * 4096);
                    asm volatile("" : "+s"(cb));
                    const unsigned vo = cb + loff;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[4 * q + i] = *(const uint4 *)(a.image + ((size_t)vo + (size_t)(i * 1024)));
                }
            };
            uint4 A[2][8];
            ld_half(0, A[0]);
            f32x16 acc = f32x16{0};
            float drain = 0.0f;
#pragma unroll
            for (int c = 0; c < 2 * TVR_GEN_T; ++c) {
                const int t = c >> 1, half = c & 1;
                // the next half slot's fragments are requested BEFORE this one's MFMAs, into the registers the half slot before this one used: its MFMAs have
                // completed (each half slot ends with a read of its last accumulator, as tvr_gemm.hip's chunks do)
                if (c + 1 < 2 * TVR_GEN_T) ld_half(c + 1, A[(c + 1) & 1]);
                if (half == 0) acc = f32x16{0};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Frag &b = bf[4 * half + k];
                    acc = MFMAH(A[c & 1][2 * k + 1], b.hi, acc);
                    acc = MFMAH(A[c & 1][2 * k], b.lo, acc);
                    acc = MFMAH(A[c & 1][2 * k], b.hi, acc);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (half == 0) drain += acc[15];
                else {
                    const int f = t == 0 ? 0 : (t <= TVR_GEN_PE ? t - 1 : t - 1 - TVR_GEN_PE);
                    const float p2 = (float)(1 << f);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (t == 0) df[r] = acc[r];
                        else if (t <= TVR_GEN_PE) df[r] = __builtin_fmaf(p2 * __builtin_amdgcn_cosf(TR[r] * p2), acc[r], df[r]);
                        else df[r] = __builtin_fmaf(-p2 * __builtin_amdgcn_sinf(TR[r] * p2), acc[r], df[r]);
                        asm volatile("" : "+v"(df[r]));                        // taken HERE (left alone, hipcc defers the thirteen slots' sums to the tile's end and keeps their 208 accumulator registers)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (drain == 1.2345e-30f) amax = 65504.0f;                         // keeps `drain` alive; never true for a finite accumulator sum of this size
#pragma unroll
            for (int r = 0; r < 16; ++r) df[r] = acc_row(r, h) < TVR_APPDIM ? df[r] * inv_scale : 0.0f;
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4 *)(a.dfeats + ent * 32 + 8 * q + 4 * h) = make_float4(df[4 * q], df[4 * q + 1], df[4 * q + 2], df[4 * q + 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
        // ---------------------------------------------------------------- dX = W1^T dH1  (5 row blocks: block t = derived value t) ----
        f32x16 dx[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) dx[t] = f32x16{0};
        {
            const int rowoff = e * TI_ROW + h * 16;
            auto frag = [&](int s, Frag &b) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[j] = dh1[s >> 1][8 * (s & 1) + j]; amax = fmaxf(amax, fabsf(x[j])); }
                b = split8(x);
            };
            // (the ten weight fragments of a step are fetched in the step itself: 144 accumulator registers leave no room for a second set;
            //  the partner wave covers the LDS latency)
            Frag bcur;
            AFragN acur;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                load_afragn<5>(acur, smem + TI_W1T_H, smem + TI_W1T_L, rowoff + s * 32);
                frag(s, bcur);
                mfma3xn<5>(acur, bcur, dx);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---------------------------------------------------------------- dF: through the positional encoding ----
        // The features are fetched HERE, behind the matrix phase (144 accumulator registers leave no room to hold them across it): first every
        // lane reads block 4 of dX, whose last MFMA is the last one issued — that completes this tile's MFMAs (the pipe is in order) —, then
        // the loads are issued and waited for.
        float df[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            df[r] = dx[4][r] * inv_scale;                                      // block 4's last MFMA is the last one issued
            asm volatile("" : "+v"(df[r]) :: "memory");                        // computed HERE: nothing below (the loads) may move above it
        }
        __builtin_amdgcn_sched_barrier(0);
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 f4 = *(const float4 *)(a.feats + le * 32 + 8 * q + 4 * h);
            v[4 * q] = f4.x; v[4 * q + 1] = f4.y; v[4 * q + 2] = f4.z; v[4 * q + 3] = f4.w;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float k = rintf(v[r] * 0.15915494309189535f);                    // the forward's reduction (tvr_shade.hip sincos_pe)
            float rr = __builtin_fmaf(k, -6.2831854820251465f, v[r]);
            rr = __builtin_fmaf(k, 1.7484555e-7f, rr);
            const float tt = rr * 0.15915494309189535f;
            const float sn = __builtin_amdgcn_sinf(tt), cs = __builtin_amdgcn_cosf(tt);
            const float s2 = 2.0f * sn * cs, c2 = __builtin_fmaf(-2.0f * sn, sn, 1.0f);
            float acc = dx[0][r];
            acc = __builtin_fmaf(cs, dx[1][r], acc);
            acc = __builtin_fmaf(2.0f * c2, dx[2][r], acc);
            acc = __builtin_fmaf(-sn, dx[3][r], acc);
            const int row = acc_row(r, h);
            df[r] = (row < (REF ? TVR_APPDIM + 3 : TVR_APPDIM)) ? __builtin_fmaf(-2.0f * s2, df[r], acc * inv_scale)
                    : ((REF && row == TVR_APPDIM + 3) ? dx[0][r] * inv_scale : 0.0f);            // row 30 (-dot) enters layer 1 plainly only
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4 *)(a.dfeats + ent * 32 + 8 * q + 4 * h) = make_float4(df[4 * q], df[4 * q + 1], df[4 * q + 2], df[4 * q + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        }           // !GEN
    }
    if (a.sat && amax >= 65504.0f) atomicOr(a.sat, 1u);
}

// REFTensoRF: the heads' backward between the two kernels.  Per entry, from the gradients of the reflection / -dot inputs (dfeats columns 27..30),
// of the colour and (optional) of the -dot OUTPUT (the normal penalty of REFTensoRF.py:236-239 is taken by the host's autograd on that output):
//   n = G[0..2], nh = n / sqrt(max(n.n, 1e-30)), d = -view, dot = d.nh, refl = 2 dot nh - d, in0 = -dot            (REFTensoRF.py:217-229)
//   d dot = 2 nh.d_refl - d_in0 ;  d nh = 2 dot d_refl + d_dot d ;  d n = (d nh - nh (nh.d nh)) / |n|
//   d tint_raw = [tint_raw > 0] sum_c grad_rgb_c rgb_s_c ;  d rgb_d = grad_rgb ;  d rho = 0                          (REFTensoRF.py:232)
__global__ __launch_bounds__(256) void ref_heads_backward_kernel(const float *__restrict__ grad_rgb, const float *__restrict__ rgb_s, const float *__restrict__ g8,
                                                                 const float *__restrict__ viewdirs, const float *__restrict__ rays, const unsigned *__restrict__ q_ray,
                                                                 const float *__restrict__ dfeats, const float *__restrict__ grad_in0, const long long m_cap,
                                                                 const unsigned *__restrict__ m_dev, float *__restrict__ dg8, unsigned *__restrict__ amax_h)
{
    const long long m = m_dev ? ((long long)*m_dev < m_cap ? (long long)*m_dev : m_cap) : m_cap;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float vmax = 0.0f;
    if (i < m) {
    const float4 ga = *(const float4 *)(g8 + i * 8), gb = *(const float4 *)(g8 + i * 8 + 4);
    const float n[3] = {ga.x, ga.y, ga.z};
    const float nn = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
    const float nrm = sqrtf(fmaxf(nn, 1e-30f));
    const float nh[3] = {n[0] / nrm, n[1] / nrm, n[2] / nrm};
    const float *vp = q_ray ? rays + (size_t)q_ray[i] * 6 + 3 : viewdirs + i * 3;
    const float d[3] = {-vp[0], -vp[1], -vp[2]};
    const float dot = (d[0] * nh[0] + d[1] * nh[1]) + d[2] * nh[2];
    const float dr[3] = {dfeats[i * 32 + 27], dfeats[i * 32 + 28], dfeats[i * 32 + 29]};
    const float din0 = dfeats[i * 32 + 30] + (grad_in0 ? grad_in0[i] : 0.0f);
    const float ddot = 2.0f * ((nh[0] * dr[0] + nh[1] * dr[1]) + nh[2] * dr[2]) - din0;
    float dnh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dnh[k] = 2.0f * dot * dr[k] + ddot * d[k];
    float dn[3];
    if (nn > 1e-30f) {
        const float pr = (nh[0] * dnh[0] + nh[1] * dnh[1]) + nh[2] * dnh[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) dn[k] = (dnh[k] - nh[k] * pr) / nrm;
    } else {                                            // the clamp holds the norm: no gradient through it
#pragma unroll
        for (int k = 0; k < 3; ++k) dn[k] = dnh[k] / nrm;
    }
    const float g0 = grad_rgb[i * 3], g1 = grad_rgb[i * 3 + 1], g2 = grad_rgb[i * 3 + 2];
    const float dt = ga.w > 0.0f ? (g0 * rgb_s[i * 3] + g1 * rgb_s[i * 3 + 1]) + g2 * rgb_s[i * 3 + 2] : 0.0f;
    *(float4 *)(dg8 + i * 8) = make_float4(dn[0], dn[1], dn[2], dt);
    *(float4 *)(dg8 + i * 8 + 4) = make_float4(g0, g1, g2, 0.0f);
    (void)gb;
    vmax = fmaxf(fmaxf(fmaxf(fabsf(dn[0]), fabsf(dn[1])), fmaxf(fabsf(dn[2]), fabsf(dt))), fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fabsf(g2)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if ((threadIdx.x & 63) == 0 && vmax > 0.0f) atomicMax(amax_h, __float_as_uint(vmax));
}

// the heads' gradient scale: 2^floor(log2(target / max |dg8|)) (their operands go straight into the transpose product: no 0.25 headroom for a chain)
__global__ void heads_scale_kernel(unsigned *__restrict__ amax_h, const float target, float *__restrict__ gscale_h)
{
    const float gmax = fmaxf(__uint_as_float(*amax_h), 1e-30f);
    float e = floorf(log2f(target / gmax));
    e = fminf(fmaxf(e, -60.0f), 60.0f);
    *gscale_h = exp2f(e);
    *amax_h = 0u;
}

// dh^T [144 x 32e] = Bas^T [144 x 32] dF^T [32 x 32e]  (gscale as above; dF is stored unscaled).  KS = 3 (REFTensoRF): a third k-step adds
// Heads^T [144 x 8] dG^T, the gradient through normal / tint / rgb_d (k 32..35 = dg8[0..3] in lane half 0, k 36..39 = dg8[4..7] in lane half 1).
template <int KS>
__global__ __launch_bounds__(256, 2) void basis_backward_kernel(const float *__restrict__ dfeats, const float *__restrict__ dg8, long long m_cap, const unsigned *__restrict__ m_dev,
                                                                const float *__restrict__ gscale_p, const unsigned char *__restrict__ image, float *__restrict__ dh, unsigned *sat)
{
    const long long m = m_dev ? ((long long)*m_dev < m_cap ? (long long)*m_dev : m_cap) : m_cap;
    // the heads' k-step has its OWN power-of-two scale (and accumulators): d normal carries 1 / |n| and d rgb_d is the colour gradient itself, neither
    // shares the range of the gradients that went through the network
    const float gscale_h = KS == 3 ? *((const float *)(image + TI_SCAL) + 1) : 1.0f, inv_scale_h = 1.0f / gscale_h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = lane & 31, h = lane >> 5;
    const long long n_tiles = (m + 31) / 32;
    const float gscale = *gscale_p, inv_scale = 1.0f / gscale;
    float amax = 0.0f;
    // the 10 (15) A fragments of this lane, hi and lo (same for every tile)
    uint4 ah[5][KS], al[5][KS];
#pragma unroll
    for (int rb = 0; rb < 5; ++rb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 *ap = (const uint4 *)(image + TI_BAST) + (size_t)((((rb * KS + s) * 2 + h) * 32 + e) * 2);
            ah[rb][s] = ap[0];
            al[rb][s] = ap[1];
        }
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
        const long long ent = tile * 32 + e;
        const bool live = ent < m;
        const long long le = live ? ent : m - 1;
        float x[8 * KS];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 f4 = *(const float4 *)(dfeats + le * 32 + 8 * q + 4 * h);
            x[4 * q] = f4.x * gscale; x[4 * q + 1] = f4.y * gscale; x[4 * q + 2] = f4.z * gscale; x[4 * q + 3] = f4.w * gscale;
        }
        if (KS == 3) {
            const float4 g4 = *(const float4 *)(dg8 + le * 8 + 4 * h);
            x[16] = g4.x * gscale_h; x[17] = g4.y * gscale_h; x[18] = g4.z * gscale_h; x[19] = g4.w * gscale_h;
            x[20] = 0.f; x[21] = 0.f; x[22] = 0.f; x[23] = 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8 * KS; ++r) amax = fmaxf(amax, fabsf(x[r]));
        Frag b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = split8(x + 8 * s);
        f32x16 acc[5];
#pragma unroll
        for (int rb = 0; rb < 5; ++rb) acc[rb] = f32x16{0};
        if (KS == 3) {
            // the heads' k-step first, at its own scale; then the accumulators are moved to the network's scale (a power of two: exact) and the two
            // feature k-steps accumulate on top — one set of accumulators (a second set spills: 160 + 120 fragment registers)
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) acc[rb] = MFMAH(al[rb][KS - 1], b[KS - 1].hi, acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) acc[rb] = MFMAH(ah[rb][KS - 1], b[KS - 1].lo, acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) acc[rb] = MFMAH(ah[rb][KS - 1], b[KS - 1].hi, acc[rb]);
            const float ratio = gscale * inv_scale_h;
#pragma unroll
            for (int rb = 0; rb < 5; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] = acc[rb][r] * ratio;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) acc[rb] = MFMAH(al[rb][s], b[s].hi, acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) acc[rb] = MFMAH(ah[rb][s], b[s].lo, acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) acc[rb] = MFMAH(ah[rb][s], b[s].hi, acc[rb]);
        }
        if (live) {
#pragma unroll
            for (int rb = 0; rb < 5; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 32 * rb + 8 * q + 4 * h;
                    if (row < TVR_KAPP)
                        *(float4 *)(dh + ent * TVR_KAPP + row) =
                            make_float4(acc[rb][4 * q] * inv_scale, acc[rb][4 * q + 1] * inv_scale, acc[rb][4 * q + 2] * inv_scale, acc[rb][4 * q + 3] * inv_scale);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (sat && amax >= 65504.0f) atomicOr(sat, 1u);
}

size_t mlp_train_image_bytes() { return TI_BYTES_ALL; }

hipError_t launch_pack_train_image(const float *W1, const float *W2, const float *W3, const float *Bas, const float *const heads[4], void *image, hipStream_t stream,
                                   int fea_pe, int view_pe, const int *app_n_comp, int featureC)
{
    if (featureC < 1 || featureC > TVR_FEATC || (heads && (featureC != TVR_FEATC || fea_pe != 2 || view_pe != 2))) return hipErrorInvalidValue;
    AppCols ac;
    ac.k_app = 0;
    for (int i = 0; i < 3; ++i) { ac.n[i] = app_n_comp ? app_n_comp[i] : TVR_CA; ac.off[i] = ac.k_app; ac.k_app += ac.n[i]; }
    if (heads && ac.k_app != TVR_KAPP) return hipErrorInvalidValue;           // (REFTensoRF's heads are packed at 144 columns)
    HeadPtrs hp;
    for (int i = 0; i < 4; ++i) hp.W[i] = heads ? heads[i] : nullptr;
    const int ref = heads ? 1 : 0, gen = (fea_pe > 2 || view_pe > 2) ? 1 : 0;
    if (gen && (ref || fea_pe > TVR_GEN_PE || view_pe > TVR_GEN_PE || fea_pe < 0 || view_pe < 0)) return hipErrorInvalidValue;
    const int n = 128 * 128 + 160 * 128 + 3 * 128 + 160 * (ref ? 48 : 32);
    hipLaunchKernelGGL(pack_train_image_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, W1, W2, W3, Bas, hp, ref, gen, ac, featureC, fea_pe, view_pe, (unsigned char *)image);
    if (gen) hipLaunchKernelGGL(pack_train_w1gen_kernel, dim3((TVR_GEN_T * 8 * 64 + 255) / 256), dim3(256), 0, stream, W1, featureC, fea_pe, view_pe, (unsigned char *)image);
    return hipGetLastError();
}

hipError_t launch_mlp_train_backward(const float *grad_rgb, const float *rgb, const float *feats, const float *h1, const float *h2, long long m, const float *gscale,
                                     float *d_out, float *dh2, float *dh1, float *dfeats, float *dh, unsigned *sat_flag, void *image, const MlpRefBwd *ref,
                                     hipStream_t stream, const unsigned *m_dev, int gen)
{
    if (gen && ref) return hipErrorInvalidValue;
    hipError_t rc = hipFuncSetAttribute(ref ? (const void *)mlp_train_backward_kernel<true> : (gen ? (const void *)mlp_train_backward_kernel<false, true> : (const void *)mlp_train_backward_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, TI_LDS_BYTES);
    if (rc != hipSuccess) return rc;
    MlpBwdArgs a;
    a.grad_rgb = grad_rgb; a.rgb = rgb; a.feats = feats; a.h1 = h1; a.h2 = h2; a.m = m; a.m_dev = m_dev; a.gscale = gscale;
    a.d_out = d_out; a.dh2 = dh2; a.dh1 = dh1; a.dfeats = dfeats; a.image = (const unsigned char *)image; a.sat = sat_flag;
    a.g8 = ref ? ref->g8 : nullptr;
    const long long groups = (m + 32 * MT_WAVES - 1) / (32 * MT_WAVES);
    unsigned grid = groups < 256 ? (unsigned)(groups > 0 ? groups : 1) : 256u;
    if (ref) hipLaunchKernelGGL(mlp_train_backward_kernel<true>, dim3(grid), dim3(MT_THREADS), TI_LDS_BYTES, stream, a);
    else if (gen) hipLaunchKernelGGL((mlp_train_backward_kernel<false, true>), dim3(grid), dim3(MT_THREADS), TI_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(mlp_train_backward_kernel<false>, dim3(grid), dim3(MT_THREADS), TI_LDS_BYTES, stream, a);
    rc = hipGetLastError();
    if (rc != hipSuccess) return rc;
    const long long g2 = (m + 127) / 128;
    const unsigned grid2 = (unsigned)(g2 < 1024 ? (g2 > 0 ? g2 : 1) : 1024);
    if (ref) {
        hipLaunchKernelGGL(ref_heads_backward_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, grad_rgb, rgb, ref->g8, ref->viewdirs, ref->rays, ref->q_ray,
                           dfeats, ref->grad_in0, m, m_dev, ref->dg8, (unsigned *)((unsigned char *)image + TI_SCAL));
        hipLaunchKernelGGL(heads_scale_kernel, dim3(1), dim3(1), 0, stream, (unsigned *)((unsigned char *)image + TI_SCAL), 4096.0f, (float *)((unsigned char *)image + TI_SCAL) + 1);
        rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(basis_backward_kernel<3>, dim3(grid2), dim3(256), 0, stream, dfeats, ref->dg8, m, m_dev, gscale, (const unsigned char *)image, dh, sat_flag);
    } else {
        hipLaunchKernelGGL(basis_backward_kernel<2>, dim3(grid2), dim3(256), 0, stream, dfeats, (const float *)nullptr, m, m_dev, gscale, (const unsigned char *)image, dh, sat_flag);
    }
    return hipGetLastError();
}
