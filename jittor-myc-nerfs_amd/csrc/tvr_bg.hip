// tvr_bg.hip — NerfPlusPlus's background network as ONE kernel (second half): `Embedder` (nerfplusplus.py:7-56) of the
// inverted-sphere points and of the view directions + `MLPNet.forward` (:66-140) — what `NerfPlusPlus.execute` evaluates on 512
// background samples per ray (:280-302), i.e. 328 M samples per 800x800 frame.
//
// Network (W = 128): base layers Linear+ReLU x D (layer 0 from the point embedding, the layer after the skip from
// cat(embedding, base)), sigma = |Linear(128,1)|, base_remap Linear(128,256), rgb = sigmoid(Linear(64,3)(relu(Linear(256+view,64)))).
// base_remap has no activation behind it, so the host folds it into the first rgb layer (W_eff = W_rgb0[:, :256] W_remap, fp64 on
// the host): 128 -> 64 instead of 128 -> 256 -> 64, a third of the network's multiply-adds gone, same function.
//
// Kernel: a workgroup of 8 waves, one per SIMD pair, each wave carrying 32 samples through all layers with activations
// register-resident (tvr_mfma.h: fp16 hi/lo split, 3 products, accumulator layout == next layer's B layout).  The weight image
// (up to 300 KB as hi/lo fragments) does not fit the 160 KB LDS, so the layers are cut into two stages; the workgroup switches the
// LDS image between them (two barriers per stage), every wave keeping its 128 activations in registers across the switch.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include "tvr_kernels.h"
#include "tvr_mfma.h"

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return tvr_set_error(TVR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define BG_MAX_BLOCKS 77                      // fragment blocks (2 KiB each) per stage image
#define BG_BIAS_FLOATS 640                    // 4 x 128 base biases, 64 rgb-hidden, sigma, 3 rgb (padded)
#define BG_LDS_BYTES (BG_MAX_BLOCKS * 2048 + BG_BIAS_FLOATS * 4)
#define BG_WAVES 8
#ifndef TVR_BG_APF
#define TVR_BG_APF 0
#endif
// Round 6: the STREAMED form (default).  The fragment image lies in the order the kernel consumes it and moves through two 64 KB halves of LDS in chunks of at most
// BG_HALF_BLOCKS blocks: while every wave computes on the chunk in one half, the next chunk arrives in the other by LDS-DMA (global_load_lds_dwordx4: no registers, no
// ds_write), ONE workgroup barrier per chunk.  Round 5's form (two 154 KB stages, each loaded synchronously between two barriers: 4 barriers and 300 KB of exposed
// L2 -> LDS traffic per 256 samples, measured at 11 % of the kernel: profiles/r06_bg_kernel.txt) stays selectable with -DTVR_BG_STREAM=0 for A/B builds.
#ifndef TVR_BG_STREAM
#define TVR_BG_STREAM 1
#endif
#define BG_HALF_BLOCKS 32
#define BG_HALF_BYTES (BG_HALF_BLOCKS * 2048)
#define BG_MAX_CHUNKS 8
#define BG_LDS_STREAM_BYTES (2 * BG_HALF_BYTES + BG_BIAS_FLOATS * 4)

// training forward: what the backward needs (all optional, NULL = inference) — relu outputs of the base layers [M,128] each, of the rgb hidden layer
// [M,64], and the sigma head's value before `abs` [M]
struct BgTrain {
    float *A[4];
    float *Hrgb;
    float *sig_pre;
    unsigned long long *MA[4];   // relu masks as bits, [n][2] 64-bit words: word h of sample s holds, at bit 16 mb + 4 q + i, whether unit 32 mb + 8 q + 4 h + i
    unsigned long long *MH;      // is positive — the accumulator order of the lane that owns them, which is also tvr_linear_dx's epilogue order
};

struct BgProgram {
    int D, n_pe_steps, input_ch, split;       // base layers [0, split) run from stage A, the rest and the heads from stage B
    int blocksA, blocksB;                     // blocks per stage (<= BG_MAX_BLOCKS); stage B's image follows stage A's in global memory
    int base_block0[4];                       // first block of each base layer inside its stage
    int base_prev[4], base_pe[4];             // does the layer read the previous activations / the point embedding
    int sig_block0, rgbh_block0, rgbo_block0; // inside stage B
    int samples_per_ray;
    // the streamed form: the image in consumption order — per base layer [the 8 k-steps of the previous activations x 4 row blocks][the n_pe_steps of the point
    // embedding x 4], then the heads [8 k-steps x {sigma, rgb-hidden 0, rgb-hidden 1}][view k-step x 2][4 k-steps of the rgb output].  Units (what a chunk is made of):
    // half a layer's previous-activation part (16 blocks), a layer's embedding part (<= 12), the heads (30); bit u of bnd_mask: unit u opens a chunk.
    int prev_blk0[4], pe_blk0[4], heads_blk0;
    int n_chunks, chunk_blk0[BG_MAX_CHUNKS], chunk_nblk[BG_MAX_CHUNKS];
    unsigned bnd_mask;
};

// ------------------------------------------------------------------------------------------------ packing
struct PackBlock {
    const float *W;      // [n_out, ld] row-major
    int ld, n_out, row0; // rows row0 .. row0+31 of W
    int kind, t, koff;   // how lane (h, j) maps to a column of W (see bg_pack_kernel)
    int n_valid;         // columns >= koff + n_valid are padding (kind PE / VIEW)
};
enum { K_PREV = 0, K_PE = 1, K_VIEW = 2 };

__global__ void __launch_bounds__(64) bg_pack_kernel(const PackBlock *__restrict__ blocks, uint4 *__restrict__ image)
{
    const PackBlock b = blocks[blockIdx.x];
    const int l = threadIdx.x, h = l >> 5, row = b.row0 + (l & 31);
    float v[8];
    for (int j = 0; j < 8; ++j) {
        int k, ok = 1;
        if (b.kind == K_PREV) k = (b.t / 2) * 32 + (2 * (b.t & 1) + j / 4) * 8 + 4 * h + (j & 3);     // accumulator order of a 32-neuron block pair
        else if (b.kind == K_PE) { k = 16 * b.t + 8 * h + j; ok = k < b.n_valid; }
        else { k = 8 * h + j; ok = k < b.n_valid; }
        v[j] = (ok && row < b.n_out) ? b.W[(size_t)row * b.ld + b.koff + k] : 0.f;
    }
    const Frag f = split8(v);
    image[((size_t)blockIdx.x * 2 + 0) * 64 + l] = f.hi;
    image[((size_t)blockIdx.x * 2 + 1) * 64 + l] = f.lo;
}

// ------------------------------------------------------------------------------------------------ kernel
// Embedder feature f of a 4-vector x (nerfplusplus.py:36-56: [x, sin(x f0), cos(x f0), sin(x f1), ...], f_q = 2^q)
__device__ __forceinline__ float pe_feature(int f, const float x[4])
{
    if (f < 4) return x[f];
    const int g = f - 4, q = g >> 3, r = g & 7;
    const float a = x[r & 3] * (float)(1 << q);
    return r < 4 ? __sinf(a) : __cosf(a);
}
template <int AR>
__device__ __forceinline__ Frag pe_frag(int t, int hh, const float x[4])
{
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float lo = pe_feature(16 * t + j, x), hi = pe_feature(16 * t + 8 + j, x);
        v[j] = hh ? hi : lo;
    }
    return frag8<AR>(v);
}
// B fragment of k-step t (0..7) of a 128-wide activation held in four accumulators, relu applied
template <int AR>
__device__ __forceinline__ Frag relu_frag4(const f32x16 a[4], int t)
{
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = relu_f(a[t >> 1][8 * (t & 1) + j]);
    return frag8<AR>(v);
}
__device__ __forceinline__ f32x16 bias_acc(const float *__restrict__ bias32, int hh)
{
    // accumulator register i <-> neuron (i/4)*8 + 4h + i%4 of the block
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(bias32 + q * 8 + 4 * hh);
        a[4 * q] = b.x; a[4 * q + 1] = b.y; a[4 * q + 2] = b.z; a[4 * q + 3] = b.w;
    }
    return a;
}
// one k-step into NB accumulators (independent blocks: their MFMAs interleave): A fragments from LDS, then AR products per block
// (AR = tvr_mlpnet_desc.arith, include/tvr.h TVR_ARITH_*: 3 = Wlo*xhi + Whi*xlo + Whi*xhi, fp32-class; 2 = Wlo*xhi + Whi*xhi, the layer inputs rounded to fp16;
// 1 = Whi*xhi.  What a mode does not multiply is neither read nor derived — frag8<AR>, tvr_mfma.h)
template <int NB>
struct AFrags {
    uint4 h[NB], l[NB];
};
template <int NB, int AR>
__device__ __forceinline__ AFrags<NB> load_a(const uint4 *__restrict__ w4, const int (&blk)[NB])
{
    AFrags<NB> a;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        a.h[m] = w4[(blk[m] * 2 + 0) * 64];
        if constexpr (AR >= 2) a.l[m] = w4[(blk[m] * 2 + 1) * 64];
    }
    return a;
}
template <int NB, int AR>
__device__ __forceinline__ void mma(const AFrags<NB> &a, const Frag &b, f32x16 (&acc)[NB])
{
    if constexpr (AR >= 2) {
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = MFMAH(a.l[m], b.hi, acc[m]);
    }
    if constexpr (AR >= 3) {
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = MFMAH(a.h[m], b.lo, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = MFMAH(a.h[m], b.hi, acc[m]);
}
template <int NB, int AR>
__device__ __forceinline__ void kstep(const uint4 *__restrict__ w4, const int (&blk)[NB], const Frag &b, f32x16 (&acc)[NB])
{
    mma<NB, AR>(load_a<NB, AR>(w4, blk), b, acc);
}

__device__ __forceinline__ void load_stage(uint4 *__restrict__ lds4, const uint4 *__restrict__ image, int n_blocks)
{
    const int n = n_blocks * 128;
    for (int e = threadIdx.x; e < n; e += BG_WAVES * 64) lds4[e] = image[e];
}

// base layers [l0, l1) on one 32-sample tile: act in (unused when l0 == 0) -> act out
// relu(act) of this lane's sample: accumulator register 4q + i of block mb <-> neuron 32 mb + 8 q + 4 h + i
template <int NB>
__device__ __forceinline__ void store_mask(unsigned long long *__restrict__ out, long long s, int hh, const f32x16 (&act)[NB])
{
    unsigned long long m = 0ull;
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= (unsigned long long)(act[mb][r] > 0.0f) << (16 * mb + r);
    out[2 * s + hh] = m;
}

__device__ __forceinline__ void store_relu128(float *__restrict__ out, long long s, int hh, const f32x16 (&act)[4])
{
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4 *)(out + s * 128 + 32 * mb + 8 * q + 4 * hh) =
                make_float4(relu_f(act[mb][4 * q]), relu_f(act[mb][4 * q + 1]), relu_f(act[mb][4 * q + 2]), relu_f(act[mb][4 * q + 3]));
}

template <int AR>
__device__ __forceinline__ void base_layers(const BgProgram &P, int l0, int l1, const uint4 *__restrict__ w4, const float *__restrict__ lbias, int hh,
                                            const float x[4], f32x16 (&act)[4], const BgTrain &T, long long s_store)
{
    for (int l = l0; l < l1; ++l) {
        f32x16 out[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) out[mb] = bias_acc(lbias + l * 128 + mb * 32, hh);
        const int prev = P.base_prev[l], pe = P.base_pe[l], spm = (prev ? 8 : 0) + (pe ? P.n_pe_steps : 0), b0 = P.base_block0[l];
        if (prev) {
#if TVR_BG_APF
            // the A fragments of k-step t+1 are fetched from LDS before the MFMAs of k-step t are issued (LDS latency off the MFMA path)
            const int blk0[4] = {b0, b0 + spm, b0 + 2 * spm, b0 + 3 * spm};
            AFrags<4> a = load_a<4, AR>(w4, blk0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tn = t < 7 ? t + 1 : 7;
                const int blk[4] = {b0 + tn, b0 + spm + tn, b0 + 2 * spm + tn, b0 + 3 * spm + tn};
                const AFrags<4> an = load_a<4, AR>(w4, blk);
                const Frag b = relu_frag4<AR>(act, t);
                mma<4, AR>(a, b, out);
                a = an;
            }
#else
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const Frag b = relu_frag4<AR>(act, t);
                const int blk[4] = {b0 + t, b0 + spm + t, b0 + 2 * spm + t, b0 + 3 * spm + t};
                kstep<4, AR>(w4, blk, b, out);
            }
#endif
        }
        if (pe) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < P.n_pe_steps) {
                    const Frag b = pe_frag<AR>(t, hh, x);
                    const int o = b0 + (prev ? 8 : 0) + t;
                    const int blk[4] = {o, o + spm, o + 2 * spm, o + 3 * spm};
                    kstep<4, AR>(w4, blk, b, out);
                }
            }
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) act[mb] = out[mb];
        if (T.A[0] && s_store >= 0) {
            store_relu128(T.A[l], s_store, hh, act);
            if (T.MA[l]) store_mask<4>(T.MA[l], s_store, hh, act);
        }
    }
}

// heads on one tile: sigma and the 64-wide rgb hidden layer share the fragments of `base`; returns (rgb, sigma) of sample col in the
// lanes with hh == 0
template <int AR>
__device__ __forceinline__ float4 heads(const BgProgram &P, const uint4 *__restrict__ w4, const float *__restrict__ lbias, int hh, const float d[3],
                                        const f32x16 (&act)[4], const BgTrain &T, long long s_store)
{
    f32x16 hd[3] = {{0}, bias_acc(lbias + 512, hh), bias_acc(lbias + 512 + 32, hh)};      // sigma, rgb hidden block 0 / 1
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const Frag b = relu_frag4<AR>(act, t);
        const int blk[3] = {P.sig_block0 + t, P.rgbh_block0 + t, P.rgbh_block0 + 9 + t};
        kstep<3, AR>(w4, blk, b, hd);
    }
    f32x16 rh[2] = {hd[1], hd[2]};
    {
        // view-direction embedding: [d, sin d, cos d, sin 2d, cos 2d] (15 values, one k-step)
        float v[8];
        const float v0[8] = {d[0], d[1], d[2], __sinf(d[0]), __sinf(d[1]), __sinf(d[2]), __cosf(d[0]), __cosf(d[1])};
        const float v1[8] = {__cosf(d[2]), __sinf(2.f * d[0]), __sinf(2.f * d[1]), __sinf(2.f * d[2]), __cosf(2.f * d[0]), __cosf(2.f * d[1]), __cosf(2.f * d[2]), 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = hh ? v1[j] : v0[j];
        const int blk[2] = {P.rgbh_block0 + 8, P.rgbh_block0 + 9 + 8};
        kstep<2, AR>(w4, blk, frag8<AR>(v), rh);
    }
    if (T.Hrgb && s_store >= 0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4 *)(T.Hrgb + s_store * 64 + 32 * mb + 8 * q + 4 * hh) =
                    make_float4(relu_f(rh[mb][4 * q]), relu_f(rh[mb][4 * q + 1]), relu_f(rh[mb][4 * q + 2]), relu_f(rh[mb][4 * q + 3]));
        if (hh == 0) T.sig_pre[s_store] = hd[0][0] + lbias[576];
        if (T.MH) store_mask<2>(T.MH, s_store, hh, rh);
    }
    f32x16 eo[1] = {{0}};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_f(t < 2 ? rh[0][8 * (t & 1) + j] : rh[1][8 * (t & 1) + j]);
        const int blk[1] = {P.rgbo_block0 + t};
        kstep<1, AR>(w4, blk, frag8<AR>(v), eo);
    }
    return make_float4(1.0f / (1.0f + __expf(-(eo[0][0] + lbias[580]))), 1.0f / (1.0f + __expf(-(eo[0][1] + lbias[581]))),
                       1.0f / (1.0f + __expf(-(eo[0][2] + lbias[582]))), fabsf(hd[0][0] + lbias[576]));
}

#ifndef TVR_BG_TICKETS
#define TVR_BG_TICKETS 1          // 1: dynamic hand-out of the super-tiles (0: static stride over the workgroups)
#endif
#ifndef TVR_BG_DIAG
#define TVR_BG_DIAG 0             // timing stand-ins (WRONG results, never shipped): 1 = the LDS stage images are loaded for the first super-tile only (what the reloads
#endif                            // cost), 2 = ... and no workgroup barriers around them either (what the lockstep costs)
#ifndef TVR_BG_NT
#define TVR_BG_NT 1               // 32-sample tiles a wave carries through each LDS stage (their stage-A activations wait in registers)
#endif

template <int AR>
__global__ void __launch_bounds__(BG_WAVES * 64, 1) bg_mlp_kernel(BgProgram P, const uint4 *__restrict__ image, const float *__restrict__ bias,
                                                                 const float *__restrict__ pts, const float *__restrict__ viewdirs, long long M,
                                                                 float *__restrict__ rgb, float *__restrict__ sigma, const BgTrain T, unsigned *__restrict__ tk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long s_next;
    uint4 *lds4 = reinterpret_cast<uint4 *>(smem);
    float *lbias = reinterpret_cast<float *>(smem + BG_MAX_BLOCKS * 2048);
    for (int e = threadIdx.x; e < BG_BIAS_FLOATS; e += BG_WAVES * 64) lbias[e] = bias[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
    constexpr int NT = TVR_BG_NT, PER_SUPER = 32 * BG_WAVES * NT;
    const long long n_super = (M + PER_SUPER - 1) / PER_SUPER;
    const uint4 *imgA = image, *imgB = image + (size_t)P.blocksA * 128;
    const int la = min(P.split, P.D);

    // Round 5: super-tiles are handed out dynamically (the first one static, later ones by an atomicAdd on `tk`, a word the launcher zeroes) — the XCDs do not run at one
    // speed under an MFMA-heavy kernel and equal static shares leave the fast ones idle at the end (profiles/r05_shade_tail.txt).  The atomic for the tile after next is
    // issued a whole super-tile before its value is read.  Which workgroup evaluates a sample does not matter to the sample.
    unsigned tk_pending = 0;
    if (tk && threadIdx.x == 0) tk_pending = atomicAdd(tk, 1u);
    for (long long super = blockIdx.x; super < n_super;) {
        int hh = h, lane_off = lane;
        asm volatile("" : "+v"(hh), "+v"(lane_off));                // opaque per tile: keeps per-lane selects / LDS reads from being hoisted
        const uint4 *w4 = lds4 + lane_off;
        f32x16 act[NT][4];
        // ---------------- stage A: base layers [0, split)
#if TVR_BG_DIAG
        const bool diag_first = super == (long long)blockIdx.x;
        if (TVR_BG_DIAG < 2 || diag_first) __syncthreads();
        if (diag_first) load_stage(lds4, imgA, P.blocksA);
        if (TVR_BG_DIAG < 2 || diag_first) __syncthreads();
#else
        __syncthreads();                                             // everyone is done with the previous tile's stage B
        load_stage(lds4, imgA, P.blocksA);
        __syncthreads();
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long long s = ((super * NT + nt) * BG_WAVES + wave) * 32 + col, sr = min(s, M - 1);
            const float4 p = *reinterpret_cast<const float4 *>(pts + 4 * sr);
            const float x[4] = {p.x, p.y, p.z, p.w};
            base_layers<AR>(P, 0, la, w4, lbias, hh, x, act[nt], T, s < M ? s : -1);
        }
        // ---------------- stage B: the remaining base layers and the heads
#if TVR_BG_DIAG
        if (TVR_BG_DIAG < 2) { __syncthreads(); __syncthreads(); }
#else
        __syncthreads();
        load_stage(lds4, imgB, P.blocksB);
        __syncthreads();
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long long s = ((super * NT + nt) * BG_WAVES + wave) * 32 + col, sr = min(s, M - 1);
            const float4 p = *reinterpret_cast<const float4 *>(pts + 4 * sr);
            const float x[4] = {p.x, p.y, p.z, p.w};
            const float *v = viewdirs + 3 * (sr / P.samples_per_ray);
            const float d[3] = {v[0], v[1], v[2]};
            base_layers<AR>(P, la, P.D, w4, lbias, hh, x, act[nt], T, s < M ? s : -1);
            const float4 r = heads<AR>(P, w4, lbias, hh, d, act[nt], T, s < M ? s : -1);
            if (h == 0 && s < M) {
                sigma[s] = r.w;
                rgb[3 * s] = r.x;
                rgb[3 * s + 1] = r.y;
                rgb[3 * s + 2] = r.z;
            }
        }
        if (tk) {
            if (threadIdx.x == 0) {
                const long long nx = (long long)gridDim.x + (long long)tk_pending;
                s_next = nx;
                if (nx < n_super) tk_pending = atomicAdd(tk, 1u);
            }
            __syncthreads();
            super = s_next;                        // (the next iteration's first barrier orders this read before thread 0's next write)
        } else {
            super += gridDim.x;
        }
    }
}
__global__ void bg_zero_ticket_kernel(unsigned *tk) { *tk = 0u; }

// ------------------------------------------------------------------------------------------------ the streamed kernel (round 6)
typedef __attribute__((address_space(3))) void bg_lds_void;
typedef __attribute__((address_space(1))) const void bg_glb_void;
struct AF { uint4 h, l; };
#define BG_SB __builtin_amdgcn_sched_barrier(0)
#define BG_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define BG_SG_VALU(n) __builtin_amdgcn_sched_group_barrier(0x402, (n), 0)      // VALU | TRANS
#define BG_SG_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#ifndef TVR_BG_SCHED
#define TVR_BG_SCHED 1            // sched_group_barrier windows in the units (0: hipcc's own order, A/B)
#endif
#ifndef TVR_BG_STAGE
#define TVR_BG_STAGE 0            // 1: the chunk transfer as global_load_dwordx4 -> registers -> ds_write_b128 one k-step later, instead of LDS-DMA (whose issue holds the wave ~100 cycles per KB)
#endif
#ifndef TVR_BG_PRIOFLIP
#define TVR_BG_PRIOFLIP 0         // experiment: the two waves of a SIMD (w and w + 4) take turns at priority 1, k-step by k-step (keeps them abreast between the chunk barriers)
#endif
#ifndef TVR_BG_TIMING
#define TVR_BG_TIMING 0           // diagnostic build: per-phase s_memtime sums of every wave into the work buffer's words 8.. (scripts/bg_phase_timing.py)
#endif
#if TVR_BG_TIMING
#define BG_STAMP(x) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define BG_STAMP(x)
#endif
#ifndef TVR_BG_PD
#define TVR_BG_PD 1               // fragment blocks read ahead of the one being multiplied (ring of PD + 2 {hi, lo} pairs: 8 registers each) in the embedding units
#endif
#ifndef TVR_BG_PD_PREV
#define TVR_BG_PD_PREV 1          // ... in the previous-activation units and the heads (2: measured below)
#endif
// the three products of one fragment block on ONE accumulator, back to back (a dependent chain of 32x32x16 MFMAs issues back to back: profiles/r04_mfma_issue_probe.txt);
// per accumulator the order of the additions is tvr_mfma.h's / round 5's: Wlo*xhi, Whi*xlo, Whi*xhi
template <int AR>
__device__ __forceinline__ void mfma3(const AF &A, const Frag &b, f32x16 &acc)
{
    if constexpr (AR >= 2) acc = MFMAH(A.l, b.hi, acc);
    if constexpr (AR >= 3) acc = MFMAH(A.h, b.lo, acc);
    acc = MFMAH(A.h, b.hi, acc);
}
// One unit: NG k-steps x NB row blocks whose fragment blocks lie consecutively at `lb` (this lane's LDS address of the unit's first block), for the NTW 32-sample tiles a
// wave carries.  hipcc's own schedule puts a k-step's A-fragment reads right in front of their MFMAs — `ds_read, s_waitcnt, MFMA` ~75 times per tile (scripts/isa_trace.py on
// round 5's kernel) — so the reads run TVR_BG_PD blocks ahead here through a ring of {hi, lo} pairs, a fragment lives for 3 NTW MFMAs (the tiles of a wave SHARE every weight
// fragment: with two tiles the LDS reads and the DMA per sample halve), and the next k-step's B fragments (relu + fp16 split, or the embedding's sin / cos) are derived under
// the MFMAs of this one (VPG vector instructions per MFMA gap), as in tvr_shade.hip's matrix phase.
template <int NB, int NG, int AR, int VPG, int NTW, int PD = TVR_BG_PD, typename GetB, typename Hook>
__device__ __forceinline__ void run_unit(const unsigned char *lb, f32x16 (&acc)[NTW][NB], GetB getB, Hook hook)
{
    constexpr int NQ = NB * NG, RN = PD + 2;
    AF ring[RN];
    auto ld = [&](int q) {
        ring[q % RN].h = *(const uint4 *)(lb + q * 2048);
        if constexpr (AR >= 2) ring[q % RN].l = *(const uint4 *)(lb + q * 2048 + 1024);
    };
#pragma unroll
    for (int q0 = 0; q0 < PD; ++q0)
        if (q0 < NQ) ld(q0);
    Frag b[NTW], nb[NTW];
#pragma unroll
    for (int w = 0; w < NTW; ++w) nb[w] = b[w] = getB(w, 0);
    BG_SB;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        hook();                                                         // (the next chunk's transfer, a piece per k-step: see the kernel)
        BG_SB;
        if (g + 1 < NG) {
#pragma unroll
            for (int w = 0; w < NTW; ++w) nb[w] = getB(w, g + 1);
        }
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int q = g * NB + m;
            if (q + PD < NQ) ld(q + PD);
            const AF &A = ring[q % RN];
            // per accumulator the order of the additions is tvr_mfma.h's (Wlo*xhi, Whi*xlo, Whi*xhi); the tiles of the wave alternate, so no MFMA waits for the one before it
            if constexpr (AR >= 2) {
#pragma unroll
                for (int w = 0; w < NTW; ++w) acc[w][m] = MFMAH(A.l, b[w].hi, acc[w][m]);
            }
            if constexpr (AR >= 3) {
#pragma unroll
                for (int w = 0; w < NTW; ++w) acc[w][m] = MFMAH(A.h, b[w].lo, acc[w][m]);
            }
#pragma unroll
            for (int w = 0; w < NTW; ++w) acc[w][m] = MFMAH(A.h, b[w].hi, acc[w][m]);
        }
#if TVR_BG_SCHED
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            if (g * NB + m + PD < NQ) BG_SG_DSR(AR >= 2 ? 2 : 1);
#pragma unroll
            for (int i = 0; i < AR * NTW; ++i) {
                BG_SG_MFMA(1);
                if (g + 1 < NG) BG_SG_VALU(VPG);
            }
        }
#endif
        if (g + 1 < NG) {
#pragma unroll
            for (int w = 0; w < NTW; ++w) b[w] = nb[w];
        }
        BG_SB;
    }
}

// NTW = 1, 8 waves (two per SIMD, 256 registers each): the form of this round's first half.  NTW = 2, 4 waves (ONE per SIMD, the whole register file): a wave carries two
// tiles through every fragment — the MFMA / VALU overlap that two lock-stepped waves of a SIMD do not give each other is built into one instruction stream.
template <int AR, int NTW, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) bg_mlp_stream_kernel(BgProgram P, const uint4 *__restrict__ image, const float *__restrict__ bias,
                                                                     const float *__restrict__ pts, const float *__restrict__ viewdirs, long long M,
                                                                     float *__restrict__ rgb, float *__restrict__ sigma, const BgTrain T, unsigned *__restrict__ tk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long s_next;
    float *lbias = reinterpret_cast<float *>(smem + 2 * BG_HALF_BYTES);
    for (int e = threadIdx.x; e < BG_BIAS_FLOATS; e += WAVES * 64) lbias[e] = bias[e];              // (visible behind the first chunk's barrier)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, col = lane & 31;
    constexpr int PER_SUPER = 32 * WAVES * NTW;
    const long long n_super = (M + PER_SUPER - 1) / PER_SUPER;

    // ---- the chunk stream.  cn: the chunk (index within a tile) the next boundary opens; hsel: the half it arrives in.  The DMA of a chunk runs one whole chunk of
    // compute ahead of its first read, issued by all waves (1 KB pieces, piece p by wave p % WAVES): at a boundary every wave waits for its own pieces (vmcnt(0)), the
    // barrier then says that ALL pieces have landed and that nobody reads the other half any more — which the chunk after next may now overwrite.
    int cn = 0, hsel = 0;
    unsigned cur = 0;                                                   // LDS byte offset of the next unit's first block
    // The pieces of a chunk's DMA are NOT issued in one burst behind the barrier: an LDS-DMA instruction holds its wave for ~100 cycles (measured: 4 060 cycles per
    // tile and wave for 37.5 pieces, with both waves of every SIMD bursting at once and nobody feeding the matrix pipe — profiles/r06_bg_kernel.txt).  A wave issues
    // its pieces k-step by k-step under the chunk it computes on (dma_step, the hook of run_unit) and whatever is left at the next boundary.
    int dma_p = 0, dma_n = 0;                                           // this wave's next piece / the chunk's piece count
    const unsigned char *dma_src = nullptr;
    unsigned dma_dst = 0;
#if TVR_BG_PRIOFLIP
    unsigned flip = (unsigned)(wave >> 2);
#endif
#if TVR_BG_STAGE
    uint4 st_val[NTW];
    unsigned st_dst[NTW];
    int st_have[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) { st_have[i] = 0; st_dst[i] = 0; st_val[i] = make_uint4(0, 0, 0, 0); }
#endif
    auto dma_step = [&]() {
#if TVR_BG_PRIOFLIP
        flip ^= 1u;
        if (flip) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
#if TVR_BG_STAGE
        // register-staged: the piece requested one k-step ago goes to LDS now, the next one is requested
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            if (st_have[i]) {
                *reinterpret_cast<uint4 *>(smem + st_dst[i] + lane * 16) = st_val[i];
                st_have[i] = 0;
            }
            if (dma_p < dma_n) {
                st_val[i] = *reinterpret_cast<const uint4 *>(dma_src + (size_t)dma_p * 1024);
                st_dst[i] = dma_dst + (unsigned)dma_p * 1024u;
                st_have[i] = 1;
                dma_p += WAVES;
            }
        }
#else
#pragma unroll
        for (int i = 0; i < NTW; ++i)                                   // (two tiles per wave: half the waves, twice the pieces each, twice the MFMAs per k-step to put them under)
            if (dma_p < dma_n) {
                __builtin_amdgcn_global_load_lds((bg_glb_void *)(dma_src + (size_t)dma_p * 1024), (bg_lds_void *)(smem + dma_dst + dma_p * 1024), 16, 0, 0);
                dma_p += WAVES;
            }
#endif
    };
    auto issue_dma = [&](int c, int half) {                             // arm the DMA of chunk c into `half`
        dma_n = P.chunk_nblk[c] * 2;
        dma_src = reinterpret_cast<const unsigned char *>(image) + (size_t)P.chunk_blk0[c] * 2048 + lane * 16;
        dma_dst = (unsigned)half * BG_HALF_BYTES;
        dma_p = wave;
    };
#if TVR_BG_TIMING
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta = 0, tb = 0, tc = 0, td = 0;
#endif
    auto enter_chunk = [&]() {
        BG_STAMP(ta);
        while (dma_p < dma_n) dma_step();                               // (a chunk with fewer k-steps than pieces per wave: the rest now)
#if TVR_BG_STAGE
        dma_step();                                                     // (the last staged piece into LDS; the barrier below waits for the ds_write)
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BG_STAMP(tb);
        __syncthreads();
        BG_STAMP(tc);
        cur = (unsigned)hsel * BG_HALF_BYTES;
        cn = cn + 1 == P.n_chunks ? 0 : cn + 1;
        hsel ^= 1;
        issue_dma(cn, hsel);                                            // the NEXT chunk (the next tile's first, behind a tile's last) into the half just vacated
        BG_STAMP(td);
#if TVR_BG_TIMING
        tsum[1] += tb - ta; tsum[2] += tc - tb; tsum[3] += td - tc;
#endif
    };
    issue_dma(0, 0);

    // Tickets (round 5) without a barrier of their own (round 6): thread 0 publishes the NEXT tile's number in front of this tile's first chunk barrier and every wave
    // reads it behind that barrier — a whole tile before it is needed, so the next tile's points are fetched under this tile's layers.
    unsigned tk_pending = 0;
    if (tk && threadIdx.x == 0) tk_pending = atomicAdd(tk, 1u);
    auto sample_of = [&](long long super, int w) { return ((super * WAVES + wave) * NTW + w) * 32 + col; };
    float4 p_next[NTW];
#pragma unroll
    for (int w = 0; w < NTW; ++w) p_next[w] = *reinterpret_cast<const float4 *>(pts + 4 * min(sample_of(blockIdx.x, w), M - 1));
    for (long long super = blockIdx.x; super < n_super;) {
#if TVR_BG_TIMING
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#endif
        BG_STAMP(t0);
        int hh = h;
        unsigned lane16 = (unsigned)(size_t)(smem + lane * 16);
        asm volatile("" : "+v"(hh), "+v"(lane16));                      // opaque per tile: keeps per-lane selects / LDS addresses from being hoisted out of the loop
        if (threadIdx.x == 0) {
            const long long nx = tk ? (long long)gridDim.x + (long long)tk_pending : super + gridDim.x;
            s_next = nx;
            if (tk && nx < n_super) tk_pending = atomicAdd(tk, 1u);
        }
        enter_chunk();                                                  // the tile's first chunk (unit 0 always opens one)
        const long long super_next = s_next;                            // (thread 0 writes it again in front of the NEXT tile's first barrier: four barriers from here)
        int u = 0;                                                      // unit counter of this tile (uniform)
        auto unit_ptr = [&]() -> const unsigned char * {                // boundary check + this lane's LDS address of the unit's first block (ONE register: the reads use immediates)
            if (u > 0 && ((P.bnd_mask >> u) & 1u)) enter_chunk();
            ++u;
            unsigned a = lane16 + cur;
            asm volatile("" : "+v"(a));
            return (const unsigned char *)(const void __attribute__((address_space(3))) *)(size_t)a;
        };
        long long s[NTW], sr[NTW];
        float x[NTW][4];
#pragma unroll
        for (int w = 0; w < NTW; ++w) {
            s[w] = sample_of(super, w);
            sr[w] = min(s[w], M - 1);
            x[w][0] = p_next[w].x; x[w][1] = p_next[w].y; x[w][2] = p_next[w].z; x[w][3] = p_next[w].w;
            p_next[w] = *reinterpret_cast<const float4 *>(pts + 4 * min(sample_of(super_next, w), M - 1));     // (a tile number beyond the last reads the last sample's point: never used)
        }
#if TVR_BG_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (timing build: everything outstanding is here before the clock is read)
#endif
        BG_STAMP(t1);
        f32x16 act[NTW][4];
        // ---------------- base layers
        for (int l = 0; l < P.D; ++l) {
            f32x16 out[NTW][4];
#pragma unroll
            for (int w = 0; w < NTW; ++w)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) out[w][mb] = bias_acc(lbias + l * 128 + mb * 32, hh);
            if (P.base_prev[l]) {
                run_unit<4, 4, AR, 2, NTW, TVR_BG_PD_PREV>(unit_ptr(), out, [&](int w, int g) { return relu_frag4<AR>(act[w], g); }, dma_step);
                cur += 16 * 2048;
                run_unit<4, 4, AR, 2, NTW, TVR_BG_PD_PREV>(unit_ptr(), out, [&](int w, int g) { return relu_frag4<AR>(act[w], 4 + g); }, dma_step);
                cur += 16 * 2048;
            }
            if (P.base_pe[l]) {
                const unsigned char *lb = unit_ptr();
                auto pe = [&](int w, int g) { return pe_frag<AR>(g, hh, x[w]); };
                if (P.n_pe_steps == 3) run_unit<4, 3, AR, 4, NTW>(lb, out, pe, dma_step);
                else if (P.n_pe_steps == 2) run_unit<4, 2, AR, 4, NTW>(lb, out, pe, dma_step);
                else run_unit<4, 1, AR, 4, NTW>(lb, out, pe, dma_step);
                cur += (unsigned)P.n_pe_steps * 4 * 2048;
            }
#pragma unroll
            for (int w = 0; w < NTW; ++w) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) act[w][mb] = out[w][mb];
                if (T.A[0] && s[w] < M) {
                    store_relu128(T.A[l], s[w], hh, act[w]);
                    if (T.MA[l]) store_mask<4>(T.MA[l], s[w], hh, act[w]);
                }
            }
        }
        BG_STAMP(t2);
        // ---------------- heads: sigma and the 64-wide rgb hidden layer share the fragments of `base`; then the view k-step, then the rgb output (tvr_bg.hip heads())
        {
            const unsigned char *lb = unit_ptr();
            float d[NTW][3];
#pragma unroll
            for (int w = 0; w < NTW; ++w) {                             // (issued here: the latency passes under the eight k-steps below)
                const float *vp = viewdirs + 3 * (sr[w] / P.samples_per_ray);
                d[w][0] = vp[0]; d[w][1] = vp[1]; d[w][2] = vp[2];
            }
            f32x16 hd[NTW][3];
#pragma unroll
            for (int w = 0; w < NTW; ++w) {
                hd[w][0] = f32x16{0};
                hd[w][1] = bias_acc(lbias + 512, hh);
                hd[w][2] = bias_acc(lbias + 512 + 32, hh);
            }
            run_unit<3, 8, AR, 3, NTW, TVR_BG_PD_PREV>(lb, hd, [&](int w, int g) { return relu_frag4<AR>(act[w], g); }, dma_step);
            f32x16 rh[NTW][2];
            Frag bv[NTW];
#pragma unroll
            for (int w = 0; w < NTW; ++w) {
                rh[w][0] = hd[w][1]; rh[w][1] = hd[w][2];
                // view-direction embedding: [d, sin d, cos d, sin 2d, cos 2d] (15 values, one k-step)
                const float *dd = d[w];
                float v[8];
                const float v0[8] = {dd[0], dd[1], dd[2], __sinf(dd[0]), __sinf(dd[1]), __sinf(dd[2]), __cosf(dd[0]), __cosf(dd[1])};
                const float v1[8] = {__cosf(dd[2]), __sinf(2.f * dd[0]), __sinf(2.f * dd[1]), __sinf(2.f * dd[2]), __cosf(2.f * dd[0]), __cosf(2.f * dd[1]), __cosf(2.f * dd[2]), 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hh ? v1[j] : v0[j];
                bv[w] = frag8<AR>(v);
            }
            run_unit<2, 1, AR, 1, NTW>(lb + 24 * 2048, rh, [&](int w, int) { return bv[w]; }, dma_step);
#pragma unroll
            for (int w = 0; w < NTW; ++w) {
                if (T.Hrgb && s[w] < M) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(float4 *)(T.Hrgb + s[w] * 64 + 32 * mb + 8 * q + 4 * hh) =
                                make_float4(relu_f(rh[w][mb][4 * q]), relu_f(rh[w][mb][4 * q + 1]), relu_f(rh[w][mb][4 * q + 2]), relu_f(rh[w][mb][4 * q + 3]));
                    if (hh == 0) T.sig_pre[s[w]] = hd[w][0][0] + lbias[576];
                    if (T.MH) store_mask<2>(T.MH, s[w], hh, rh[w]);
                }
            }
            f32x16 eo[NTW][1];
#pragma unroll
            for (int w = 0; w < NTW; ++w) eo[w][0] = f32x16{0};
            run_unit<1, 4, AR, 8 / NTW, NTW>(lb + 26 * 2048, eo, [&](int w, int t) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu_f(t < 2 ? rh[w][0][8 * (t & 1) + j] : rh[w][1][8 * (t & 1) + j]);
                return frag8<AR>(v);
            }, dma_step);
            cur += 30 * 2048;
#pragma unroll
            for (int w = 0; w < NTW; ++w)
                if (h == 0 && s[w] < M) {
                    sigma[s[w]] = fabsf(hd[w][0][0] + lbias[576]);
                    rgb[3 * s[w]] = 1.0f / (1.0f + __expf(-(eo[w][0][0] + lbias[580])));
                    rgb[3 * s[w] + 1] = 1.0f / (1.0f + __expf(-(eo[w][0][1] + lbias[581])));
                    rgb[3 * s[w] + 2] = 1.0f / (1.0f + __expf(-(eo[w][0][2] + lbias[582])));
                }
        }
        BG_STAMP(t3);
        super = super_next;
        BG_STAMP(t4);
#if TVR_BG_TIMING
        tsum[0] += t4 - t0; tsum[4] += t1 - t0; tsum[5] += t2 - t1; tsum[6] += t3 - t2; tsum[7] += t4 - t3;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the one chunk fetched ahead and never used: no DMA may outlive the wave
#if TVR_BG_TIMING
    // per wave: [0] tile total, [1] boundary: wait for the DMA (+ everything else outstanding), [2] boundary: barrier, [3] boundary: arming the next DMA, [4] tile start (first
    // boundary included) -> everything outstanding landed, [5] base layers (boundaries included), [6] heads (its boundary included) + stores issued, [7] -
    if (tk && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(tk) + 8 + i, tsum[i]);
#endif
}

// ------------------------------------------------------------------------------------------------ background geometry and compositing
// NerfPlusPlus.execute around the network (nerfplusplus.py:280-308): perturbed depths (`perturb_samples` :196-205), inverted-sphere
// points (`depth2pts_outside` :207-237), the flip along the sample axis (:296-297) — one thread per (ray, flipped sample) — and the
// front-to-back weights  alpha = 1 - exp(-sigma * dist), T = cumprod(1 - alpha + 1e-6), rgb = sum alpha T rgb  (:298-308) — one wave per ray.
__global__ void __launch_bounds__(256) bg_points_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d, long long n_rays,
                                                        const float *__restrict__ z_lin, int N, const float *__restrict__ t_rand, float radii,
                                                        float4 *__restrict__ pts, float *__restrict__ z_out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rays * N) return;
    const long long ray = idx / N;
    const int kf = (int)(idx - ray * N), k = N - 1 - kf;             // kf: position after the flip, k: position before it
    const float z = z_lin[k], zl = k > 0 ? 0.5f * (z + z_lin[k - 1]) : z, zu = k < N - 1 ? 0.5f * (z_lin[k + 1] + z) : z;
    const float depth = zl + (zu - zl) * t_rand[ray * N + k];
    const float o0 = rays_o[3 * ray], o1 = rays_o[3 * ray + 1], o2 = rays_o[3 * ray + 2], d0 = rays_d[3 * ray], d1v = rays_d[3 * ray + 1], d2v = rays_d[3 * ray + 2];
    const float dd = d0 * d0 + d1v * d1v + d2v * d2v;
    const float d1 = -(d0 * o0 + d1v * o1 + d2v * o2) / dd;
    const float m0 = o0 + d1 * d0, m1 = o1 + d1 * d1v, m2 = o2 + d1 * d2v;
    const float mn = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
    const float dcos = 1.0f / sqrtf(dd);
    const float d2 = sqrtf(radii * radii - mn * mn) * dcos;
    const float s0 = o0 + (d1 + d2) * d0, s1 = o1 + (d1 + d2) * d1v, s2 = o2 + (d1 + d2) * d2v;
    float a0 = o1 * s2 - o2 * s1, a1 = o2 * s0 - o0 * s2, a2 = o0 * s1 - o1 * s0;                // cross(ray_o, p_sphere)
    const float an = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    a0 /= an; a1 /= an; a2 /= an;
    const float ang = asinf(mn / radii) - asinf(mn * depth / (radii * radii));
    const float ca = cosf(ang), sa = sinf(ang);
    const float c0 = a1 * s2 - a2 * s1, c1 = a2 * s0 - a0 * s2, c2 = a0 * s1 - a1 * s0;          // cross(rot_axis, p_sphere)
    const float dot = a0 * s0 + a1 * s1 + a2 * s2;
    pts[idx] = make_float4(s0 * ca + c0 * sa + a0 * dot * (1.f - ca), s1 * ca + c1 * sa + a1 * dot * (1.f - ca), s2 * ca + c2 * sa + a2 * dot * (1.f - ca), depth);
    z_out[idx] = depth;
}

__global__ void __launch_bounds__(256) bg_composite_kernel(const float *__restrict__ rgb, const float *__restrict__ sigma, const float *__restrict__ z, long long n_rays, int N,
                                                           float *__restrict__ out)
{
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    const float *sg = sigma + ray * N, *zz = z + ray * N, *cc = rgb + 3 * ray * N;
    float carry = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k0 = 0; k0 < N; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < N;
        const float dist = in ? (k < N - 1 ? zz[k] - zz[k + 1] : 1e10f) : 0.f;
        const float alpha = in ? 1.f - expf(-sg[k] * dist) : 0.f;
        float f = in ? 1.f - alpha + 1e-6f : 1.f;                  // inclusive prefix product over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float u = __shfl_up(f, d, 64);
            if (lane >= d) f *= u;
        }
        const float excl = __shfl_up(f, 1, 64);
        const float T = carry * (lane == 0 ? 1.f : excl);
        if (in) {
            const float w = alpha * T;
            c0 += w * cc[3 * k];
            c1 += w * cc[3 * k + 1];
            c2 += w * cc[3 * k + 2];
        }
        carry *= __shfl(f, 63, 64);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        c0 += __shfl_xor(c0, m, 64);
        c1 += __shfl_xor(c1, m, 64);
        c2 += __shfl_xor(c2, m, 64);
    }
    if (lane == 0) {
        out[3 * ray] = c0;
        out[3 * ray + 1] = c1;
        out[3 * ray + 2] = c2;
    }
}

// The embedded inputs as matrices, for the weight gradients of the layers that read them (dW = dY^T E): E_pos [M, input_ch] = Embedder(pts) and
// E_view [M, 16] = Embedder(viewdirs) of the sample's ray (15 values + a zero), with the network kernel's own functions and column order.
__global__ void __launch_bounds__(256) bg_embed_kernel(const float *__restrict__ pts, const float *__restrict__ viewdirs, long long M, int input_ch, int samples_per_ray,
                                                       float *__restrict__ Epos, float *__restrict__ Eview)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= M) return;
    const float4 p = *reinterpret_cast<const float4 *>(pts + 4 * s);
    const float x[4] = {p.x, p.y, p.z, p.w};
    for (int f = 0; f < input_ch; f += 4)
        *(float4 *)(Epos + s * input_ch + f) = make_float4(pe_feature(f, x), pe_feature(f + 1, x), pe_feature(f + 2, x), pe_feature(f + 3, x));
    const float *v = viewdirs + 3 * (s / samples_per_ray);
    const float d[3] = {v[0], v[1], v[2]};
    float *e = Eview + s * 16;
    *(float4 *)(e) = make_float4(d[0], d[1], d[2], __sinf(d[0]));
    *(float4 *)(e + 4) = make_float4(__sinf(d[1]), __sinf(d[2]), __cosf(d[0]), __cosf(d[1]));
    *(float4 *)(e + 8) = make_float4(__cosf(d[2]), __sinf(2.f * d[0]), __sinf(2.f * d[1]), __sinf(2.f * d[2]));
    *(float4 *)(e + 12) = make_float4(__cosf(2.f * d[0]), __cosf(2.f * d[1]), __cosf(2.f * d[2]), 0.f);
}

// ------------------------------------------------------------------------------------------------ C-ABI
static inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

struct BgLayout {
    BgProgram P;
    int total_blocks;
};

static int plan(const tvr_mlpnet_desc *d, BgLayout &L)
{
    if (!d) return tvr_set_error(TVR_ERR_INVALID, "mlpnet desc is NULL");
    if (d->W != 128) return tvr_set_error(TVR_ERR_UNSUPPORTED, "MLPNet width %d (the kernel is built for W = 128, nerfplusplus.py:159)", d->W);
    if (d->D < 2 || d->D > 4) return tvr_set_error(TVR_ERR_UNSUPPORTED, "MLPNet depth %d outside [2,4]", d->D);
    if (d->pos_freqs < 1 || d->pos_freqs > 4) return tvr_set_error(TVR_ERR_UNSUPPORTED, "position embedding with %d frequencies outside [1,4]", d->pos_freqs);
    if (d->view_freqs != 2) return tvr_set_error(TVR_ERR_UNSUPPORTED, "view embedding with %d frequencies (built for 2)", d->view_freqs);
    if (d->skip < 0 || d->skip >= d->D) return tvr_set_error(TVR_ERR_INVALID, "skip layer %d outside [0,%d)", d->skip, d->D);
    if (d->samples_per_ray < 1) return tvr_set_error(TVR_ERR_INVALID, "samples_per_ray < 1");
    BgProgram &P = L.P;
    P.D = d->D;
    P.input_ch = 4 + 8 * d->pos_freqs;
    P.n_pe_steps = (P.input_ch + 15) / 16;
    P.samples_per_ray = d->samples_per_ray;
    int blocks[4];
    for (int l = 0; l < P.D; ++l) {
        P.base_prev[l] = l > 0;
        P.base_pe[l] = l == 0 || (l - 1 == d->skip && l - 1 != P.D - 1);     // MLPNet.__init__: `if i in skips and i != D-1: dim += input_ch`
        blocks[l] = 4 * ((P.base_prev[l] ? 8 : 0) + (P.base_pe[l] ? P.n_pe_steps : 0));
    }
    const int heads = 8 + 18 + 4;
    // the split that balances the two stages and fits both
    P.split = -1;
    for (int s = 1; s <= P.D; ++s) {
        int a = 0, b = heads;
        for (int l = 0; l < P.D; ++l) (l < s ? a : b) += blocks[l];
        if (a <= BG_MAX_BLOCKS && b <= BG_MAX_BLOCKS) P.split = s;
    }
    if (P.split < 0) return tvr_set_error(TVR_ERR_UNSUPPORTED, "MLPNet does not fit two LDS stages");
    int a = 0, b = 0;
    for (int l = 0; l < P.D; ++l) {
        if (l < P.split) { P.base_block0[l] = a; a += blocks[l]; }
        else { P.base_block0[l] = b; b += blocks[l]; }
    }
    P.sig_block0 = b; b += 8;
    P.rgbh_block0 = b; b += 18;
    P.rgbo_block0 = b; b += 4;
    P.blocksA = a;
    P.blocksB = b;
    L.total_blocks = a + b;
    // the streamed form (TVR_BG_STREAM): the same blocks in consumption order, cut into chunks of whole units that fit one LDS half
    {
        int pos = 0, n_units = 0, unit_blocks[16];
        for (int l = 0; l < P.D; ++l) {
            P.prev_blk0[l] = P.pe_blk0[l] = -1;
            if (P.base_prev[l]) { P.prev_blk0[l] = pos; pos += 32; unit_blocks[n_units++] = 16; unit_blocks[n_units++] = 16; }
            if (P.base_pe[l]) { P.pe_blk0[l] = pos; pos += 4 * P.n_pe_steps; unit_blocks[n_units++] = 4 * P.n_pe_steps; }
        }
        P.heads_blk0 = pos; pos += heads; unit_blocks[n_units++] = heads;
        if (pos != L.total_blocks) return tvr_set_error(TVR_ERR_INVALID, "tvr_bg plan: the two layouts disagree (%d vs %d blocks)", pos, L.total_blocks);
        P.bnd_mask = 0u;
        P.n_chunks = 0;
        int fill = BG_HALF_BLOCKS + 1, at = 0;                      // (the first unit always opens a chunk)
        for (int u = 0; u < n_units; ++u) {
            if (fill + unit_blocks[u] > BG_HALF_BLOCKS) {
                if (P.n_chunks == BG_MAX_CHUNKS) return tvr_set_error(TVR_ERR_UNSUPPORTED, "MLPNet needs more than %d LDS chunks", BG_MAX_CHUNKS);
                P.bnd_mask |= 1u << u;
                P.chunk_blk0[P.n_chunks] = at;
                P.chunk_nblk[P.n_chunks] = 0;
                ++P.n_chunks;
                fill = 0;
            }
            fill += unit_blocks[u];
            P.chunk_nblk[P.n_chunks - 1] += unit_blocks[u];
            at += unit_blocks[u];
        }
        for (int c = P.n_chunks; c < BG_MAX_CHUNKS; ++c) P.chunk_blk0[c] = P.chunk_nblk[c] = 0;
    }
    return TVR_OK;
}

// The packed network: fragment image, biases, and the block table the pack kernel reads (tvr_mlpnet_repack walks it on the device every training step).
// Nothing in it is written by a forward: the forward kernel's ticket word lives in the caller's `work` buffer (round 6; in round 5 it sat in 256 extra bytes
// behind the table, written through the `const` image — and a working-tree form that placed it ON the table ended in a device abort).
static size_t bg_packed_bytes(const BgLayout &L)
{
    return ((size_t)L.total_blocks * 2048 + BG_BIAS_FLOATS * 4 + (size_t)L.total_blocks * sizeof(PackBlock) + 255) & ~(size_t)255;
}

template <int AR>
static int mlpnet_launch(const BgLayout &L, const void *packed, const void *pts, const void *viewdirs, int64_t n_samples, void *rgb, void *sigma, const BgTrain &T, void *work,
                         void *stream)
{
#if TVR_BG_STREAM == 2
    auto *kern = bg_mlp_stream_kernel<AR, 2, 4>;                     // one wave per SIMD, two tiles per wave
    const int lds_bytes = BG_LDS_STREAM_BYTES;
    constexpr int KW = 4, KNT = 2;
#elif TVR_BG_STREAM
    auto *kern = bg_mlp_stream_kernel<AR, 1, BG_WAVES>;
    const int lds_bytes = BG_LDS_STREAM_BYTES;
    constexpr int KW = BG_WAVES, KNT = 1;
#else
    auto *kern = bg_mlp_kernel<AR>;
    const int lds_bytes = BG_LDS_BYTES;
    constexpr int KW = BG_WAVES, KNT = TVR_BG_NT;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    const long long per_super = 32 * KW * KNT, n_super = (n_samples + per_super - 1) / per_super;
    const unsigned blocks = (unsigned)(n_super < 256 ? n_super : 256);                     // one workgroup per CU (the LDS image)
    const char *base = static_cast<const char *>(packed);
    // the ticket word: word 0 of the caller's `work` buffer (tvr_mlpnet_work_bytes), zeroed here and advanced by the kernel.  One launch per work buffer at a time;
    // launches on different streams take different work buffers and may share the (read-only) packed network.
    unsigned *tk = TVR_BG_TICKETS ? static_cast<unsigned *>(work) : nullptr;
    if (tk) hipLaunchKernelGGL(bg_zero_ticket_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), tk);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(KW * 64), lds_bytes, static_cast<hipStream_t>(stream), L.P, reinterpret_cast<const uint4 *>(base),
                       reinterpret_cast<const float *>(base + (size_t)L.total_blocks * 2048), static_cast<const float *>(pts), static_cast<const float *>(viewdirs),
                       (long long)n_samples, static_cast<float *>(rgb), static_cast<float *>(sigma), T, tk);
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

extern "C" {

size_t tvr_mlpnet_packed_bytes(const tvr_mlpnet_desc *desc)
{
    BgLayout L;
    if (plan(desc, L) != TVR_OK) return 0;
    return bg_packed_bytes(L);
}

size_t tvr_mlpnet_work_bytes(void) { return 256; }

int tvr_mlpnet_describe(const tvr_mlpnet_desc *desc, tvr_mlpnet_layout *out)
{
    BgLayout L;
    if (!out) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_describe: out is NULL");
    if (int rc = plan(desc, L)) return rc;
    out->fragments = 0;
    out->biases = (size_t)L.total_blocks * 2048;
    out->block_table = out->biases + BG_BIAS_FLOATS * 4;
    out->block_table_bytes = (size_t)L.total_blocks * sizeof(PackBlock);
    out->total = bg_packed_bytes(L);
    return TVR_OK;
}

int tvr_mlpnet_pack(const tvr_mlpnet_desc *desc, const tvr_mlpnet_params *p, void *packed, size_t packed_bytes, void *stream)
{
    BgLayout L;
    if (int rc = plan(desc, L)) return rc;
    if (!p || !p->sigma_W || !p->sigma_b || !p->rgbh_W_base || !p->rgbh_W_view || !p->rgbh_b || !p->rgbo_W || !p->rgbo_b)
        return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_pack: NULL weights");
    for (int l = 0; l < desc->D; ++l)
        if (!p->base_W[l] || !p->base_b[l]) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_pack: NULL base layer %d", l);
    if (!packed || misaligned(packed) || packed_bytes < tvr_mlpnet_packed_bytes(desc)) return tvr_set_error(TVR_ERR_SCRATCH, "tvr_mlpnet_pack: packed buffer too small or misaligned");
    const BgProgram &P = L.P;
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<PackBlock> tab((size_t)L.total_blocks);
#if TVR_BG_STREAM
    // consumption order (BgProgram): block (k-step t, row block mb) of a part at part_blk0 + t * (row blocks of the part) + mb
    auto put = [&](int blk, const void *W, int ld, int n_out, int row0, int kind, int t, int koff, int n_valid) {
        tab[(size_t)blk] = PackBlock{static_cast<const float *>(W), ld, n_out, row0, kind, t, koff, n_valid};
    };
    for (int l = 0; l < P.D; ++l) {
        const int prev = P.base_prev[l], pe = P.base_pe[l], ld = (prev ? 128 : 0) + (pe ? P.input_ch : 0);
        for (int mb = 0; mb < 4; ++mb) {
            // cat(input_pts, base): the embedding occupies the first input_ch columns, the previous activations follow (MLPNet.forward)
            if (prev) for (int t = 0; t < 8; ++t) put(P.prev_blk0[l] + 4 * t + mb, p->base_W[l], ld, 128, mb * 32, K_PREV, t, pe ? P.input_ch : 0, 128);
            if (pe) for (int t = 0; t < P.n_pe_steps; ++t) put(P.pe_blk0[l] + 4 * t + mb, p->base_W[l], ld, 128, mb * 32, K_PE, t, 0, P.input_ch);
        }
    }
    for (int t = 0; t < 8; ++t) {
        put(P.heads_blk0 + 3 * t, p->sigma_W, 128, 1, 0, K_PREV, t, 0, 128);
        for (int mb = 0; mb < 2; ++mb) put(P.heads_blk0 + 3 * t + 1 + mb, p->rgbh_W_base, 128, 64, mb * 32, K_PREV, t, 0, 128);
    }
    for (int mb = 0; mb < 2; ++mb) put(P.heads_blk0 + 24 + mb, p->rgbh_W_view, 15, 64, mb * 32, K_VIEW, 0, 0, 15);
    for (int t = 0; t < 4; ++t) put(P.heads_blk0 + 26 + t, p->rgbo_W, 64, 3, 0, K_PREV, t, 0, 64);
#else
    auto put = [&](int stage, int blk, const void *W, int ld, int n_out, int row0, int kind, int t, int koff, int n_valid) {
        tab[(size_t)(stage ? P.blocksA : 0) + blk] = PackBlock{static_cast<const float *>(W), ld, n_out, row0, kind, t, koff, n_valid};
    };
    for (int l = 0; l < P.D; ++l) {
        const int stage = l >= P.split, prev = P.base_prev[l], pe = P.base_pe[l], spm = (prev ? 8 : 0) + (pe ? P.n_pe_steps : 0);
        const int ld = (prev ? 128 : 0) + (pe ? P.input_ch : 0);
        for (int mb = 0; mb < 4; ++mb) {
            const int b0 = P.base_block0[l] + mb * spm;
            // cat(input_pts, base): the embedding occupies the first input_ch columns, the previous activations follow (MLPNet.forward)
            if (prev) for (int t = 0; t < 8; ++t) put(stage, b0 + t, p->base_W[l], ld, 128, mb * 32, K_PREV, t, pe ? P.input_ch : 0, 128);
            if (pe) for (int t = 0; t < P.n_pe_steps; ++t) put(stage, b0 + (prev ? 8 : 0) + t, p->base_W[l], ld, 128, mb * 32, K_PE, t, 0, P.input_ch);
        }
    }
    for (int t = 0; t < 8; ++t) put(1, P.sig_block0 + t, p->sigma_W, 128, 1, 0, K_PREV, t, 0, 128);
    for (int mb = 0; mb < 2; ++mb) {
        for (int t = 0; t < 8; ++t) put(1, P.rgbh_block0 + mb * 9 + t, p->rgbh_W_base, 128, 64, mb * 32, K_PREV, t, 0, 128);
        put(1, P.rgbh_block0 + mb * 9 + 8, p->rgbh_W_view, 15, 64, mb * 32, K_VIEW, 0, 0, 15);
    }
    for (int t = 0; t < 4; ++t) put(1, P.rgbo_block0 + t, p->rgbo_W, 64, 3, 0, K_PREV, t, 0, 64);
#endif
    char *base = static_cast<char *>(packed);
    float *bias = reinterpret_cast<float *>(base + (size_t)L.total_blocks * 2048);
    PackBlock *dtab = reinterpret_cast<PackBlock *>(base + (size_t)L.total_blocks * 2048 + BG_BIAS_FLOATS * 4);
    HIP_TRY(hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(PackBlock), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));                               // `tab` is a host temporary (packing is a rare, explicit call)
    HIP_TRY(hipMemsetAsync(bias, 0, BG_BIAS_FLOATS * 4, st));
    for (int l = 0; l < P.D; ++l) HIP_TRY(hipMemcpyAsync(bias + l * 128, p->base_b[l], 128 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(bias + 512, p->rgbh_b, 64 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(bias + 576, p->sigma_b, 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(bias + 580, p->rgbo_b, 12, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(bg_pack_kernel, dim3((unsigned)L.total_blocks), dim3(64), 0, st, dtab, reinterpret_cast<uint4 *>(base));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

// arith: TVR_ARITH_* of the INFERENCE call (tvr_mlpnet_desc.arith); the training forward passes TVR_ARITH_F32 whatever the descriptor says
static int mlpnet_forward_impl(const tvr_mlpnet_desc *desc, const void *packed, size_t packed_bytes, const void *pts, const void *viewdirs, int64_t n_samples, void *rgb,
                               void *sigma, const BgTrain &T, void *work, size_t work_bytes, void *stream, int arith)
{
    BgLayout L;
    if (int rc = plan(desc, L)) return rc;
    if (n_samples < 0) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_forward: n_samples < 0");
    if (n_samples == 0) return TVR_OK;
    if (!packed || !pts || !viewdirs || !rgb || !sigma || misaligned(packed) || misaligned(pts)) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_forward: NULL or misaligned argument");
    if (packed_bytes < bg_packed_bytes(L)) return tvr_set_error(TVR_ERR_SCRATCH, "tvr_mlpnet_forward: packed network smaller than tvr_mlpnet_packed_bytes()");
    if (!work || misaligned(work) || work_bytes < tvr_mlpnet_work_bytes())
        return tvr_set_error(work ? TVR_ERR_SCRATCH : TVR_ERR_INVALID, "tvr_mlpnet_forward: work buffer NULL, misaligned or smaller than tvr_mlpnet_work_bytes()");
    if (arith != TVR_ARITH_F32 && arith != TVR_ARITH_F16ACT && arith != TVR_ARITH_F16) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_forward: desc.arith %d is none of TVR_ARITH_*", arith);
    if (arith == TVR_ARITH_F16ACT) return mlpnet_launch<2>(L, packed, pts, viewdirs, n_samples, rgb, sigma, T, work, stream);
    if (arith == TVR_ARITH_F16) return mlpnet_launch<1>(L, packed, pts, viewdirs, n_samples, rgb, sigma, T, work, stream);
    return mlpnet_launch<3>(L, packed, pts, viewdirs, n_samples, rgb, sigma, T, work, stream);
}

int tvr_mlpnet_forward(const tvr_mlpnet_desc *desc, const void *packed, size_t packed_bytes, const void *pts, const void *viewdirs, int64_t n_samples, void *rgb,
                       void *sigma, void *work, size_t work_bytes, void *stream)
{
    BgTrain T = {};
    return mlpnet_forward_impl(desc, packed, packed_bytes, pts, viewdirs, n_samples, rgb, sigma, T, work, work_bytes, stream, desc ? desc->arith : 0);
}

int tvr_mlpnet_train_forward(const tvr_mlpnet_desc *desc, const void *packed, size_t packed_bytes, const void *pts, const void *viewdirs, int64_t n_samples, void *rgb,
                             void *sigma, const tvr_mlpnet_saved *saved, void *work, size_t work_bytes, void *stream)
{
    if (!desc || !saved) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_train_forward: desc / saved NULL");
    const size_t rows = n_samples < 0 ? 0 : (size_t)n_samples;
    BgTrain T = {};
    for (int l = 0; l < desc->D && l < 4; ++l) {
        if (!saved->act[l] || misaligned(saved->act[l]) || saved->act_bytes < rows * 128 * 4)
            return tvr_set_error(saved->act[l] ? TVR_ERR_SCRATCH : TVR_ERR_INVALID, "tvr_mlpnet_train_forward: act[%d] NULL, misaligned or smaller than n_samples x 128 floats", l);
        T.A[l] = static_cast<float *>(saved->act[l]);
    }
    if (!saved->rgb_hidden || !saved->sigma_pre || !saved->embed_pos || !saved->embed_view || misaligned(saved->rgb_hidden) || misaligned(saved->embed_pos) ||
        misaligned(saved->embed_view))
        return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_train_forward: rgb_hidden / sigma_pre / embed_pos / embed_view NULL or misaligned");
    const int input_ch = 4 + 8 * desc->pos_freqs;
    if (saved->rgb_hidden_bytes < rows * 64 * 4 || saved->sigma_pre_bytes < rows * 4 || saved->embed_pos_bytes < rows * input_ch * 4 || saved->embed_view_bytes < rows * 16 * 4)
        return tvr_set_error(TVR_ERR_SCRATCH, "tvr_mlpnet_train_forward: a saved buffer is smaller than its n_samples rows");
    T.Hrgb = static_cast<float *>(saved->rgb_hidden);
    T.sig_pre = static_cast<float *>(saved->sigma_pre);
    if (saved->mask_bytes) {                                         // optional: the relu masks as bits
        if (saved->mask_bytes < rows * 16) return tvr_set_error(TVR_ERR_SCRATCH, "tvr_mlpnet_train_forward: a mask buffer is smaller than n_samples x 16 bytes");
        for (int l = 0; l < desc->D && l < 4; ++l) {
            if (!saved->act_mask[l] || misaligned(saved->act_mask[l])) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_train_forward: act_mask[%d] NULL or misaligned", l);
            T.MA[l] = static_cast<unsigned long long *>(saved->act_mask[l]);
        }
        if (!saved->rgb_hidden_mask || misaligned(saved->rgb_hidden_mask)) return tvr_set_error(TVR_ERR_INVALID, "tvr_mlpnet_train_forward: rgb_hidden_mask NULL or misaligned");
        T.MH = static_cast<unsigned long long *>(saved->rgb_hidden_mask);
    }
    if (int rc = mlpnet_forward_impl(desc, packed, packed_bytes, pts, viewdirs, n_samples, rgb, sigma, T, work, work_bytes, stream, TVR_ARITH_F32)) return rc;
    if (n_samples > 0) {
        hipLaunchKernelGGL(bg_embed_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float *>(pts),
                           static_cast<const float *>(viewdirs), (long long)n_samples, input_ch, (int)desc->samples_per_ray, static_cast<float *>(saved->embed_pos),
                           static_cast<float *>(saved->embed_view));
        HIP_TRY(hipGetLastError());
    }
    return TVR_OK;
}

// The image again from the weights the table in `packed` already points at (same tensors, new values: every training step) — no host synchronisation.
int tvr_mlpnet_repack(const tvr_mlpnet_desc *desc, const tvr_mlpnet_params *p, void *packed, size_t packed_bytes, void *stream)
{
    BgLayout L;
    if (int rc = plan(desc, L)) return rc;
    if (!p || !packed || misaligned(packed) || packed_bytes < tvr_mlpnet_packed_bytes(desc)) return tvr_set_error(TVR_ERR_SCRATCH, "tvr_mlpnet_repack: packed buffer NULL, too small or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(packed);
    float *bias = reinterpret_cast<float *>(base + (size_t)L.total_blocks * 2048);
    PackBlock *dtab = reinterpret_cast<PackBlock *>(base + (size_t)L.total_blocks * 2048 + BG_BIAS_FLOATS * 4);
    for (int l = 0; l < L.P.D; ++l) HIP_TRY(hipMemcpyAsync(bias + l * 128, p->base_b[l], 128 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(bias + 512, p->rgbh_b, 64 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(bias + 576, p->sigma_b, 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(bias + 580, p->rgbo_b, 12, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(bg_pack_kernel, dim3((unsigned)L.total_blocks), dim3(64), 0, st, dtab, reinterpret_cast<uint4 *>(base));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

int tvr_npp_bg_points(const void *rays_o, const void *rays_d, int64_t n_rays, const void *z_lin, int32_t n_samples, const void *t_rand, float radii,
                      void *pts, void *z, void *stream)
{
    if (n_rays < 0 || n_samples < 1 || !(radii > 0.f)) return tvr_set_error(TVR_ERR_INVALID, "tvr_npp_bg_points: n_rays < 0, n_samples < 1 or radii <= 0");
    if (n_rays == 0) return TVR_OK;
    if (!rays_o || !rays_d || !z_lin || !t_rand || !pts || !z || misaligned(pts)) return tvr_set_error(TVR_ERR_INVALID, "tvr_npp_bg_points: NULL or misaligned argument");
    const long long total = (long long)n_rays * n_samples;
    hipLaunchKernelGGL(bg_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float *>(rays_o),
                       static_cast<const float *>(rays_d), (long long)n_rays, static_cast<const float *>(z_lin), (int)n_samples, static_cast<const float *>(t_rand),
                       radii, static_cast<float4 *>(pts), static_cast<float *>(z));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

int tvr_npp_bg_composite(const void *rgb, const void *sigma, const void *z, int64_t n_rays, int32_t n_samples, void *rgb_out, void *stream)
{
    if (n_rays < 0 || n_samples < 1) return tvr_set_error(TVR_ERR_INVALID, "tvr_npp_bg_composite: n_rays < 0 or n_samples < 1");
    if (n_rays == 0) return TVR_OK;
    if (!rgb || !sigma || !z || !rgb_out) return tvr_set_error(TVR_ERR_INVALID, "tvr_npp_bg_composite: NULL argument");
    hipLaunchKernelGGL(bg_composite_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float *>(rgb),
                       static_cast<const float *>(sigma), static_cast<const float *>(z), (long long)n_rays, (int)n_samples, static_cast<float *>(rgb_out));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

}  // extern "C"
