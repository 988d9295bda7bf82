// tvr_device.h — device-side scene descriptor and shared helpers (gfx950 only).
//
// HBM layout ("packed scene", built by pack kernels from the reference layout):
//   density plane i : [H_i+1][W_i+1][16] fp32  (channels-last: one texel = 64 B = 4 x float4; the extra
//                      row/column is zero, which realises grid_sample's zeros padding for the +1 taps)
//   density line  i : [L_i+1][16]
//   app plane     i : [H_i+1][W_i+1][48]       (192 B texel = 12 x float4)
//   app line      i : [L_i+1][48]
//   MLP LDS image (TVR_MLP_IMAGE_BYTES): W1 / W2 as fp16 hi and lo parts, k-step major [k-step][lane half][128 hidden][8 halfs]
//                      (conflict-free ds_read_b128 A-operand reads, no padding) with permuted k columns (see tvr_shade.hip), b2, b3, W3 fp32,
//                      and the hi parts of the basis fragments; the lo parts [k-step][lane half][row 32][8 halfs] stay in global memory
// Compiled with -ffp-contract=off: every a*b+c below is two rounded ops unless written as fmaf, so the
// position / mask / cell-index arithmetic reproduces SURVEY.md Appendix A steps 1-7 bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TVR_CD 16      // density components per plane
#define TVR_CA 48      // appearance components per plane
#define TVR_APPDIM 27
#define TVR_FEATC 128
#define TVR_NIN 150    // 27 + 3 + 2*2*27 + 2*2*3
#define TVR_KAPP 144

// byte offsets inside the MLP LDS image.  Round 2: the weight images are K-STEP MAJOR — [k-step][lane half][128 rows][8 halfs = 16 B] — so the 32
// lanes of a ds_read_b128 A-fragment read (row = lane & 31) touch 512 contiguous bytes: conflict-free without the 16 B of padding per row
// that the row-major layout of round 1 needed.  The 8 KB this frees hold the hi parts of the basis fragments (27 rows): 9 of the 18 basis
// loads per tile no longer go through the vector L1, which is what binds the shade kernel.
#define TVR_IMG_STEP 4096                  // bytes per k-step of one image: 2 halves x 128 rows x 16 B
#define TVR_IMG_RB 512                     // bytes between 32-row blocks inside a (k-step, half)
#define TVR_IMG_W1H 0
#define TVR_IMG_W1L (10 * TVR_IMG_STEP)
#define TVR_IMG_W2H (20 * TVR_IMG_STEP)
#define TVR_IMG_W2L (28 * TVR_IMG_STEP)
#define TVR_IMG_B1 (36 * TVR_IMG_STEP)             // 147 456
#define TVR_IMG_B2 (TVR_IMG_B1 + 512)
#define TVR_IMG_B3 TVR_IMG_B1                      // b3 (3 floats + pad): b1 itself rides in W1's image as the column of a constant-1 input
#define TVR_IMG_W3 (TVR_IMG_B2 + 512)              // W3 [3][128] fp32 (layer 3 runs as fp32 FMAs), then 512 unused zero bytes
#define TVR_IMG_W3_ROW 512
#define TVR_IMG_BASH (TVR_IMG_W3 + 4 * TVR_IMG_W3_ROW)   // basis fragments, hi parts: [9 k-steps][2 halves][27 rows][16 B]
#define TVR_IMG_BASH_ROWS 27
#define TVR_MLP_IMAGE_BYTES (TVR_IMG_BASH + 9 * 2 * TVR_IMG_BASH_ROWS * 16)     // 158 304 B of the 163 840 B LDS
// REFTensoRF (variant 1) appends the four 144 -> {3,3,1,1} linears of REFTensoRF.compute_appfeature (models/REFTensoRF.py:126-132):
//   576 zero bytes (the A-operand row of the lanes that own no head: 9 k-steps x 2 halves x 32 B, all of it zero — round 2 pointed this at
//   W3's 512-B zero row, whose k-step 8 ran into the basis fragments), 8 rows [9 k-steps][2 halves][hi 8 | lo 8] fp16
//   (normal 0..2, specular 3, diffuse 4..6, rho 7) and 16 fp32 biases in accumulator-row order.
#define TVR_IMG_REF_ROW 576
#define TVR_IMG_REF_ZROW TVR_MLP_IMAGE_BYTES
#define TVR_IMG_REFW (TVR_MLP_IMAGE_BYTES + TVR_IMG_REF_ROW)
#define TVR_IMG_REFB (TVR_IMG_REFW + 8 * TVR_IMG_REF_ROW)
#define TVR_MLP_IMAGE_BYTES_REF (TVR_IMG_REFB + 64)               // 163 552 B (+ 16 B of matrix tokens <= 163 840)
static_assert(TVR_MLP_IMAGE_BYTES_REF + 16 <= 160 * 1024, "REFTensoRF's LDS image must fit the CU's 160 KB");
// General encoding frequencies (round 4): TensorBase.__init__'s own defaults are view_pe = fea_pe = 6 (tensorBase.py:141-145; opt.py:84,104) — 390 MLP inputs, a
// layer-1 image of 213 KB that no LDS holds.  Scenes with more than two frequencies run the same kernel with layer 1 in LOCKSTEP: every base value has
// TVR_GEN_T = 1 + 2 * 6 derived values {v, sin(2^f v), cos(2^f v)}, a lane's 16 base values fill 26 k-steps, and the workgroup's eight waves stage one k-step
// (8 KB of fragments) at a time through two LDS slots, with a barrier per k-step.  Slower than the two-frequency kernel (no phase alternation, W1 re-read from
// L2 for every 256 entries) — it exists so that every shape the reference constructs renders through HIP kernels; the shipped configs all use 2 / 2.
#define TVR_GEN_PE 6
#define TVR_GEN_T (1 + 2 * TVR_GEN_PE)
#define TVR_GEN_KS (2 * TVR_GEN_T)                          // 16 base values x 13 derived / 8 slots per k-step and lane
#define TVR_W1GEN_BYTES (TVR_GEN_KS * 2 * TVR_IMG_STEP)     // 212 992
#define TVR_NIN_REF 151  // 1 + 27 + 3 + 2*2*27 + 2*2*3 (MLPRender_Fea_Ref, models/REFTensoRF.py:9)
#define TVR_BASIS_FRAG_BYTES (9 * 2 * 32 * 16)   // global: the LO parts of the basis fragments [9 k-steps][2 halves][32 rows][16 B] (hi parts: LDS image)

// ---- round 5: the render path's MLP on v_mfma_f32_16x16x32_f16 (csrc/tvr_shade16.hip) -----------------------------------------------------------------
// A fragment of that shape = 16 rows x 32 k = 64 lanes x 16 B = 1 KB, lane (i = lane & 15, g = lane >> 4) holding row i, k = 8g .. 8g+7 of the k-step; stored as the
// lanes read it ([fragment][lane][8 halves]: a ds_read_b128 of 64 consecutive 16-B slots is conflict-free).  One {hi, lo} pair feeds SIX MFMAs: two 16-column B tiles
// x three products — the LDS bytes per FLOP of the 32x32x16 form, on the shape that delivers 1.10 x its FLOP/s at the chip's power limit (profiles/r05_mfma_shape_probe.txt).
//   W1  [5 k-steps][8 row blocks] fragments, hi | lo;  k slot 8s + j of lane group g  <->  derived value (8s + j) % 5 of base value (8s + j) / 5, base value r of group g
//       = row 4g + r (r < 4) or 16 + 4g + r - 4 of the 32-row feature tile (rows 27..29 view direction, 31 the constant 1 whose column is b1)
//   W2  [4][8] fragments, hi | lo;  k slot j of k-step s, group g  <->  hidden unit 32s + 4g + j (j < 4) or 32s + 16 + 4g + j - 4 (the layer-1 accumulators as they lie)
//   b2 [128], b3 [3 + pad], W3 [3][128] fp32, 16 zero bytes
//   basis (27 x 144): per k-step s < 4 {row block 0: 64 lanes x 16 B; row block 1: rows 16..26 only, [g][11][16 B]} = 1728 B, k-step 4 (k = 128..143: groups 0, 1
//       only; groups 2, 3 read the zero bytes) 864 B; hi parts all five k-steps, lo parts k-steps 0, 1, 2, 4 — the lo parts of k-step 3 (2 KB as two full fragments)
//       stay in global memory and are fetched once per tile: 160 KB of LDS hold no more
#define TVR16_FRAG 1024
#define TVR16_W1H 0
#define TVR16_W1L (40 * TVR16_FRAG)
#define TVR16_W2H (80 * TVR16_FRAG)
#define TVR16_W2L (112 * TVR16_FRAG)
#define TVR16_B2 (144 * TVR16_FRAG)
#define TVR16_B3 (TVR16_B2 + 512)
#define TVR16_W3 (TVR16_B3 + 16)
#define TVR16_ZERO (TVR16_W3 + 3 * 512)
#define TVR16_BASH (TVR16_ZERO + 16)
#define TVR16_BAS_STEP 1728
#define TVR16_BAS_RB1 1024
#define TVR16_BAS_ROWS1 11                           // rows 16..26 of the basis: row block 1 keeps these
#define TVR16_BAS_S4 (4 * TVR16_BAS_STEP)
#define TVR16_BAS_S4_RB1 512
#define TVR16_BAS_BYTES (TVR16_BAS_S4 + 864)         // 7776 = 27 x 144 x 2
#define TVR16_BASL (TVR16_BASH + TVR16_BAS_BYTES)    // lo parts: k-steps 0..2 at s * 1728, k-step 4 at 3 * 1728
#define TVR16_BASL_S4 (3 * TVR16_BAS_STEP)
#define TVR16_IMAGE_BYTES (TVR16_BASL + TVR16_BAS_BYTES - TVR16_BAS_STEP)       // 163 360 B (+ 16 B of matrix tokens <= 163 840)
#define TVR16_REFG_BYTES (10 * TVR16_FRAG + 256)
#define TVR16_BASG_BYTES (2 * TVR16_FRAG)            // global: lo parts of basis k-step 3, row blocks 0 and 1 as full fragments (rows >= 27 zero)
static_assert(TVR16_IMAGE_BYTES + 16 <= 160 * 1024, "the 16x16x32 LDS image must fit the CU's 160 KB");
static_assert(TVR16_W3 % 16 == 0 && TVR16_BASH % 16 == 0, "float4 / uint4 reads need 16-B alignment");

struct SceneDev {
    float lo[3], hi[3], inv[3];
    float gm1[3];                 // float(grid-1)
    int grid[3];
    const float4 *dplane[3];      // packed
    const float4 *dline[3];
    const float4 *aplane[3];
    const float4 *aline[3];
    const uint4 *aplane16[3];     // the appearance factors once more as fp16 (same channels-last layout, 96 B per texel): what the one-product arithmetic (TVR_ARITH_F16) gathers;
    const uint4 *aline16[3];      // converted from the fp32 images by tvr_scene_update / the first render in that mode (tvr_api.hip: h16_stale)
    const void *mlp_image;        // TVR_MLP_IMAGE_BYTES(_REF), copied to LDS by the shade kernel
    const void *basis_frag;       // lo parts of the basis fragments [9][2][32][8] fp16 (hi parts: mlp_image + TVR_IMG_BASH)
    const float *b3;              // [3]
    const void *img16;            // TVR16_IMAGE_BYTES: the LDS image of the 16x16x32 render kernel (TensorVMSplit scenes with at most two encoding frequencies), or nullptr
    const void *basg16;           // TVR16_BASG_BYTES
    const void *refg16;           // REFTensoRF (round 6): TVR16_REFG_BYTES — the four heads as the third row block of the basis product: [5 k-steps][hi | lo] full fragments + 16 bias floats
    float near_, far_, step, shift, scale, thres;
    int act;
    int variant;                  // 0 TensorVMSplit, 1 REFTensoRF
    int gen;                      // 1: more than two encoding frequencies (view_pe / fea_pe up to TVR_GEN_PE): layer 1 runs from the streamed image w1gen
    const void *w1gen;            // [26 k-steps][hi 4 KB | lo 4 KB], each [2 halves][128 rows][8 halfs]
    int range_check;              // 1 (default): the inference shade kernels mark entries whose fp16-split operands leave fp16's range with NaN (tvr_scene_set_range_check)
    int arith;                    // TVR_ARITH_*: the products per k-step of the render / mlp_render shade kernels (tvr_scene_set_arith); 0 = three (fp32-class)
    const float *avol;            // (gz,gy,gx) or nullptr
    const unsigned *abits;        // optional: bit ((z*gy + y)*gx + x) = (avol > 0), built by tvr_scene_set_alpha
    int ag[3];
    float alo[3], ainv[3], agm1[3];
};

static const __device__ int kMat[3][2] = {{0, 1}, {0, 2}, {1, 2}};   // tensorBase.py:168
static const __device__ int kVec[3] = {2, 1, 0};                     // tensorBase.py:169

__device__ __forceinline__ float unnorm(float c, float gm1) { return ((c + 1.0f) / 2.0f) * gm1; }

__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 acc)
{
    return make_float4(__builtin_fmaf(s, a.x, acc.x), __builtin_fmaf(s, a.y, acc.y),
                       __builtin_fmaf(s, a.z, acc.z), __builtin_fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float4 f4_mul(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// One quad-lane's share (4 of C channels, selected by `sub`) of  bilinear(plane)*linear(line)  for one
// sample.  P: packed plane, texel = TPT float4s; Wp = W+1 (padded row length in texels).
// CHECK=false assumes 0<=x0<=W-1, 0<=y0<=H-1, 0<=l0<=L-1 (in-box samples): the +1 taps land on the zero pad.
template <int TPT, bool CHECK>
__device__ __forceinline__ float4 vm_term(const float4 *__restrict__ P, const float4 *__restrict__ Ln, int W, int H, int L,
                                          int x0, int y0, int l0, float wx, float wy, float wl, int sub)
{
    // weights in the grid_sampler formulation: (x1 - fx) etc.
    const float ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
    const int Wp = W + 1;
    float4 t00, t01, t10, t11, l0v, l1v;
    if (!CHECK) {
        const float4 *p = P + ((size_t)y0 * Wp + x0) * TPT + sub;
        t00 = p[0];
        t01 = p[TPT];
        t10 = p[(size_t)Wp * TPT];
        t11 = p[(size_t)Wp * TPT + TPT];
        const float4 *q = Ln + (size_t)l0 * TPT + sub;
        l0v = q[0];
        l1v = q[TPT];
    } else {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool xi0 = (x0 >= 0) & (x0 < W), xi1 = (x0 + 1 >= 0) & (x0 + 1 < W);
        const bool yi0 = (y0 >= 0) & (y0 < H), yi1 = (y0 + 1 >= 0) & (y0 + 1 < H);
        const bool li0 = (l0 >= 0) & (l0 < L), li1 = (l0 + 1 >= 0) & (l0 + 1 < L);
        const int xc = min(max(x0, 0), W - 1), yc = min(max(y0, 0), H - 1), lc = min(max(l0, 0), L - 1);
        const int xd = min(max(x0 + 1, 0), W - 1), yd = min(max(y0 + 1, 0), H - 1), ld = min(max(l0 + 1, 0), L - 1);
        t00 = (xi0 & yi0) ? P[((size_t)yc * Wp + xc) * TPT + sub] : z;
        t01 = (xi1 & yi0) ? P[((size_t)yc * Wp + xd) * TPT + sub] : z;
        t10 = (xi0 & yi1) ? P[((size_t)yd * Wp + xc) * TPT + sub] : z;
        t11 = (xi1 & yi1) ? P[((size_t)yd * Wp + xd) * TPT + sub] : z;
        l0v = li0 ? Ln[(size_t)lc * TPT + sub] : z;
        l1v = li1 ? Ln[(size_t)ld * TPT + sub] : z;
    }
    float4 p4 = f4_mul(ux * uy, t00);
    p4 = f4_fma(wx * uy, t01, p4);
    p4 = f4_fma(ux * wy, t10, p4);
    p4 = f4_fma(wx * wy, t11, p4);
    float4 q4 = f4_mul(ul, l0v);
    q4 = f4_fma(wl, l1v, q4);
    return make_float4(p4.x * q4.x, p4.y * q4.y, p4.z * q4.z, p4.w * q4.w);
}

// trilinear lookup of the alpha volume at world position p (AlphaGridMask.sample_alpha, tensorBase.py:50-59)
__device__ __forceinline__ float alpha_lookup(const SceneDev &sc, const float p[3])
{
    float f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float q = (p[k] - sc.alo[k]) * sc.ainv[k] - 1.0f;
        f[k] = unnorm(q, sc.agm1[k]);
    }
    const float x0f = floorf(f[0]), y0f = floorf(f[1]), z0f = floorf(f[2]);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const float tx = f[0] - x0f, ty = f[1] - y0f, tz = f[2] - z0f;
    const int W = sc.ag[0], H = sc.ag[1], D = sc.ag[2];
    float r = 0.0f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                float w = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty);
                w = w * (dz ? tz : 1.0f - tz);
                if ((x >= 0) & (x < W) & (y >= 0) & (y < H) & (z >= 0) & (z < D))
                    r = r + sc.avol[((size_t)z * H + y) * W + x] * w;
            }
    return r;
}

// `alpha_lookup(p) > 0` (the mask merge of tensorBase.py:491-496) from the bit volume: the trilinear sum of non-negative values is positive
// iff some in-range corner with a non-zero weight holds a non-zero value; the +1 corner of an axis has weight zero exactly when the
// fractional coordinate on that axis is zero.  Same result as the float path, 1/32 of its footprint and half its loads.
__device__ __forceinline__ bool alpha_positive(const SceneDev &sc, const float p[3])
{
    float f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float q = (p[k] - sc.alo[k]) * sc.ainv[k] - 1.0f;
        f[k] = unnorm(q, sc.agm1[k]);
    }
    const float x0f = floorf(f[0]), y0f = floorf(f[1]), z0f = floorf(f[2]);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const bool hx = (f[0] - x0f) > 0.0f, hy = (f[1] - y0f) > 0.0f, hz = (f[2] - z0f) > 0.0f;
    const int W = sc.ag[0], H = sc.ag[1], D = sc.ag[2];
    unsigned any = 0u;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = y0 + dy, z = z0 + dz;
            if ((dz && !hz) || (dy && !hy) || y < 0 || y >= H || z < 0 || z >= D) continue;
            const long long row = ((long long)z * H + y) * W;
            if (x0 >= 0 && x0 < W) { const long long b = row + x0; any |= (sc.abits[b >> 5] >> (b & 31)) & 1u; }
            if (hx && x0 + 1 >= 0 && x0 + 1 < W) { const long long b = row + x0 + 1; any |= (sc.abits[b >> 5] >> (b & 31)) & 1u; }
        }
    return any != 0u;
}

// sample_ray steps 1-3 (tensorBase.py:345-348): entry distance clamped to [near, far]
__device__ __forceinline__ float ray_tmin(const SceneDev &sc, const float o[3], const float d[3])
{
    float t = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = (d[k] == 0.0f) ? 1e-6f : d[k];
        const float ra = (sc.hi[k] - o[k]) / v, rb = (sc.lo[k] - o[k]) / v;
        const float m = ra < rb ? ra : rb;      // jt.minimum
        t = (m > t) ? m : t;                    // .max(-1)
    }
    return t < sc.near_ ? sc.near_ : (t > sc.far_ ? sc.far_ : t);
}

// clamp(x, 0, 1) of tensorBase.py:527 with the reference framework's NaN behaviour: a NaN stays a NaN (fminf / fmaxf return the OTHER operand and would turn a
// NaN sample — the shade kernel's "an operand left fp16's range" mark — into a valid-looking 0)
__device__ __forceinline__ float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// bijective XCD-aware remap: workgroups b, b+8, ... share an XCD (observed round-robin dispatch), so give each
// XCD a contiguous run of logical tiles -> neighbouring ray tiles share that XCD's L2.  Speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk)
{
    const unsigned q = nblk >> 3, r = nblk & 7u, x = b & 7u, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// accumulator register r of lane half h  <->  row of the 32x32 tile
__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// reference input index (tensorBase.py:77-82 concat order [features, viewdirs, PE(features, fea_pe), PE(viewdirs, view_pe)], PE = [sin | cos] with
// entry Ff*c + f inside each, tensorBase.py:9-15) of derived value t (0: v, 1: sin v, 2: sin 2v, 3: cos v, 4: cos 2v) of base value c
// (0..26 features, 27..29 view direction); -1 = the reference network has no such input (fea_pe / view_pe < 2): zero weight
__host__ __device__ __forceinline__ int ref_in_index(int c, int t, int fea_pe, int view_pe)
{
    const int f = (t == 2 || t == 4) ? 1 : 0, is_cos = t >= 3;
    const int off1 = TVR_APPDIM + 3, off2 = off1 + 2 * TVR_APPDIM * fea_pe;
    if (c < TVR_APPDIM) {
        if (t == 0) return c;
        return f < fea_pe ? off1 + (is_cos ? TVR_APPDIM * fea_pe : 0) + fea_pe * c + f : -1;
    }
    if (c < TVR_APPDIM + 3) {
        const int d = c - TVR_APPDIM;
        if (t == 0) return TVR_APPDIM + d;
        return f < view_pe ? off2 + (is_cos ? 3 * view_pe : 0) + view_pe * d + f : -1;
    }
    return -1;
}
__host__ __device__ __forceinline__ int ref_in_index(int c, int t) { return ref_in_index(c, t, 2, 2); }
// the same for the general slot order: derived value t of base value c is  v (t = 0), sin(2^(t-1) v) (1 <= t <= 6), cos(2^(t-7) v) (7 <= t <= 12)
__host__ __device__ __forceinline__ int gen_in_index(int c, int t, int fea_pe, int view_pe)
{
    const int f = t == 0 ? 0 : (t <= TVR_GEN_PE ? t - 1 : t - 1 - TVR_GEN_PE), is_cos = t > TVR_GEN_PE;
    const int off1 = TVR_APPDIM + 3, off2 = off1 + 2 * TVR_APPDIM * fea_pe;
    if (c < TVR_APPDIM) {
        if (t == 0) return c;
        return f < fea_pe ? off1 + (is_cos ? TVR_APPDIM * fea_pe : 0) + fea_pe * c + f : -1;
    }
    if (c < TVR_APPDIM + 3) {
        const int d = c - TVR_APPDIM;
        if (t == 0) return TVR_APPDIM + d;
        return f < view_pe ? off2 + (is_cos ? 3 * view_pe : 0) + view_pe * d + f : -1;
    }
    return -1;
}

