/* oracle/ngp_oracle.c — scalar-C restatement of the reference's alt path (BASELINE configs[4]):
 * JNeRF Instant-NGP inference = occupancy-bitfield ray march -> hash-grid + SH encoders -> two small MLPs -> compositing.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product path (jittor-myc-nerfs_amd/) never does.
 *
 * PARITY UNPINNED: the reference runs these steps as CUDA kernels inside Jittor `jt.code` ops; neither Jittor nor a CUDA
 * toolchain exists here (the headers include "cuda_fp16.h" and Jittor's "utils/log.h"), so nothing of it can be built or run,
 * and two of its kernels (`compute_rgbs_inference`, `mlp_fused_forward`) ship as sm_80 objects without source.  What pins
 * this file: the published PCG32 known-answer vector (tests/test_ngp_oracle.py), closed-form cases, and a second,
 * vectorised numpy restatement (oracle/ngp_oracle.py) of the encoders / networks / compositing.
 *
 * Reference lines followed (all under jnerf-myc/python/jnerf/):
 *   march        models/samplers/density_grid_sampler/op_header/ray_sampler.h:4-114, ray_sampler_header.h:60-78 (mip_from_pos/dt),
 *                :408-462 (ray_intersect), :472-477 (contains), :642-656 (morton3D), :728-776 (voxel skip, cascaded index, bit test),
 *                :790-843 (warps); dt from density_grid_sampler.py:94-113; RNG ops/op_include/pcg32/pcg32.h (PCG32, O'Neill 2014)
 *   bitfield     op_header/update_bitfield.h:23-70, update_bitfield.py:14-31
 *   hash grid    models/position_encoders/hash_encoder/op_header/HashEncode.h:69-93,107-115,117-199; hash from
 *                projects/ngp/configs/ngp_comp.py:89
 *   SH           models/position_encoders/sh_encoder/op_header/SphericalEncode.h:44-100 (degree 4: the 16 real SH polynomials)
 *   networks     models/networks/ngp_network.py:60-68,78-85 (bias-free Linear stacks, fp32 configs Car.py / Easyship.py)
 *   compositing  models/samplers/density_grid_sampler/op_header/calc_rgb.h:45-60 (declaration only) + calc_rgb.py:118-150 +
 *                ray_sampler_header.h:889-939 (activations): restated from the published Instant-NGP / JNeRF algorithm.
 *
 * Rounding assumptions (nvcc's default -fmad=true contracts a*b+c written in one expression): `o + t*d`, `startt += dt*u`,
 * `x*scale + 0.5` and `result += w*v` are single-rounded FMAs here (fmaf) and in the HIP kernels; everything else is
 * separately rounded (built with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>

#define NGP_GRIDSIZE 128u
#define NGP_CASCADES 5u
#define NGP_STEPS 1024u
#define NGP_RANDS_PER_RAY 8u

/* ------------------------------------------------------------------ PCG32 (pcg32.h; constants are the published ones) */
typedef struct { uint64_t state, inc; } ngp_rng;
#define PCG_MULT 0x5851f42d4c957f2dULL

static uint32_t rng_next(ngp_rng *r)
{
    uint64_t old = r->state;
    r->state = old * PCG_MULT + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
void ngp_rng_seed(ngp_rng *r, uint64_t initstate, uint64_t initseq)
{
    r->state = 0u;
    r->inc = (initseq << 1u) | 1u;
    rng_next(r);
    r->state += initstate;
    rng_next(r);
}
void ngp_rng_advance(ngp_rng *r, uint64_t delta)
{
    uint64_t cur_mult = PCG_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}
uint32_t ngp_rng_next_uint(ngp_rng *r) { return rng_next(r); }
float ngp_rng_next_float(ngp_rng *r)
{
    union { uint32_t u; float f; } x;
    x.u = (rng_next(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

/* ------------------------------------------------------------------ march helpers */
typedef struct {
    float lo[3], hi[3];       /* aabb (dataset.py:214-215: 0.5 -+ aabb_scale/2) */
    float near_distance;      /* cfg near_distance */
    float cone_angle;         /* cfg cone_angle_constant */
    int32_t const_dt;         /* cfg const_dt */
    uint32_t slab_rays;       /* 0: one call of the reference; k>0: the rays are consecutive k-ray slabs of render_img's loop
                                 (runner.py:209-222), the global generator advancing by 2^32 between slabs (ray_sampler.py:61) */
} ngp_march_cfg;

static float min_cone_step(void) { return 1.73205080757f / (float)NGP_STEPS; }
static float max_cone_step(void) { return min_cone_step() * (float)(1u << (NGP_CASCADES - 1)) * (float)NGP_STEPS / (float)NGP_GRIDSIZE; }
static float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
static float calc_dt(const ngp_march_cfg *c, float t)
{
    if (c->const_dt) return (float)((double)min_cone_step() * 0.5);
    return clampf(t * c->cone_angle, min_cone_step(), max_cone_step());
}
static int mip_from_pos(const float p[3])
{
    float m = fmaxf(fmaxf(fabsf(p[0] - 0.5f), fabsf(p[1] - 0.5f)), fabsf(p[2] - 0.5f));
    int e;
    frexpf(m, &e);
    int v = e + 1;
    if (v < 0) v = 0;
    return v < (int)NGP_CASCADES - 1 ? v : (int)NGP_CASCADES - 1;
}
static int mip_from_dt(float dt, const float p[3])
{
    int mip = mip_from_pos(p);
    dt *= (float)(2 * NGP_GRIDSIZE);
    if (dt < 1.f) return mip;
    int e;
    frexpf(dt, &e);
    int v = e > mip ? e : mip;
    return v < (int)NGP_CASCADES - 1 ? v : (int)NGP_CASCADES - 1;
}
static uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
uint32_t ngp_morton3d(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
uint32_t ngp_morton3d_invert(uint32_t x)
{
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
uint32_t ngp_cascaded_grid_idx_at(const float pos[3], uint32_t mip)
{
    float s = scalbnf(1.0f, -(int)mip);
    int i[3];
    for (int k = 0; k < 3; ++k) {
        float p = pos[k] - 0.5f;
        p = p * s;
        p = p + 0.5f;
        i[k] = (int)(p * (float)NGP_GRIDSIZE);
    }
    return ngp_morton3d((uint32_t)clampi(i[0], 0, NGP_GRIDSIZE - 1), (uint32_t)clampi(i[1], 0, NGP_GRIDSIZE - 1),
                        (uint32_t)clampi(i[2], 0, NGP_GRIDSIZE - 1));
}
static int occupied_at(const float pos[3], const uint8_t *bits, uint32_t mip)
{
    uint32_t idx = ngp_cascaded_grid_idx_at(pos, mip);
    return bits[idx / 8 + (NGP_GRIDSIZE * NGP_GRIDSIZE * NGP_GRIDSIZE) * mip / 8] & (1u << (idx % 8));
}
static float signf(float x) { return copysignf(1.0f, x); }
static float distance_to_next_voxel(const float pos[3], const float dir[3], const float idir[3], uint32_t res)
{
    float t[3];
    for (int k = 0; k < 3; ++k) {
        float p = (float)res * pos[k];
        t[k] = (floorf(p + 0.5f + 0.5f * signf(dir[k])) - p) * idir[k];
    }
    float m = fminf(fminf(t[0], t[1]), t[2]);
    return fmaxf(m / (float)res, 0.0f);
}
static float advance_to_next_voxel(const ngp_march_cfg *c, float t, const float pos[3], const float dir[3], const float idir[3], uint32_t res)
{
    float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
    do { t += calc_dt(c, t); } while (t < t_target);
    return t;
}
static void ray_intersect(const ngp_march_cfg *c, const float o[3], const float d[3], float out[2])
{
    float tmin = (c->lo[0] - o[0]) / d[0], tmax = (c->hi[0] - o[0]) / d[0], s;
    if (tmin > tmax) { s = tmin; tmin = tmax; tmax = s; }
    float tymin = (c->lo[1] - o[1]) / d[1], tymax = (c->hi[1] - o[1]) / d[1];
    if (tymin > tymax) { s = tymin; tymin = tymax; tymax = s; }
    if (tmin > tymax || tymin > tmax) { out[0] = out[1] = FLT_MAX; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (c->lo[2] - o[2]) / d[2], tzmax = (c->hi[2] - o[2]) / d[2];
    if (tzmin > tzmax) { s = tzmin; tzmin = tzmax; tzmax = s; }
    if (tmin > tzmax || tzmin > tmax) { out[0] = out[1] = FLT_MAX; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    out[0] = tmin; out[1] = tmax;
}
static int contains(const ngp_march_cfg *c, const float p[3])
{
    return p[0] >= c->lo[0] && p[0] <= c->hi[0] && p[1] >= c->lo[1] && p[1] <= c->hi[1] && p[2] >= c->lo[2] && p[2] <= c->hi[2];
}
static void point_at(const float o[3], const float d[3], float t, float p[3])
{
    for (int k = 0; k < 3; ++k) p[k] = fmaf(t, d[k], o[k]);
}
static float warp_dt(float dt)
{
    float max_step = min_cone_step() * (float)(1u << (NGP_CASCADES - 1));
    return (dt - min_cone_step()) / (max_step - min_cone_step());
}
static float unwarp_dt(float dt)
{
    float max_step = min_cone_step() * (float)(1u << (NGP_CASCADES - 1));
    return dt * (max_step - min_cone_step()) + min_cone_step();
}

/* One ray of rays_sampler: pass 1 (count) when coords == NULL, pass 2 (write `numsteps` entries) otherwise.  Returns the step count. */
static uint32_t march_ray(const ngp_march_cfg *c, const float o[3], const float d[3], const uint8_t *bits, float startt,
                          uint32_t limit, float *coords)
{
    float idir[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    float wdir[3] = {(d[0] + 1.0f) * 0.5f, (d[1] + 1.0f) * 0.5f, (d[2] + 1.0f) * 0.5f};
    uint32_t j = 0;
    float t = startt, pos[3];
    while (point_at(o, d, t, pos), contains(c, pos) && j < limit) {
        float dt = calc_dt(c, t);
        uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
        if (occupied_at(pos, bits, mip)) {
            if (coords) {
                float *q = coords + 7 * (size_t)j;
                for (int k = 0; k < 3; ++k) q[k] = (pos[k] - c->lo[k]) / (c->hi[k] - c->lo[k]);
                q[3] = warp_dt(dt);
                q[4] = wdir[0]; q[5] = wdir[1]; q[6] = wdir[2];
            }
            ++j;
            t += dt;
        } else {
            t = advance_to_next_voxel(c, t, pos, d, idir, NGP_GRIDSIZE >> mip);
        }
    }
    return j;
}

/* rays_sampler over n_rays.  Bases are the exclusive prefix sum of the step counts in ray order (the reference hands them out with
 * an atomicAdd in arrival order: same per-ray contents, arbitrary slab order).  numsteps_out [R,2] = (steps, base); counter_out =
 * (rays that got a slab, total steps); ray_index_out: rank among rays that got a slab, -1 for a ray without steps. */
int64_t ngp_oracle_sample(const ngp_march_cfg *c, const float *rays_o, const float *rays_d, int64_t n_rays, const uint8_t *bits,
                          uint64_t rng_state, uint64_t rng_inc, uint32_t max_samples, float *coords_out, int32_t *numsteps_out,
                          int32_t *ray_index_out, uint32_t *counter_out, float *startt_out)
{
    uint32_t base = 0, ray_counter = 0;
    for (int64_t i = 0; i < n_rays; ++i) {
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        ngp_rng r = {rng_state, rng_inc};
        uint32_t slab = c->slab_rays ? (uint32_t)(i / c->slab_rays) : 0u, in_slab = c->slab_rays ? (uint32_t)(i % c->slab_rays) : (uint32_t)i;
        for (uint32_t k = 0; k < slab; ++k) ngp_rng_advance(&r, 1ull << 32);
        ngp_rng_advance(&r, (uint64_t)(int64_t)(uint32_t)(in_slab * NGP_RANDS_PER_RAY));
        float tm[2];
        ray_intersect(c, o, d, tm);
        float startt = fmaxf(tm[0], c->near_distance);
        startt = fmaf(calc_dt(c, startt), ngp_rng_next_float(&r), startt);
        if (startt_out) startt_out[i] = startt;
        uint32_t n = march_ray(c, o, d, bits, startt, NGP_STEPS, NULL);
        uint32_t b = base;
        base += n;
        if (b + n > max_samples) {
            numsteps_out[2 * i] = 0; numsteps_out[2 * i + 1] = (int32_t)b;
            if (ray_index_out) ray_index_out[i] = -1;
            continue;
        }
        numsteps_out[2 * i] = (int32_t)n; numsteps_out[2 * i + 1] = (int32_t)b;
        if (ray_index_out) ray_index_out[i] = n ? (int32_t)ray_counter : -1;
        ++ray_counter;
        if (n && coords_out) march_ray(c, o, d, bits, startt, n, coords_out + 7 * (size_t)b);
    }
    counter_out[0] = ray_counter; counter_out[1] = base;
    return (int64_t)base;
}

/* ------------------------------------------------------------------ bitfield (update_bitfield.h) */
void ngp_oracle_update_bitfield(const float *density_grid, uint8_t *bitfield, float *mean_out)
{
    const uint32_t n = NGP_GRIDSIZE * NGP_GRIDSIZE * NGP_GRIDSIZE;
    double acc = 0;
    for (uint32_t i = 0; i < n; ++i) acc += (double)(fmaxf(density_grid[i], 0.f) / (float)n);
    float mean = (float)acc;
    *mean_out = mean;
    float thresh = 0.01f < mean ? 0.01f : mean;
    for (uint32_t i = 0; i < n / 8 * NGP_CASCADES; ++i) {
        uint8_t b = 0;
        for (int j = 0; j < 8; ++j) b |= density_grid[(size_t)i * 8 + j] > thresh ? (uint8_t)(1u << j) : 0;
        bitfield[i] = b;
    }
    for (uint32_t level = 1; level < NGP_CASCADES; ++level) {
        const uint8_t *prev = bitfield + (size_t)n * (level - 1) / 8;
        uint8_t *next = bitfield + (size_t)n * level / 8;
        for (uint32_t i = 0; i < n / 64; ++i) {
            uint8_t b = 0;
            for (int j = 0; j < 8; ++j) b |= prev[(size_t)i * 8 + j] > 0 ? (uint8_t)(1u << j) : 0;
            uint32_t x = ngp_morton3d_invert(i >> 0) + NGP_GRIDSIZE / 8;
            uint32_t y = ngp_morton3d_invert(i >> 1) + NGP_GRIDSIZE / 8;
            uint32_t z = ngp_morton3d_invert(i >> 2) + NGP_GRIDSIZE / 8;
            next[ngp_morton3d(x, y, z)] |= b;
        }
    }
}

/* ------------------------------------------------------------------ hash grid (HashEncode.h kernel_grid) */
typedef struct {
    int32_t n_levels;
    uint32_t offsets[33];     /* grid_encode.py:24-39, entries (not floats) */
    float scale[32];          /* kernel_grid: exp2f(level*log2_per_level_scale)*base_resolution - 1, evaluated on the host */
} ngp_grid_cfg;

static uint32_t grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t p[3])
{
    uint32_t stride = 1, index = 0;
    for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
        index += p[dim] * stride;
        stride *= res;
    }
    if (hashmap_size < stride) index = p[0] ^ (p[1] * 19349663u) ^ (p[2] * 83492791u);
    return (index % hashmap_size) * 2u;
}
/* pos [n,3] in [0,1] -> out [n, 2*n_levels]; cell_out (optional) [n, n_levels, 3] integer cell of the low corner */
void ngp_oracle_hash_encode(const ngp_grid_cfg *g, const float *grid, const float *pos, int64_t n, float *out, uint32_t *cell_out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int l = 0; l < g->n_levels; ++l) {
            const float *tab = grid + (size_t)g->offsets[l] * 2;
            uint32_t size = g->offsets[l + 1] - g->offsets[l];
            float scale = g->scale[l];
            uint32_t res = (uint32_t)ceilf(scale) + 1u;
            float f[3];
            uint32_t c[3];
            for (int k = 0; k < 3; ++k) {
                float p = fmaf(pos[3 * i + k], scale, 0.5f);
                int t = (int)floorf(p);
                c[k] = (uint32_t)t;
                f[k] = p - (float)t;
                if (cell_out) cell_out[((size_t)i * g->n_levels + l) * 3 + k] = c[k];
            }
            float r0 = 0.f, r1 = 0.f;
            for (uint32_t idx = 0; idx < 8; ++idx) {
                float w = 1;
                uint32_t q[3];
                for (int k = 0; k < 3; ++k) {
                    if ((idx & (1u << k)) == 0) { w *= 1 - f[k]; q[k] = c[k]; }
                    else { w *= f[k]; q[k] = c[k] + 1; }
                }
                uint32_t e = grid_index(size, res, q);
                r0 = fmaf(w, tab[e], r0);
                r1 = fmaf(w, tab[e + 1], r1);
            }
            out[(size_t)i * 2 * g->n_levels + 2 * l] = r0;
            out[(size_t)i * 2 * g->n_levels + 2 * l + 1] = r1;
        }
}

/* ------------------------------------------------------------------ SH degree 4 (SphericalEncode.h:60-100) */
void ngp_oracle_sh_encode(const float *dir01, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float x = dir01[3 * i] * 2.f - 1.f, y = dir01[3 * i + 1] * 2.f - 1.f, z = dir01[3 * i + 2] * 2.f - 1.f;
        float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        float *o = out + 16 * i;
        o[0] = 0.28209479177387814f;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
        o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
}

/* ------------------------------------------------------------------ networks (ngp_network.py:60-68,78-85) */
typedef struct {
    const float *d0, *d1;        /* density_mlp: Linear(32,64) [64,32], Linear(64,16) [16,64], no bias */
    const float *c0, *c1, *c2;   /* rgb_mlp: Linear(32,64) [64,32], Linear(64,64) [64,64], Linear(64,3) [3,64] */
} ngp_net;

static void linear(const float *W, const float *x, int n_in, int n_out, float *y, int relu)
{
    for (int o = 0; o < n_out; ++o) {
        float a = 0.f;
        for (int k = 0; k < n_in; ++k) a += x[k] * W[o * n_in + k];
        y[o] = relu ? (a > 0.f ? a : 0.f) : a;
    }
}
/* coords [n,7] (pos, dt, dir) -> out [n,4] = (rgb raw, density raw) */
void ngp_oracle_network(const ngp_grid_cfg *g, const float *grid, const ngp_net *net, const float *coords, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float enc[32], h[64], den[16], in2[32], h2[64], h3[64], rgb[3];
        ngp_oracle_hash_encode(g, grid, coords + 7 * i, 1, enc, NULL);
        linear(net->d0, enc, 32, 64, h, 1);
        linear(net->d1, h, 64, 16, den, 0);
        memcpy(in2, den, sizeof den);
        ngp_oracle_sh_encode(coords + 7 * i + 4, 1, in2 + 16);
        linear(net->c0, in2, 32, 64, h2, 1);
        linear(net->c1, h2, 64, 64, h3, 1);
        linear(net->c2, h3, 64, 3, rgb, 0);
        out[4 * i] = rgb[0]; out[4 * i + 1] = rgb[1]; out[4 * i + 2] = rgb[2]; out[4 * i + 3] = den[0];
    }
}

/* ------------------------------------------------------------------ compositing (compute_rgbs_inference; activations logistic / exp) */
void ngp_oracle_composite(const float *net_out, const float *coords, const int32_t *numsteps, int64_t n_rays, const float bg[3],
                          float *rgb_out, float *T_out)
{
    for (int64_t i = 0; i < n_rays; ++i) {
        uint32_t n = (uint32_t)numsteps[2 * i], base = (uint32_t)numsteps[2 * i + 1];
        float T = 1.f, c[3] = {0, 0, 0};
        uint32_t j = 0;
        for (; j < n; ++j) {
            if (T < 1e-4f) break;
            const float *o = net_out + 4 * (size_t)(base + j);
            float dt = unwarp_dt(coords[7 * (size_t)(base + j) + 3]);
            float density = expf(o[3]);
            float alpha = 1.f - expf(-density * dt);
            float w = alpha * T;
            for (int k = 0; k < 3; ++k) c[k] += w * (1.0f / (1.0f + expf(-o[k])));
            T *= (1.f - alpha);
        }
        if (j == n)
            for (int k = 0; k < 3; ++k) c[k] += T * bg[k];
        rgb_out[3 * i] = c[0]; rgb_out[3 * i + 1] = c[1]; rgb_out[3 * i + 2] = c[2];
        if (T_out) T_out[i] = T;
    }
}
