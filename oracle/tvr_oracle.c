/* CPU oracle (b): scalar per-ray / per-sample restatement of the TensoRF render path.
 *
 * TEST INFRASTRUCTURE ONLY — see oracle/tensorf_oracle.py for the rules: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: the reference (tensorf-myc, Python over Jittor) cannot be run here and
 * holds no golden vectors for this path.  This file follows the op order of
 * SURVEY.md Appendix A, derived from the reference lines cited per function (paths relative
 * to /root/reference/).  It is cross-checked against oracle/tensorf_oracle.py (torch CPU).
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off: every line below is one rounded fp32 op,
 * no FMA contraction, so sample positions / masks / cell indices are reproducible bit for bit).
 * Data layout is the reference's own: planes (1,C,H,W) channel-first, lines (1,C,L,1),
 * Linear weights [out,in] row-major.
 */
#include <alloca.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float aabb[6];            /* lo[3], hi[3]                       tensorBase.py:152 */
    int32_t grid[3];          /* gridSize (x,y,z)                   tensorBase.py:203 */
    int32_t cd[3], ca[3];     /* density_n_comp, appearance_n_comp  tensoRF.py:148-149 */
    int32_t app_dim, featC, view_pe, fea_pe;
    const float *dplane[3], *dline[3], *aplane[3], *aline[3];
    const float *basis;       /* [app_dim, sum(ca)]                 tensoRF.py:150 */
    const float *W1, *b1, *W2, *b2, *W3, *b3;   /* tensorBase.py:69-71 */
    float near_, far_, step, density_shift, distance_scale, thres;
    int32_t act;              /* 0 softplus, 1 relu                 tensorBase.py:444-448 */
    const float *alpha_vol;   /* (gz,gy,gx) or NULL                 tensorBase.py:47 */
    int32_t agrid[3];         /* gx,gy,gz */
    float alpha_aabb[6];
} oracle_scene;

typedef struct {              /* optional per-sample dumps, each [n*S] (cell: [n*S*3], rgb: [n*S*3]); NULL = skip */
    float *z; uint8_t *valid; uint8_t *bbox_valid; int32_t *cell; float *sf; float *sigma;
    float *alpha; float *weight; uint8_t *app; float *rgb; float *tmin; float *acc;
} oracle_dump;

static const int MAT[3][2] = {{0, 1}, {0, 2}, {1, 2}};  /* tensorBase.py:168 */
static const int VEC[3] = {2, 1, 0};                    /* tensorBase.py:169 */

/* grid_sample(align_corners=True) un-normalisation: ((c+1)/2)*(size-1) */
static inline float unnorm(float c, int size) { float a = c + 1.0f; float b = a / 2.0f; return b * (float)(size - 1); }

/* bilinear tap set, zeros padding.  Weights formed as products of (corner - coord) differences
 * (the grid_sampler_2d formulation the reference's F.grid_sample ports). */
static float bilinear(const float *img, int H, int W, float fx, float fy)
{
    float x0f = floorf(fx), y0f = floorf(fy);
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    float wnw = (x1f - fx) * (y1f - fy);
    float wne = (fx - x0f) * (y1f - fy);
    float wsw = (x1f - fx) * (fy - y0f);
    float wse = (fx - x0f) * (fy - y0f);
    float vnw = (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) ? img[(size_t)y0 * W + x0] : 0.0f;
    float vne = (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) ? img[(size_t)y0 * W + x1] : 0.0f;
    float vsw = (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) ? img[(size_t)y1 * W + x0] : 0.0f;
    float vse = (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) ? img[(size_t)y1 * W + x1] : 0.0f;
    float r = vnw * wnw;
    r = r + vne * wne;
    r = r + vsw * wsw;
    r = r + vse * wse;
    return r;
}

/* a line is a (L,1) image sampled at x = 0 (normalised) -> fx = ((0+1)/2)*(1-1) = 0: only x0 = 0 in range */
static float linear(const float *line, int L, float fl) { return bilinear(line, L, 1, 0.0f, fl); }

static float trilinear(const float *vol, int D, int H, int W, float fx, float fy, float fz)
{
    float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    float tx = fx - x0f, ty = fy - y0f, tz = fz - z0f;
    float r = 0.0f;
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                float w = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty);
                w = w * (dz ? tz : 1.0f - tz);
                if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D)
                    r = r + vol[((size_t)z * H + y) * W + x] * w;
            }
    return r;
}

/* tensoRF.py:209-225 — one sample, xyz_norm in [-1,1]^3 (any value allowed: zeros padding) */
float tvr_oracle_density_feature(const oracle_scene *s, const float n[3])
{
    float sf = 0.0f;
    for (int i = 0; i < 3; ++i) {
        int a = MAT[i][0], b = MAT[i][1], v = VEC[i];
        int W = s->grid[a], H = s->grid[b], L = s->grid[v];
        float fx = unnorm(n[a], W), fy = unnorm(n[b], H), fl = unnorm(n[v], L);
        float acc = 0.0f;
        for (int c = 0; c < s->cd[i]; ++c) {
            float p = bilinear(s->dplane[i] + (size_t)c * H * W, H, W, fx, fy);
            float q = linear(s->dline[i] + (size_t)c * L, L, fl);
            acc = acc + p * q;
        }
        sf = sf + acc;
    }
    return sf;
}

/* tensoRF.py:228-244 — h[sum ca] plane-major, f = basis · h */
void tvr_oracle_app_feature(const oracle_scene *s, const float n[3], float *f, float *h_out)
{
    int K = s->ca[0] + s->ca[1] + s->ca[2];
    float *h = (float *)alloca(sizeof(float) * K);
    int k = 0;
    for (int i = 0; i < 3; ++i) {
        int a = MAT[i][0], b = MAT[i][1], v = VEC[i];
        int W = s->grid[a], H = s->grid[b], L = s->grid[v];
        float fx = unnorm(n[a], W), fy = unnorm(n[b], H), fl = unnorm(n[v], L);
        for (int c = 0; c < s->ca[i]; ++c, ++k) {
            float p = bilinear(s->aplane[i] + (size_t)c * H * W, H, W, fx, fy);
            float q = linear(s->aline[i] + (size_t)c * L, L, fl);
            h[k] = p * q;
        }
    }
    for (int o = 0; o < s->app_dim; ++o) {
        float acc = 0.0f;
        for (int j = 0; j < K; ++j) acc = acc + s->basis[(size_t)o * K + j] * h[j];
        f[o] = acc;
    }
    if (h_out) memcpy(h_out, h, sizeof(float) * K);
}

/* tensorBase.py:9-15, :76-86 — MLP_Fea: in = [f, d, sin(PE f), cos(PE f), sin(PE d), cos(PE d)] */
void tvr_oracle_mlp(const oracle_scene *s, const float *f, const float d[3], float rgb[3], float *in_out)
{
    int A = s->app_dim, FP = s->fea_pe, VP = s->view_pe, C = s->featC;
    int nin = A + 3 + 2 * FP * A + 2 * VP * 3;
    float *in = (float *)alloca(sizeof(float) * nin);
    float *h1 = (float *)alloca(sizeof(float) * C), *h2 = (float *)alloca(sizeof(float) * C);
    int k = 0;
    for (int c = 0; c < A; ++c) in[k++] = f[c];
    for (int c = 0; c < 3; ++c) in[k++] = d[c];
    for (int c = 0; c < A; ++c) for (int q = 0; q < FP; ++q) in[k++] = sinf(f[c] * (float)(1 << q));
    for (int c = 0; c < A; ++c) for (int q = 0; q < FP; ++q) in[k++] = cosf(f[c] * (float)(1 << q));
    for (int c = 0; c < 3; ++c) for (int q = 0; q < VP; ++q) in[k++] = sinf(d[c] * (float)(1 << q));
    for (int c = 0; c < 3; ++c) for (int q = 0; q < VP; ++q) in[k++] = cosf(d[c] * (float)(1 << q));
    for (int o = 0; o < C; ++o) {
        float acc = 0.0f;
        for (int j = 0; j < nin; ++j) acc = acc + in[j] * s->W1[(size_t)o * nin + j];
        acc = acc + s->b1[o];
        h1[o] = acc > 0.0f ? acc : 0.0f;
    }
    for (int o = 0; o < C; ++o) {
        float acc = 0.0f;
        for (int j = 0; j < C; ++j) acc = acc + h1[j] * s->W2[(size_t)o * C + j];
        acc = acc + s->b2[o];
        h2[o] = acc > 0.0f ? acc : 0.0f;
    }
    for (int o = 0; o < 3; ++o) {
        float acc = 0.0f;
        for (int j = 0; j < C; ++j) acc = acc + h2[j] * s->W3[(size_t)o * C + j];
        acc = acc + s->b3[o];
        rgb[o] = 1.0f / (1.0f + expf(-acc));
    }
    if (in_out) memcpy(in_out, in, sizeof(float) * nin);
}

/* tensorBase.py:50-59 */
float tvr_oracle_alpha_sample(const oracle_scene *s, const float p[3])
{
    float q[3];
    for (int k = 0; k < 3; ++k) {
        float size = s->alpha_aabb[3 + k] - s->alpha_aabb[k];
        float inv = 1.0f / size;
        inv = inv * 2.0f;
        float t = p[k] - s->alpha_aabb[k];
        t = t * inv;
        q[k] = t - 1.0f;
    }
    return trilinear(s->alpha_vol, s->agrid[2], s->agrid[1], s->agrid[0],
                     unnorm(q[0], s->agrid[0]), unnorm(q[1], s->agrid[1]), unnorm(q[2], s->agrid[2]));
}

static inline float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   /* beta 1, threshold 20 */

/* tensorBase.py:476-536 for ONE ray (Appendix A steps 1-17) */
static void render_ray(const oracle_scene *s, const float *ray, int S, int white_bg, const float *jit,
                       float *rgb_out, float *depth_out, const oracle_dump *dp, size_t r,
                       float *z, float *sig, float *wgt)
{
    const float *o = ray, *d = ray + 3;
    const float *lo = s->aabb, *hi = s->aabb + 3;
    /* 1-3  sample_ray tensorBase.py:345-348 */
    float tmax = -INFINITY;
    for (int k = 0; k < 3; ++k) {
        float v = (d[k] == 0.0f) ? 1e-6f : d[k];
        float ra = (hi[k] - o[k]) / v, rb = (lo[k] - o[k]) / v;
        float m = ra < rb ? ra : rb;
        if (m > tmax) tmax = m;
    }
    float tmin = tmax < s->near_ ? s->near_ : (tmax > s->far_ ? s->far_ : tmax);
    if (dp && dp->tmin) dp->tmin[r] = tmin;
    float inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = 2.0f / (hi[k] - lo[k]);           /* :201 */
    float T = 1.0f, acc = 0.0f, C[3] = {0, 0, 0}, dsum = 0.0f;
    float *f = (float *)alloca(sizeof(float) * s->app_dim);
    for (int j = 0; j <= S; ++j) {                                          /* z[S] only feeds dist[S-1] */
        float fj = (float)j;
        if (jit) fj = fj + jit[0];                                           /* :351-353 */
        float st = s->step * fj;                                             /* :354 */
        z[j] = tmin + st;                                                    /* :355 */
    }
    for (int j = 0; j < S; ++j) {
        float p[3], n[3];
        int valid = 1;
        for (int k = 0; k < 3; ++k) {
            float t = d[k] * z[j];
            p[k] = o[k] + t;                                                 /* :357 */
            if (lo[k] > p[k] || p[k] > hi[k]) valid = 0;                     /* :358 */
        }
        int bbox_valid = valid;
        if (valid && s->alpha_vol) valid = tvr_oracle_alpha_sample(s, p) > 0.0f;   /* :491-496 */
        for (int k = 0; k < 3; ++k) {
            float t = p[k] - lo[k];
            t = t * inv[k];
            n[k] = t - 1.0f;                                                 /* :223-224 */
        }
        float sf = 0.0f, sigma = 0.0f;
        if (valid) {
            sf = tvr_oracle_density_feature(s, n);
            sigma = s->act == 0 ? softplus_t(sf + s->density_shift) : (sf > 0.0f ? sf : 0.0f);
        }
        float dist = (j < S - 1) ? (z[j + 1] - z[j]) : 0.0f;                 /* :488 */
        dist = dist * s->distance_scale;                                      /* :511 */
        float alpha = 1.0f - expf(-sigma * dist);                            /* :19 */
        float w = alpha * T;                                                  /* :23 */
        float onem = 1.0f - alpha;
        onem = onem + 1e-10f;
        T = T * onem;                                                         /* :21 */
        sig[j] = sigma; wgt[j] = w;
        int app = w > s->thres;                                              /* :513 */
        float rgb[3] = {0, 0, 0};
        if (app) {
            tvr_oracle_app_feature(s, n, f, NULL);
            tvr_oracle_mlp(s, f, d, rgb, NULL);
        }
        acc = acc + w;                                                        /* :520 */
        for (int c = 0; c < 3; ++c) C[c] = C[c] + w * rgb[c];                /* :521 */
        dsum = dsum + w * z[j];                                               /* :530 */
        if (dp) {
            size_t q = r * (size_t)S + j;
            if (dp->z) dp->z[q] = z[j];
            if (dp->valid) dp->valid[q] = (uint8_t)valid;
            if (dp->bbox_valid) dp->bbox_valid[q] = (uint8_t)bbox_valid;
            if (dp->cell) for (int k = 0; k < 3; ++k) dp->cell[q * 3 + k] = (int32_t)floorf(unnorm(n[k], s->grid[k]));
            if (dp->sf) dp->sf[q] = sf;
            if (dp->sigma) dp->sigma[q] = sigma;
            if (dp->alpha) dp->alpha[q] = alpha;
            if (dp->weight) dp->weight[q] = w;
            if (dp->app) dp->app[q] = (uint8_t)app;
            if (dp->rgb) for (int c = 0; c < 3; ++c) dp->rgb[q * 3 + c] = rgb[c];
        }
    }
    for (int c = 0; c < 3; ++c) {
        float v = C[c];
        if (white_bg) v = v + (1.0f - acc);                                  /* :523-524 */
        rgb_out[c] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);                /* :527 */
    }
    *depth_out = dsum + (1.0f - acc) * ray[5];                               /* :531 (rays[..., -1] = d_z) */
    if (dp && dp->acc) dp->acc[r] = acc;
}

/* tensorf-myc/renderer.py:12-27 — chunking is irrelevant to per-ray results; rays are independent.
 * nthreads > 1 parallelises over rays with OpenMP (same per-ray arithmetic). */
int tvr_oracle_render(const oracle_scene *s, const float *rays, int64_t n, int S, int white_bg,
                      const float *jitter, float *rgb, float *depth, const oracle_dump *dp, int nthreads)
{
    if (!s || !rays || !rgb || !depth || S <= 0) return -1;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
    {
        float *z = (float *)malloc(sizeof(float) * (S + 1));
        float *sig = (float *)malloc(sizeof(float) * S), *wgt = (float *)malloc(sizeof(float) * S);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n; ++r)
            render_ray(s, rays + r * 6, S, white_bg, jitter ? jitter + r : NULL, rgb + r * 3, depth + r, dp, (size_t)r,
                       z, sig, wgt);
        free(z); free(sig); free(wgt);
    }
    return 0;
}

void tvr_oracle_density_features(const oracle_scene *s, const float *xyz_norm, int64_t m, float *out)
{
    for (int64_t i = 0; i < m; ++i) out[i] = tvr_oracle_density_feature(s, xyz_norm + i * 3);
}

void tvr_oracle_app_features(const oracle_scene *s, const float *xyz_norm, int64_t m, float *out)
{
    for (int64_t i = 0; i < m; ++i) tvr_oracle_app_feature(s, xyz_norm + i * 3, out + i * s->app_dim, NULL);
}

void tvr_oracle_alpha_samples(const oracle_scene *s, const float *xyz, int64_t m, float *out)
{
    for (int64_t i = 0; i < m; ++i) out[i] = tvr_oracle_alpha_sample(s, xyz + i * 3);
}
