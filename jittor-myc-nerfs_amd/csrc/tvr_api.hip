// tvr_api.hip — the C-ABI of libtvr.so (include/tvr.h).  Host code only: argument checks, packed-scene layout,
// scratch carving and kernel launches on the caller's stream.  No device allocation, no synchronisation.
#include "tvr_device.h"
#include "tvr_kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

static thread_local char g_err[512] = "";

int tvr_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define fail tvr_set_error

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(TVR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// an output (or gradient input) matrix of `rows` x `cols` fp32 must have been allocated with at least that many bytes: checked on the host,
// before anything is launched (round 2's device fault was a caller buffer 144 - sum(app_n_comp) columns short of the kernel's row width)
static int need_bytes(const char *fn, const char *what, size_t have, int64_t rows, int cols)
{
    const size_t want = (size_t)rows * (size_t)cols * sizeof(float);
    if (have < want)
        return fail(TVR_ERR_SCRATCH, "%s: %s holds %zu B, the kernel writes %lld rows x %d fp32 = %zu B", fn, what, have, (long long)rows, cols, want);
    return TVR_OK;
}
#define NEED(what, have, rows, cols)                                     \
    do {                                                                 \
        int rc_ = need_bytes(__func__, what, (have), (rows), (cols));    \
        if (rc_ != TVR_OK) return rc_;                                   \
    } while (0)

static const int kMatH[3][2] = {{0, 1}, {0, 2}, {1, 2}};
static const int kVecH[3] = {2, 1, 0};

struct PackedLayout {
    size_t dplane[3], dline[3], aplane[3], aline[3];
    size_t aplane16[3], aline16[3];            // fp16 copies of the appearance factors (TVR_ARITH_F16's gather), floats / 2
    size_t mlp_image, basis_frag, b3, w1gen, img16, basg16, refg16, total;
};

static PackedLayout packed_layout(const tvr_scene_desc &d)
{
    PackedLayout L;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * sizeof(float), 256); return o; };
    for (int i = 0; i < 3; ++i) {
        const size_t W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        L.dplane[i] = take((H + 1) * (W + 1) * TVR_CD);
        L.dline[i] = take((Ln + 1) * TVR_CD);
        L.aplane[i] = take((H + 1) * (W + 1) * TVR_CA);
        L.aline[i] = take((Ln + 1) * TVR_CA);
    }
    L.mlp_image = take(TVR_MLP_IMAGE_BYTES_REF / 4);
    L.basis_frag = take(TVR_BASIS_FRAG_BYTES / 4);
    L.b3 = take(16);
    // scenes with more than two encoding frequencies (TensorBase's own default is 6 / 6: 390 MLP inputs): layer 1's fragment image, 26 k-steps, streamed through LDS
    L.w1gen = (d.view_pe > 2 || d.fea_pe > 2) ? take(TVR_W1GEN_BYTES / 4) : 0;
    // the 16x16x32 render kernel's fragment images (TensorVMSplit, at most two encoding frequencies: the shape every shipped config has)
    // (round 6: REFTensoRF scenes too — check_desc admits them at the built-for shape only — with the heads' fragments beside them)
    const bool has16 = !(d.view_pe > 2 || d.fea_pe > 2);
    L.img16 = has16 ? take(TVR16_IMAGE_BYTES / 4) : 0;
    L.basg16 = has16 ? take(TVR16_BASG_BYTES / 4) : 0;
    L.refg16 = (has16 && d.variant == 1) ? take(TVR16_REFG_BYTES / 4) : 0;
    for (int i = 0; i < 3; ++i) {
        const size_t W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        L.aplane16[i] = take((H + 1) * (W + 1) * TVR_CA / 2);
        L.aline16[i] = take((Ln + 1) * TVR_CA / 2);
    }
    L.total = off;
    return L;
}

struct tvr_scene {
    tvr_scene_desc desc;
    PackedLayout lay;
    char *packed;
    SceneDev dev;
    bool params_set;
    bool h16_stale;            // the fp16 copies of the appearance factors are older than the fp32 images
    int arith_req;             // what tvr_scene_set_arith asked for; dev.arith is what the kernels RUN: arith_req once tvr_scene_validate_arith has measured it inside
    int arith_valid;           // its tolerance for the parameters as packed (0 = nothing validated), TVR_ARITH_F32 until then
    // piecewise rendering (round 6, tvr_scene_set_render_pieces): a tvr_render(_z) call of at least 2 x piece_rays rays goes out as pieces of ~piece_rays consecutive rays,
    // alternately on two library-owned streams that fork from and join the caller's stream by events
    int piece_rays;
    hipStream_t side[2];
    hipEvent_t ev_fork, ev_join[2];
    bool side_ready;
};

struct tvr_profile {
    std::vector<hipEvent_t> ev;   // 4 per recorded launch set (one per call; one per PIECE of a call rendered piecewise)
    int max_calls, n_calls;       // n_calls counts calls, n_sets launch sets
    int n_sets;
};

// the default piece: measured on the 800x800 bench frame (scripts/split_frame_experiment.py, profiles/r06_split_frame.txt); TVR_PIECE_RAYS overrides it for experiments
#ifndef TVR_DEFAULT_PIECE_RAYS
#define TVR_DEFAULT_PIECE_RAYS 30720
#endif
static int default_piece_rays()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("TVR_PIECE_RAYS");
        v = e ? atoi(e) : TVR_DEFAULT_PIECE_RAYS;
        if (v < 0) v = 0;
    }
    return v;
}

// the appearance factors as fp16, from the packed fp32 images (the one-product arithmetic gathers these: half the bytes through L1)
static int refresh_h16(tvr_scene *s, hipStream_t stream)
{
    const tvr_scene_desc &d = s->desc;
    for (int i = 0; i < 3; ++i) {
        const long long W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        HIP_TRY(launch_f32_to_f16((const float *)(s->packed + s->lay.aplane[i]), s->packed + s->lay.aplane16[i], (H + 1) * (W + 1) * TVR_CA, stream));
        HIP_TRY(launch_f32_to_f16((const float *)(s->packed + s->lay.aline[i]), s->packed + s->lay.aline16[i], (Ln + 1) * TVR_CA, stream));
    }
    s->h16_stale = false;
    return TVR_OK;
}

static int check_desc(const tvr_scene_desc *d)
{
    if (!d) return fail(TVR_ERR_INVALID, "desc is NULL");
    for (int i = 0; i < 3; ++i) {
        if (d->grid[i] < 2 || d->grid[i] > 4096) return fail(TVR_ERR_INVALID, "grid[%d]=%d out of [2,4096]", i, d->grid[i]);
        // fewer components / narrower layers than the kernels are built for are packed with zero padding (exact); more are not supported
        if (d->density_n_comp[i] < 1 || d->density_n_comp[i] > TVR_CD)
            return fail(TVR_ERR_UNSUPPORTED, "density_n_comp[%d]=%d; this build supports 1..%d", i, d->density_n_comp[i], TVR_CD);
        if (d->app_n_comp[i] < 1 || d->app_n_comp[i] > TVR_CA)
            return fail(TVR_ERR_UNSUPPORTED, "appearance_n_comp[%d]=%d; this build supports 1..%d", i, d->app_n_comp[i], TVR_CA);
        if (!(d->aabb[3 + i] > d->aabb[i])) return fail(TVR_ERR_INVALID, "aabb hi <= lo on axis %d", i);
    }
    if (d->app_dim != TVR_APPDIM || d->featureC < 1 || d->featureC > TVR_FEATC || d->view_pe < 0 || d->view_pe > TVR_GEN_PE || d->fea_pe < 0 || d->fea_pe > TVR_GEN_PE)
        return fail(TVR_ERR_UNSUPPORTED, "MLP_Fea shape app_dim=%d featureC=%d view_pe=%d fea_pe=%d; this build supports app_dim 27, featureC 1..128, view_pe / fea_pe 0..%d",
                    d->app_dim, d->featureC, d->view_pe, d->fea_pe, TVR_GEN_PE);
    if (d->variant == 1) {
        bool std_shape = d->featureC == TVR_FEATC && d->view_pe == 2 && d->fea_pe == 2;
        for (int i = 0; i < 3; ++i) std_shape = std_shape && d->density_n_comp[i] == TVR_CD && d->app_n_comp[i] == TVR_CA;
        if (!std_shape) return fail(TVR_ERR_UNSUPPORTED, "REFTensoRF scenes are supported at 16 / 48 components, featureC 128, view_pe = fea_pe = 2 only");
    }
    if (d->fea2dense_act != 0 && d->fea2dense_act != 1) return fail(TVR_ERR_INVALID, "fea2dense_act must be 0 or 1");
    if (d->variant != 0 && d->variant != 1) return fail(TVR_ERR_INVALID, "variant must be 0 (TensorVMSplit) or 1 (REFTensoRF)");
    if (!(d->step_size > 0.0f)) return fail(TVR_ERR_INVALID, "step_size must be > 0");
    return TVR_OK;
}

extern "C" {

int tvr_version(void) { return TVR_VERSION; }
const char *tvr_last_error(void) { return g_err; }

size_t tvr_scene_packed_bytes(const tvr_scene_desc *desc)
{
    if (check_desc(desc) != TVR_OK) return 0;
    return packed_layout(*desc).total;
}

int tvr_scene_create(const tvr_scene_desc *desc, void *packed_dev, size_t packed_bytes, tvr_scene **out)
{
    if (!out) return fail(TVR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = check_desc(desc);
    if (rc != TVR_OK) return rc;
    PackedLayout L = packed_layout(*desc);
    if (!packed_dev || packed_bytes < L.total) return fail(TVR_ERR_SCRATCH, "packed buffer %zu B < required %zu B", packed_bytes, L.total);
    if ((uintptr_t)packed_dev % 256) return fail(TVR_ERR_SCRATCH, "packed buffer must be 256-byte aligned");
    tvr_scene *s = new (std::nothrow) tvr_scene;
    if (!s) return fail(TVR_ERR_INVALID, "out of host memory");
    s->desc = *desc;
    s->lay = L;
    s->packed = (char *)packed_dev;
    s->params_set = false;
    s->h16_stale = true;
    s->arith_req = TVR_ARITH_F32;
    s->arith_valid = 0;
    s->piece_rays = default_piece_rays();
    s->side[0] = s->side[1] = nullptr;
    s->ev_fork = s->ev_join[0] = s->ev_join[1] = nullptr;
    s->side_ready = false;
    SceneDev &v = s->dev;
    memset(&v, 0, sizeof(v));
    for (int k = 0; k < 3; ++k) {
        v.lo[k] = desc->aabb[k];
        v.hi[k] = desc->aabb[3 + k];
        v.inv[k] = desc->inv_aabb_size[k];
        v.grid[k] = desc->grid[k];
        v.gm1[k] = (float)(desc->grid[k] - 1);
        v.dplane[k] = (const float4 *)(s->packed + L.dplane[k]);
        v.dline[k] = (const float4 *)(s->packed + L.dline[k]);
        v.aplane[k] = (const float4 *)(s->packed + L.aplane[k]);
        v.aline[k] = (const float4 *)(s->packed + L.aline[k]);
        v.aplane16[k] = (const uint4 *)(s->packed + L.aplane16[k]);
        v.aline16[k] = (const uint4 *)(s->packed + L.aline16[k]);
    }
    v.mlp_image = s->packed + L.mlp_image;
    v.basis_frag = s->packed + L.basis_frag;
    v.b3 = (const float *)(s->packed + L.b3);
    v.gen = (desc->view_pe > 2 || desc->fea_pe > 2) ? 1 : 0;
    v.w1gen = v.gen ? (const void *)(s->packed + L.w1gen) : nullptr;
    v.img16 = !v.gen ? (const void *)(s->packed + L.img16) : nullptr;
    v.basg16 = v.img16 ? (const void *)(s->packed + L.basg16) : nullptr;
    v.refg16 = (v.img16 && desc->variant == 1) ? (const void *)(s->packed + L.refg16) : nullptr;
    v.near_ = desc->near_;
    v.far_ = desc->far_;
    v.step = desc->step_size;
    v.shift = desc->density_shift;
    v.scale = desc->distance_scale;
    v.thres = desc->weight_thres;
    v.act = desc->fea2dense_act;
    v.variant = desc->variant;
    v.range_check = 1;
    v.arith = TVR_ARITH_F32;
    v.avol = nullptr;
    *out = s;
    return TVR_OK;
}

int tvr_scene_update(tvr_scene *s, const tvr_scene_params *p, void *stream_)
{
    if (!s || !p) return fail(TVR_ERR_INVALID, "scene/params is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    const tvr_scene_desc &d = s->desc;
    for (int i = 0; i < 3; ++i) {
        if (!p->density_plane[i] || !p->density_line[i] || !p->app_plane[i] || !p->app_line[i])
            return fail(TVR_ERR_INVALID, "plane/line pointer %d is NULL", i);
        const int W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        HIP_TRY(launch_pack_plane(p->density_plane[i], (float *)(s->packed + s->lay.dplane[i]), d.density_n_comp[i], TVR_CD, H, W, stream));
        HIP_TRY(launch_pack_plane(p->density_line[i], (float *)(s->packed + s->lay.dline[i]), d.density_n_comp[i], TVR_CD, Ln, 1, stream));
        HIP_TRY(launch_pack_plane(p->app_plane[i], (float *)(s->packed + s->lay.aplane[i]), d.app_n_comp[i], TVR_CA, H, W, stream));
        HIP_TRY(launch_pack_plane(p->app_line[i], (float *)(s->packed + s->lay.aline[i]), d.app_n_comp[i], TVR_CA, Ln, 1, stream));
    }
    if (!p->basis_mat || !p->W1 || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3) return fail(TVR_ERR_INVALID, "MLP pointer is NULL");
    char *img = s->packed + s->lay.mlp_image;
    MlpShape sh;
    sh.featureC = d.featureC; sh.fea_pe = d.fea_pe; sh.view_pe = d.view_pe;
    sh.n_in = TVR_APPDIM + 3 + 2 * TVR_APPDIM * d.fea_pe + 6 * d.view_pe + (d.variant == 1 ? 1 : 0);
    sh.k_app = 0;
    for (int i = 0; i < 3; ++i) { sh.app_n_comp[i] = d.app_n_comp[i]; sh.app_off[i] = sh.k_app; sh.k_app += d.app_n_comp[i]; }
    if (s->dev.gen) {       // more than two frequencies: layer 1's image is the 26-k-step general one in global memory; the LDS image's W1 region is two staging slots
        HIP_TRY(launch_zero_f32((float *)(img + TVR_IMG_W1H), (TVR_IMG_W2H - TVR_IMG_W1H) / 4, stream));
        HIP_TRY(launch_pack_mlp(p->W1, p->b1, s->packed + s->lay.w1gen, nullptr, 5, sh, stream));
    } else {
        HIP_TRY(launch_pack_mlp(p->W1, p->b1, img + TVR_IMG_W1H, img + TVR_IMG_W1L, d.variant == 1 ? 4 : 0, sh, stream));
    }
    HIP_TRY(launch_pack_mlp(p->W2, nullptr, img + TVR_IMG_W2H, img + TVR_IMG_W2L, 1, sh, stream));
    // (clears and small copies as KERNELS: tvr_scene_update runs inside a captured training step, and memset / memcpy nodes of a graph replayed back to back
    //  were observed to run ahead of the previous replay's kernels — tvr_step.hip)
    HIP_TRY(launch_zero_f32((float *)(img + TVR_IMG_B1), (2 * 512 + 4 * TVR_IMG_W3_ROW) / 4, stream));       // b3 slot, b2, W3 + zero row: hidden units >= featureC stay zero
    HIP_TRY(launch_copy_f32((float *)(img + TVR_IMG_B3), p->b3, 3, stream));
    HIP_TRY(launch_copy_f32((float *)(img + TVR_IMG_B2), p->b2, d.featureC, stream));
    HIP_TRY(launch_pack_mlp(p->basis_mat, nullptr, img + TVR_IMG_BASH, s->packed + s->lay.basis_frag, 2, sh, stream));
    for (int r = 0; r < 3; ++r) HIP_TRY(launch_copy_f32((float *)(img + TVR_IMG_W3 + r * TVR_IMG_W3_ROW), p->W3 + (size_t)r * d.featureC, d.featureC, stream));
    HIP_TRY(launch_copy_f32((float *)(s->packed + s->lay.b3), p->b3, 3, stream));
    HIP_TRY(launch_zero_f32((float *)(img + TVR_MLP_IMAGE_BYTES), (TVR_MLP_IMAGE_BYTES_REF - TVR_MLP_IMAGE_BYTES) / 4, stream));
    if (d.variant == 1) {
        for (int i = 0; i < 4; ++i)
            if (!p->ref_W[i] || !p->ref_b[i]) return fail(TVR_ERR_INVALID, "REFTensoRF linear %d (normal, diffuse, specular, rho) is NULL", i);
    }
    if (s->dev.img16)
        HIP_TRY(launch_pack16(p->W1, p->b1, p->W2, p->b2, p->W3, p->b3, p->basis_mat, s->packed + s->lay.img16, s->packed + s->lay.basg16, sh, stream,
                              d.variant == 1 ? p->ref_W : nullptr, d.variant == 1 ? p->ref_b : nullptr, d.variant == 1 ? s->packed + s->lay.refg16 : nullptr));
    if (d.variant == 1) {
        HIP_TRY(launch_pack_ref(p->ref_W, p->ref_b, img + TVR_IMG_REFW, (float *)(img + TVR_IMG_REFB), stream));
    }
    s->params_set = true;
    s->h16_stale = true;                  // (the first render in TVR_ARITH_F16 converts the fp16 copies: nothing is converted for steps that never read them)
    s->arith_valid = 0;                   // new parameters: a reduced arithmetic has to be measured again before it runs (tvr_scene_validate_arith)
    s->dev.arith = TVR_ARITH_F32;
    return TVR_OK;
}

size_t tvr_alpha_bits_bytes(const int32_t ag[3])
{
    if (!ag || ag[0] < 1 || ag[1] < 1 || ag[2] < 1) return 0;
    return (size_t)(((long long)ag[0] * ag[1] * ag[2] + 31) / 32) * 4;
}

int tvr_scene_set_alpha(tvr_scene *s, const float *vol, const int32_t ag[3], const float aabb[6], const float inv[3], void *bits, size_t bits_bytes,
                        void *stream)
{
    if (!s) return fail(TVR_ERR_INVALID, "scene is NULL");
    if (!vol) { s->dev.avol = nullptr; s->dev.abits = nullptr; return TVR_OK; }
    if (!ag || !aabb || !inv) return fail(TVR_ERR_INVALID, "alpha grid/aabb/inv is NULL");
    for (int k = 0; k < 3; ++k) {
        if (ag[k] < 1) return fail(TVR_ERR_INVALID, "alpha grid[%d]=%d", k, ag[k]);
        s->dev.ag[k] = ag[k];
        s->dev.alo[k] = aabb[k];
        s->dev.ainv[k] = inv[k];
        s->dev.agm1[k] = (float)(ag[k] - 1);
    }
    s->dev.avol = vol;
    s->dev.abits = nullptr;
    if (bits) {
        if (bits_bytes < tvr_alpha_bits_bytes(ag) || (uintptr_t)bits % 4) return fail(TVR_ERR_SCRATCH, "alpha bit buffer too small or misaligned");
        HIP_TRY(launch_alpha_bits(vol, (long long)ag[0] * ag[1] * ag[2], (unsigned *)bits, (hipStream_t)stream));
        s->dev.abits = (const unsigned *)bits;
    }
    return TVR_OK;
}

int tvr_scene_set_range_check(tvr_scene *s, int32_t on)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_set_range_check: scene is NULL");
    s->dev.range_check = on ? 1 : 0;
    return TVR_OK;
}

int tvr_scene_touch(tvr_scene *s)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_touch: scene is NULL");
    s->h16_stale = true;         // the next TVR_ARITH_F16 render converts the fp32 images first
    s->arith_valid = 0;          // and a reduced arithmetic has to be validated again
    s->dev.arith = TVR_ARITH_F32;
    return TVR_OK;
}

int tvr_scene_set_arith(tvr_scene *s, int32_t mode)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_set_arith: scene is NULL");
    if (mode != TVR_ARITH_F32 && mode != TVR_ARITH_F16ACT && mode != TVR_ARITH_F16) return fail(TVR_ERR_INVALID, "tvr_scene_set_arith: mode %d is none of TVR_ARITH_*", (int)mode);
    s->arith_req = mode;
    s->dev.arith = (mode == TVR_ARITH_F32 || s->arith_valid == mode) ? mode : TVR_ARITH_F32;
    if (s->dev.arith != mode)
        tvr_set_error(TVR_OK, "tvr_scene_set_arith: mode %d is requested but NOT in effect — the kernels compute in TVR_ARITH_F32 until tvr_scene_validate_arith has measured "
                              "the mode inside its tolerance for the current parameters", (int)mode);
    return TVR_OK;
}

int tvr_scene_get_arith(const tvr_scene *s)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_get_arith: scene is NULL");
    return s->dev.arith;
}

int tvr_scene_get_arith_requested(const tvr_scene *s)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_get_arith_requested: scene is NULL");
    return s->arith_req;
}

int tvr_scene_set_render_pieces(tvr_scene *s, int32_t piece_rays)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_set_render_pieces: scene is NULL");
    if (piece_rays < 0) piece_rays = default_piece_rays();
    if (piece_rays != 0 && piece_rays < 16) return fail(TVR_ERR_INVALID, "tvr_scene_set_render_pieces: piece_rays %d (0 = off, < 0 = the library's default, else at least 16)", (int)piece_rays);
    s->piece_rays = piece_rays;
    return TVR_OK;
}

int tvr_scene_get_render_pieces(const tvr_scene *s)
{
    if (!s) return fail(TVR_ERR_INVALID, "tvr_scene_get_render_pieces: scene is NULL");
    return s->piece_rays;
}

int tvr_scene_destroy(tvr_scene *s)
{
    if (s && s->side_ready) {          // (the caller has synchronised whatever it enqueued on this scene: the side streams only ever carry work joined back to a caller's stream)
        for (int i = 0; i < 2; ++i) {
            if (s->side[i]) (void)hipStreamDestroy(s->side[i]);
            if (s->ev_join[i]) (void)hipEventDestroy(s->ev_join[i]);
        }
        if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    }
    delete s;
    return TVR_OK;
}

// scratch carving shared by the size query and tvr_render
#define TVR_RAY_ORDER_MAX_RAYS 65536      // march_forward_impl: batches up to this many rays get a ray-ordered queue
struct ScratchLayout { size_t counter, ray_off, ray_cnt, acc, q_pos, q_out, q_ray, q_j, ray_new, total; };
static ScratchLayout scratch_layout(int64_t n_rays, int32_t S)
{
    ScratchLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t cap = (size_t)n_rays * (size_t)S;
    L.counter = take(256);
    L.ray_off = take(n_rays * 4);
    L.ray_cnt = take(n_rays * 4);
    L.acc = take(n_rays * 4);
    L.q_pos = take(cap * 16);
    L.q_out = take(cap * 16);
    L.q_ray = take(cap * 4);
    L.q_j = take(cap * 4);
    L.ray_new = take(n_rays * 4);          // the training queue's ray-ordered offsets (launch_queue_ray_order); behind everything the public layout names
    L.total = off;
    return L;
}

// Piecewise rendering: a call of n rays goes out as K = round(n / piece_rays) pieces of ceil(n / K) rays rounded up to 512 (the last one shorter), if that makes at least
// TVR_MIN_PIECES.  Measured on one box (bench.py --emulate-world N --pieces P, profiles/r06_split_frame.txt): the 640 000-ray frame -3.6 % .. -4.6 %, a 320 000-ray share
// -4.6 %, a 160 000-ray share +-0.5 %, an 80 000-ray share +0.9 % (two or three pieces do not settle into the overlap that pays): small calls stay one launch set.
#ifndef TVR_MIN_PIECES
#define TVR_MIN_PIECES 6
#endif
struct PiecePlan { int K; int64_t rays; };
static PiecePlan piece_plan(const tvr_scene *s, int64_t n_rays)
{
    PiecePlan P = {1, n_rays};
    const int64_t pr = s ? s->piece_rays : default_piece_rays();
    if (pr <= 0 || n_rays < TVR_MIN_PIECES * pr) return P;
    const int64_t K = (n_rays + pr / 2) / pr, q = pr >= 512 ? 512 : 16;          // (pieces below 512 rays: tests on tiny fixtures)
    P.rays = ((n_rays + K - 1) / K + q - 1) / q * q;
    P.K = (int)((n_rays + P.rays - 1) / P.rays);
    if (P.K < 2) { P.K = 1; P.rays = n_rays; }
    return P;
}

size_t tvr_render_scratch_bytes(const tvr_scene *s, int64_t n_rays, int32_t n_samples)
{
    if (n_rays <= 0 || n_samples <= 0) return 256;
    const size_t whole = scratch_layout(n_rays, n_samples).total;
    const PiecePlan P = piece_plan(s, n_rays);             // two pieces in flight, each with a scratch of its own carved from the caller's buffer
    const size_t two = P.K > 1 ? 2 * scratch_layout(P.rays, n_samples).total : 0;
    return whole > two ? whole : two;
}

// What a call WITHOUT `dense` needs: a call rendered in pieces works in two pieces' scratch (1.3 GB instead of 13 GB for the 800x800 x 512 frame: the queue is sized for the
// worst case, every sample of every ray shaded); a `dense` call is one launch set over the whole batch and needs tvr_render_scratch_bytes().
size_t tvr_render_scratch_bytes_min(const tvr_scene *s, int64_t n_rays, int32_t n_samples)
{
    if (n_rays <= 0 || n_samples <= 0) return 256;
    const PiecePlan P = piece_plan(s, n_rays);
    return P.K > 1 ? 2 * scratch_layout(P.rays, n_samples).total : scratch_layout(n_rays, n_samples).total;
}

}  // extern "C"

// one launch set (header clear, march, shade, composite) on `stream`; ev: four events or nullptr
static int render_one(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, int32_t white_bg, const MarchSampling &sm, float eps_T, float *rgb_out, float *depth_out,
                      float *lam6_out, char *b, const ScratchLayout &L, const tvr_dense_out *dense, uint64_t *stats, hipEvent_t *ev, hipStream_t stream)
{
    MarchOut mo;
    mo.counter = (unsigned *)(b + L.counter);
    mo.ray_off = (unsigned *)(b + L.ray_off);
    mo.ray_cnt = (unsigned *)(b + L.ray_cnt);
    mo.acc = (float *)(b + L.acc);
    mo.depth = depth_out;
    mo.q_pos = (float4 *)(b + L.q_pos);
    mo.q_out = (float4 *)(b + L.q_out);
    mo.q_ray = (unsigned *)(b + L.q_ray);
    mo.q_j = (dense && dense->rgb) ? (unsigned *)(b + L.q_j) : nullptr;
    mo.stats = (unsigned long long *)stats;
    mo.lam6 = lam6_out;
    HIP_TRY(launch_zero_header(mo.counter, stream));                   // [0] queue length, [1] the march's tile counter, [2] its fault flag, [3] unused here
    if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
    HIP_TRY(launch_march(s->dev, rays, (int)n_rays, S, sm, eps_T, mo, dense, stream));
    if (ev) HIP_TRY(hipEventRecord(ev[1], stream));
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.counter = mo.counter;
    sa.q_pos = mo.q_pos;
    sa.q_out = mo.q_out;
    sa.q_ray = mo.q_ray;
    sa.rays = rays;
    sa.stats = mo.stats;
    HIP_TRY(launch_shade(s->dev, SH_SRC_QUEUE, SH_DST_QUEUE, sa, stream));
    if (ev) HIP_TRY(hipEventRecord(ev[2], stream));
    if (dense && dense->rgb) {
        HIP_TRY(hipMemsetAsync(dense->rgb, 0, (size_t)n_rays * S * 3 * sizeof(float), stream));
        HIP_TRY(launch_scatter_rgb(mo, S, dense->rgb, stream));
    }
    HIP_TRY(launch_composite(mo, (int)n_rays, white_bg, rgb_out, stream));
    if (ev) HIP_TRY(hipEventRecord(ev[3], stream));
    return TVR_OK;
}

static int ensure_side_streams(tvr_scene *s)
{
    if (s->side_ready) return TVR_OK;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&s->side[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_join[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    s->side_ready = true;
    return TVR_OK;
}

static hipEvent_t *profile_slot(tvr_profile *prof);

static int render_impl(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, int32_t white_bg, const MarchSampling &sm, float eps_T,
                       float *rgb_out, float *depth_out, float *lam6_out, void *scratch, size_t scratch_bytes, const tvr_dense_out *dense,
                       uint64_t *stats, tvr_profile *prof, void *stream_)
{
    if (!s || !s->params_set) return fail(TVR_ERR_INVALID, "scene is NULL or tvr_scene_update has not run");
    if (n_rays == 0) return TVR_OK;
    if (!rays || !rgb_out || !depth_out || n_rays < 0) return fail(TVR_ERR_INVALID, "rays/rgb_out/depth_out NULL or n_rays < 0");
    if (S <= 0 || S > 4096) return fail(TVR_ERR_INVALID, "n_samples=%d out of [1,4096]", S);
    if ((size_t)n_rays * (size_t)S >= (1ull << 32)) return fail(TVR_ERR_INVALID, "n_rays*n_samples must be < 2^32 per call (chunk the rays)");
    if (!(eps_T >= 0.0f) || eps_T > s->desc.weight_thres)
        return fail(TVR_ERR_INVALID, "eps_T=%g must be in [0, weight_thres=%g] so that no appearance sample is skipped", eps_T, s->desc.weight_thres);
    const size_t need = dense ? scratch_layout(n_rays, S).total : tvr_render_scratch_bytes_min(s, n_rays, S);
    if (!scratch || scratch_bytes < need) return fail(TVR_ERR_SCRATCH, "scratch %zu B < required %zu B", scratch_bytes, need);
    if ((uintptr_t)scratch % 256) return fail(TVR_ERR_SCRATCH, "scratch must be 256-byte aligned");
    if (prof && prof->n_calls >= prof->max_calls) return fail(TVR_ERR_INVALID, "profile is full (%d calls)", prof->max_calls);
    hipStream_t stream = (hipStream_t)stream_;
    if (s->dev.arith == TVR_ARITH_F16 && !s->dev.gen && s->h16_stale) {        // the mode was set after the last tvr_scene_update
        int rc16 = refresh_h16(s, stream);
        if (rc16 != TVR_OK) return rc16;
    }
    // the per-sample outputs of `dense` are a debugging / parity surface ([n,S,*] arrays): such calls stay one launch set
    const PiecePlan P = dense ? PiecePlan{1, n_rays} : piece_plan(s, n_rays);
    if (P.K == 1) {
        hipEvent_t *ev = nullptr;
        if (prof && !(ev = profile_slot(prof))) return fail(TVR_ERR_HIP, "tvr_profile: hipEventCreate failed");
        int rc = render_one(s, rays, n_rays, S, white_bg, sm, eps_T, rgb_out, depth_out, lam6_out, (char *)scratch, scratch_layout(n_rays, S), dense, stats, ev, stream);
        if (rc != TVR_OK) return rc;
        if (prof) prof->n_calls++;
        return TVR_OK;
    }
    // Piecewise (round 6): the overlap measured ACROSS frames with two frames in flight (round 5, render.FrameStream: 18.8 vs 19.7 ms) inside ONE call.  Piece k runs on
    // library-owned stream k & 1 in that stream's half of the caller's scratch; the two streams fork from the caller's stream by an event and are joined back into it by
    // two more, so the caller sees what it saw before: everything ordered on its stream.  The persistent kernels of one piece take the CUs the other piece's kernels leave
    // as they drain, and a march (L1-bound, 2.1 GHz) beside a shade kernel (matrix-bound, 1.75 GHz) shares the chip's power budget better than either alone.  Every ray's
    // result is independent of the batch it arrives in (bit for bit: tests/test_gpu_parity.py), so the pixels are those of the one-piece call.
    int rc = ensure_side_streams(s);
    if (rc != TVR_OK) return rc;
    const ScratchLayout Lp = scratch_layout(P.rays, S);
    HIP_TRY(hipEventRecord(s->ev_fork, stream));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamWaitEvent(s->side[i], s->ev_fork, 0));
    // (equal pieces: a half- or third-size first piece on stream 1 — a stagger from the start — measured 18.71 - 18.76 ms against 18.64 - 18.67, profiles/r06_split_frame.txt)
    int64_t a = 0;
    for (int k = 0; a < n_rays && rc == TVR_OK; ++k) {
        const int64_t m = (a + P.rays <= n_rays) ? P.rays : n_rays - a;
        const MarchSampling smk = {sm.jitter ? sm.jitter + a : nullptr, sm.zv ? sm.zv + (size_t)a * S : nullptr};
        hipEvent_t *ev = nullptr;
        if (prof && !(ev = profile_slot(prof))) { rc = fail(TVR_ERR_HIP, "tvr_profile: hipEventCreate failed"); break; }
        rc = render_one(s, rays + 6 * a, m, S, white_bg, smk, eps_T, rgb_out + 3 * a, depth_out + a, lam6_out ? lam6_out + a : nullptr,
                        (char *)scratch + (size_t)(k & 1) * Lp.total, Lp, nullptr, stats, ev, s->side[k & 1]);
        a += m;
    }
    // the join is enqueued whatever happened above: whatever DID go out on the two streams is ordered in front of the caller's next work (an error leaves the outputs
    // undefined, never a side stream still writing them behind the caller's back)
    for (int i = 0; i < 2; ++i) {
        const hipError_t e1 = hipEventRecord(s->ev_join[i], s->side[i]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(stream, s->ev_join[i], 0) : e1;
        if (e2 != hipSuccess && rc == TVR_OK) rc = fail(TVR_ERR_HIP, "tvr_render: joining the piece streams: %s", hipGetErrorString(e2));
    }
    if (rc != TVR_OK) return rc;
    if (prof) prof->n_calls++;
    return TVR_OK;
}

extern "C" {

int tvr_render(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, int32_t white_bg, const float *jitter, float eps_T,
               float *rgb_out, float *depth_out, void *scratch, size_t scratch_bytes, const tvr_dense_out *dense,
               uint64_t *stats, tvr_profile *prof, void *stream)
{
    const MarchSampling sm = {jitter, nullptr};
    return render_impl(s, rays, n_rays, S, white_bg, sm, eps_T, rgb_out, depth_out, nullptr, scratch, scratch_bytes, dense, stats, prof, stream);
}

int tvr_render_z(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, int32_t white_bg, const float *z_vals, float eps_T,
                 float *rgb_out, float *depth_out, float *t_last_tiny_out, void *scratch, size_t scratch_bytes, const tvr_dense_out *dense,
                 uint64_t *stats, void *stream)
{
    if (!z_vals) return fail(TVR_ERR_INVALID, "z_vals is NULL");
    const MarchSampling sm = {nullptr, z_vals};
    return render_impl(s, rays, n_rays, S, white_bg, sm, eps_T, rgb_out, depth_out, t_last_tiny_out, scratch, scratch_bytes, dense, stats, nullptr, stream);
}

}  // extern "C"

static int scene_ready(const tvr_scene *s)
{
    if (!s || !s->params_set) return fail(TVR_ERR_INVALID, "scene is NULL or tvr_scene_update has not run");
    return TVR_OK;
}

static MarchOut carve_scratch(char *b, const ScratchLayout &L, float *depth_out)
{
    MarchOut mo;
    mo.counter = (unsigned *)(b + L.counter);
    mo.ray_off = (unsigned *)(b + L.ray_off);
    mo.ray_cnt = (unsigned *)(b + L.ray_cnt);
    mo.acc = (float *)(b + L.acc);
    mo.depth = depth_out;
    mo.q_pos = (float4 *)(b + L.q_pos);
    mo.q_out = (float4 *)(b + L.q_out);
    mo.q_ray = (unsigned *)(b + L.q_ray);
    mo.q_j = nullptr;
    mo.stats = nullptr;
    mo.lam6 = nullptr;
    return mo;
}

extern "C" {

int tvr_scratch_describe(int64_t n_rays, int32_t n_samples, tvr_scratch_layout *out)
{
    if (!out || n_rays <= 0 || n_samples <= 0) return fail(TVR_ERR_INVALID, "bad arguments");
    ScratchLayout L = scratch_layout(n_rays, n_samples);
    out->counter = L.counter; out->ray_off = L.ray_off; out->ray_cnt = L.ray_cnt; out->acc = L.acc;
    out->q_pos = L.q_pos; out->q_out = L.q_out; out->q_ray = L.q_ray; out->q_j = L.q_j; out->total = L.total;
    return TVR_OK;
}

}  // extern "C"

static int march_forward_impl(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const MarchSampling &sm, float eps_T, float *depth_out,
                              float *lam6_out, void *scratch, size_t scratch_bytes, void *stream_)
{
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (!rays || !depth_out || n_rays <= 0) return fail(TVR_ERR_INVALID, "rays/depth_out NULL or n_rays <= 0");
    if (S <= 0 || S > 4096) return fail(TVR_ERR_INVALID, "n_samples=%d out of [1,4096]", S);
    if ((size_t)n_rays * (size_t)S >= (1ull << 32)) return fail(TVR_ERR_INVALID, "n_rays*n_samples must be < 2^32 per call");
    if (!(eps_T >= 0.0f) || eps_T > s->desc.weight_thres) return fail(TVR_ERR_INVALID, "eps_T=%g must be in [0, weight_thres]", eps_T);
    ScratchLayout L = scratch_layout(n_rays, S);
    if (!scratch || scratch_bytes < L.total || (uintptr_t)scratch % 256) return fail(TVR_ERR_SCRATCH, "scratch too small (%zu < %zu) or misaligned", scratch_bytes, L.total);
    hipStream_t stream = (hipStream_t)stream_;
    MarchOut mo = carve_scratch((char *)scratch, L, depth_out);
    mo.lam6 = lam6_out;
    HIP_TRY(launch_zero_header(mo.counter, stream));                   // [0] queue length, [1] the march's tile counter, [2] its fault flag, [3] workspace overflow
    HIP_TRY(launch_march(s->dev, rays, (int)n_rays, S, sm, eps_T, mo, nullptr, stream));
    // a training queue is put into ray order (the q_out / q_j regions are unused on this path): the weight gradients then do not depend on the order the
    // march kernel's waves finished in.  Small batches only — the training path's own (train.py draws 4096 rays); a caller that marches a whole frame through
    // this entry point keeps the kernel's order and rounding-level reproducibility.
    if (n_rays <= TVR_RAY_ORDER_MAX_RAYS)
        HIP_TRY(launch_queue_ray_order(mo, (unsigned *)((char *)scratch + L.ray_new), mo.q_out, (unsigned *)((char *)scratch + L.q_j), (int)n_rays, stream));
    return TVR_OK;
}

extern "C" {

// max |a - b| over n floats -> out[0] (as the bits of a non-negative float: integer max); a NaN on either side counts as +inf
__global__ __launch_bounds__(256) void maxdiff_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n, unsigned *__restrict__ out)
{
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d == d) ? fmaxf(m, d) : INFINITY;
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

int tvr_scene_validate_arith(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, int32_t white_bg, float eps_T, float tol, void *scratch, size_t scratch_bytes,
                             float *work, size_t work_bytes, float *max_diff_out, int64_t *probe_app_samples_out, void *stream_)
{
    if (!s || !s->params_set) return fail(TVR_ERR_INVALID, "tvr_scene_validate_arith: scene is NULL or tvr_scene_update has not run");
    if (!max_diff_out) return fail(TVR_ERR_INVALID, "tvr_scene_validate_arith: max_diff_out is NULL");
    *max_diff_out = 0.0f;
    if (probe_app_samples_out) *probe_app_samples_out = 0;
    if (s->arith_req == TVR_ARITH_F32) return TVR_OK;                                    // nothing to validate
    if (s->dev.gen) {                                   // more than two encoding frequencies: the lockstep kernels compute with three products whatever the mode says
        tvr_set_error(TVR_OK, "tvr_scene_validate_arith: scenes with more than two encoding frequencies compute in TVR_ARITH_F32 whatever the mode says");
        return TVR_OK;
    }
    if (n_rays <= 0 || !rays || !work) return fail(TVR_ERR_INVALID, "tvr_scene_validate_arith: rays / work NULL or n_rays <= 0");
    if ((uintptr_t)work % 16) return fail(TVR_ERR_SCRATCH, "tvr_scene_validate_arith: work must be 16-byte aligned");
    if (!(tol > 0.0f)) return fail(TVR_ERR_INVALID, "tvr_scene_validate_arith: tol must be positive");
    if (work_bytes < (size_t)n_rays * 8 * sizeof(float) + 256) return fail(TVR_ERR_SCRATCH, "tvr_scene_validate_arith: work %zu B < required %zu B", work_bytes, (size_t)n_rays * 32 + 256);
    hipStream_t stream = (hipStream_t)stream_;
    float *rgb0 = work, *dep0 = work + 3 * n_rays, *rgb1 = work + 4 * n_rays, *dep1 = work + 7 * n_rays;
    unsigned *mx = (unsigned *)(work + 8 * n_rays);
    const MarchSampling sm = {nullptr, nullptr};
    s->dev.arith = TVR_ARITH_F32;
    int rc = render_impl(s, rays, n_rays, S, white_bg, sm, eps_T, rgb0, dep0, nullptr, scratch, scratch_bytes, nullptr, nullptr, nullptr, stream_);
    if (rc != TVR_OK) return rc;
    s->dev.arith = s->arith_req;
    rc = render_impl(s, rays, n_rays, S, white_bg, sm, eps_T, rgb1, dep1, nullptr, scratch, scratch_bytes, nullptr, nullptr, nullptr, stream_);
    s->dev.arith = TVR_ARITH_F32;
    if (rc != TVR_OK) return rc;
    HIP_TRY(launch_zero_f32((float *)mx, 4, stream));                                   // (float4 granules: the 256 B behind the eight arrays are for this)
    hipLaunchKernelGGL(maxdiff_kernel, dim3(256), dim3(256), 0, stream, rgb0, rgb1, (long long)n_rays * 3, mx);
    HIP_TRY(hipGetLastError());
    unsigned bits = 0, shaded = 0;
    HIP_TRY(hipMemcpyAsync(&bits, mx, sizeof(bits), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&shaded, scratch, sizeof(shaded), hipMemcpyDeviceToHost, stream));   // word 0 of the scratch header: the probe's appearance-sample count
    HIP_TRY(hipStreamSynchronize(stream));
    float d;
    memcpy(&d, &bits, sizeof(d));
    *max_diff_out = d;
    if (probe_app_samples_out) *probe_app_samples_out = (int64_t)shaded;
    // A probe that shaded (almost) nothing measured nothing: rays that miss the box or cross empty space give two background pictures, d = 0.  Such a
    // probe neither validates nor refuses — the mode stays un-measured and the caller probes again with rays that hit something.
    const int64_t need = 2 * n_rays < TVR_ARITH_MIN_PROBE_SAMPLES ? 2 * n_rays : TVR_ARITH_MIN_PROBE_SAMPLES;      // (a probe of a few dozen rays: two samples per ray)
    if ((int64_t)shaded < need) {
        s->arith_valid = 0;
        tvr_set_error(TVR_OK, "tvr_scene_validate_arith: mode %d NOT MEASURED — the %lld probe rays shaded %u appearance samples, fewer than the %lld a measurement needs; "
                              "the scene computes in TVR_ARITH_F32 until a probe that hits the scene validates the mode", s->arith_req, (long long)n_rays, shaded,
                      (long long)need);
        return TVR_OK;
    }
    if (d <= tol) {
        s->arith_valid = s->arith_req;
        s->dev.arith = s->arith_req;
        return TVR_OK;
    }
    s->arith_valid = 0;
    tvr_set_error(TVR_OK, "tvr_scene_validate_arith: mode %d REFUSED for these parameters — its picture differs from TVR_ARITH_F32's by %.3g on the %lld probe rays, the tolerance is %.3g; "
                          "the scene computes in TVR_ARITH_F32", s->arith_req, (double)d, (long long)n_rays, (double)tol);
    return TVR_OK;
}

int tvr_march_forward(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const float *jitter, float eps_T, float *depth_out,
                      void *scratch, size_t scratch_bytes, void *stream)
{
    const MarchSampling sm = {jitter, nullptr};
    return march_forward_impl(s, rays, n_rays, S, sm, eps_T, depth_out, nullptr, scratch, scratch_bytes, stream);
}

int tvr_march_forward_z(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const float *z_vals, float eps_T, float *depth_out,
                        float *t_last_tiny_out, void *scratch, size_t scratch_bytes, void *stream)
{
    if (!z_vals) return fail(TVR_ERR_INVALID, "z_vals is NULL");
    const MarchSampling sm = {nullptr, z_vals};
    return march_forward_impl(s, rays, n_rays, S, sm, eps_T, depth_out, t_last_tiny_out, scratch, scratch_bytes, stream);
}

int tvr_filter_rays(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, int32_t bbox_only, uint8_t *mask, size_t mask_bytes, void *stream)
{
    if (!s) return fail(TVR_ERR_INVALID, "scene is NULL");
    if (n_rays < 0 || (!bbox_only && S < 1)) return fail(TVR_ERR_INVALID, "n_rays < 0 or N_samples < 1");
    if (n_rays == 0) return TVR_OK;
    if (!rays || !mask) return fail(TVR_ERR_INVALID, "rays / mask NULL");
    if (mask_bytes < (size_t)n_rays) return fail(TVR_ERR_SCRATCH, "mask holds fewer than n_rays bytes");
    if (!bbox_only && !s->dev.avol) return fail(TVR_ERR_INVALID, "the scene has no alpha mask (tvr_scene_set_alpha): only bbox_only filtering is defined");
    HIP_TRY(launch_filter_rays(s->dev, rays, n_rays, S, bbox_only ? 1 : 0, mask, (hipStream_t)stream));
    return TVR_OK;
}

size_t tvr_grad_scratch_bytes(const tvr_scene *s)
{
    if (!s) return 0;
    return s->lay.mlp_image;      // the VM blocks come first in the packed layout; the gradient images mirror them
}

static TrainGrads carve_grads(const tvr_scene *s, char *g)
{
    TrainGrads tg;
    for (int i = 0; i < 3; ++i) {
        tg.dplane[i] = (float *)(g + s->lay.dplane[i]);
        tg.dline[i] = (float *)(g + s->lay.dline[i]);
        tg.aplane[i] = (float *)(g + s->lay.aplane[i]);
        tg.aline[i] = (float *)(g + s->lay.aline[i]);
    }
    return tg;
}

}  // extern "C"

static int march_backward_impl(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const MarchSampling &sm, float eps_T, const void *fwd_scratch,
                               size_t fwd_scratch_bytes, const float *grad_w, const float *grad_acc, const float *lam6, const float *grad_lam6,
                               void *grad_scratch, size_t grad_scratch_bytes, const tvr_vm_grads *out, void *stream_, long long gw_cap = -1)
{
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (!rays || !fwd_scratch || !grad_w || !grad_acc || !grad_scratch || !out || n_rays <= 0) return fail(TVR_ERR_INVALID, "NULL argument");
    ScratchLayout L = scratch_layout(n_rays, S);
    if (fwd_scratch_bytes < L.total) return fail(TVR_ERR_SCRATCH, "forward scratch %zu B < %zu B", fwd_scratch_bytes, L.total);
    if (grad_scratch_bytes < s->lay.mlp_image || (uintptr_t)grad_scratch % 256) return fail(TVR_ERR_SCRATCH, "gradient scratch too small or misaligned");
    hipStream_t stream = (hipStream_t)stream_;
    const tvr_scene_desc &d = s->desc;
    char *g = (char *)grad_scratch;
    TrainGrads tg = carve_grads(s, g);
    for (int i = 0; i < 3; ++i) {
        const size_t W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        HIP_TRY(launch_zero_f32(tg.dplane[i], (long long)((H + 1) * (W + 1) * TVR_CD), stream));       // (channel counts are multiples of 4, blocks 256-B aligned)
        HIP_TRY(launch_zero_f32(tg.dline[i], (long long)((Ln + 1) * TVR_CD), stream));
    }
    MarchOut mo = carve_scratch((char *)fwd_scratch, L, nullptr);
    HIP_TRY(launch_march_backward(s->dev, rays, (int)n_rays, S, sm, eps_T, mo, grad_w, grad_acc, lam6, grad_lam6, tg, stream, gw_cap));
    for (int i = 0; i < 3; ++i) {
        if (!out->density_plane[i] || !out->density_line[i]) return fail(TVR_ERR_INVALID, "density gradient pointer %d is NULL", i);
        const int W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        HIP_TRY(launch_unpack_grad(tg.dplane[i], out->density_plane[i], s->desc.density_n_comp[i], TVR_CD, H, W, stream));
        HIP_TRY(launch_unpack_grad(tg.dline[i], out->density_line[i], s->desc.density_n_comp[i], TVR_CD, Ln, 1, stream));
    }
    return TVR_OK;
}

extern "C" {

int tvr_march_backward(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const float *jitter, float eps_T, const void *fwd_scratch,
                       size_t fwd_scratch_bytes, const float *grad_w, const float *grad_acc, void *grad_scratch, size_t grad_scratch_bytes,
                       const tvr_vm_grads *out, void *stream)
{
    const MarchSampling sm = {jitter, nullptr};
    return march_backward_impl(s, rays, n_rays, S, sm, eps_T, fwd_scratch, fwd_scratch_bytes, grad_w, grad_acc, nullptr, nullptr, grad_scratch,
                               grad_scratch_bytes, out, stream);
}

int tvr_march_backward_z(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const float *z_vals, float eps_T, const void *fwd_scratch,
                         size_t fwd_scratch_bytes, const float *grad_w, const float *grad_acc, const float *t_last_tiny,
                         const float *grad_t_last_tiny, void *grad_scratch, size_t grad_scratch_bytes, const tvr_vm_grads *out, void *stream)
{
    if (!z_vals) return fail(TVR_ERR_INVALID, "z_vals is NULL");
    if ((t_last_tiny == nullptr) != (grad_t_last_tiny == nullptr)) return fail(TVR_ERR_INVALID, "t_last_tiny and its gradient go together");
    const MarchSampling sm = {nullptr, z_vals};
    return march_backward_impl(s, rays, n_rays, S, sm, eps_T, fwd_scratch, fwd_scratch_bytes, grad_w, grad_acc, t_last_tiny, grad_t_last_tiny,
                               grad_scratch, grad_scratch_bytes, out, stream);
}

int tvr_app_h_forward(tvr_scene *s, const float *xyz, int64_t m, float *h_out, size_t h_bytes, void *stream)
{
    if (m > 0) NEED("h_out [m,144]", h_bytes, m, TVR_KAPP);
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (m == 0) return TVR_OK;
    if (!xyz || !h_out || m < 0) return fail(TVR_ERR_INVALID, "xyz/h_out NULL or m < 0");
    HIP_TRY(launch_app_h_forward(s->dev, xyz, m, h_out, (hipStream_t)stream));
    return TVR_OK;
}

size_t tvr_mlp_train_image_bytes(void) { return mlp_train_image_bytes(); }

static int mlp_train_forward_impl(tvr_scene *s, int variant, const float *h, const float *viewdirs, int64_t m, float *rgb, size_t rgb_bytes, float *feats32,
                                  size_t feats32_bytes, float *h1, size_t h1_bytes, float *h2, size_t h2_bytes, float *g8, size_t g8_bytes, float *rgb_s, size_t rgb_s_bytes,
                                  void *stream)
{
    const char *fn = variant ? "tvr_mlp_train_forward_ref" : "tvr_mlp_train_forward";
    if (m > 0) {
        int rc_;
        if ((rc_ = need_bytes(fn, "rgb [m,3]", rgb_bytes, m, 3)) != TVR_OK) return rc_;
        if ((rc_ = need_bytes(fn, "feats32 [m,32]", feats32_bytes, m, 32)) != TVR_OK) return rc_;
        if ((rc_ = need_bytes(fn, "h1 [m,128]", h1_bytes, m, TVR_FEATC)) != TVR_OK) return rc_;
        if ((rc_ = need_bytes(fn, "h2 [m,128]", h2_bytes, m, TVR_FEATC)) != TVR_OK) return rc_;
        if (variant) {
            if ((rc_ = need_bytes(fn, "g8 [m,8]", g8_bytes, m, 8)) != TVR_OK) return rc_;
            if ((rc_ = need_bytes(fn, "rgb_s [m,3]", rgb_s_bytes, m, 3)) != TVR_OK) return rc_;
        }
        if ((uint64_t)m * 576u >= (1ull << 32)) return fail(TVR_ERR_INVALID, "m = %lld: 32-bit row offsets inside the kernels allow 7.4 M entries per call", (long long)m);
    }
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (s->desc.variant != variant)
        return fail(TVR_ERR_UNSUPPORTED, "%s: the scene is %s", fn, s->desc.variant ? "a REFTensoRF scene (tvr_mlp_train_forward_ref)" : "a TensorVMSplit scene (tvr_mlp_train_forward)");
    {
        const tvr_scene_desc &d = s->desc;
        bool std_shape = d.featureC == TVR_FEATC && d.view_pe == 2 && d.fea_pe == 2;
        for (int i = 0; i < 3; ++i) std_shape = std_shape && d.app_n_comp[i] == TVR_CA;
        if (!std_shape) return fail(TVR_ERR_UNSUPPORTED, "%s: the fused training kernels take 48 appearance components, featureC 128, view_pe = fea_pe = 2 "
                                                         "(zero-padded shapes train through the library-GEMM path)", fn);
    }
    if (m == 0) return TVR_OK;
    if (!h || !viewdirs || !rgb || !feats32 || !h1 || !h2 || m < 0 || (variant && (!g8 || !rgb_s))) return fail(TVR_ERR_INVALID, "NULL argument or m < 0");
    if (((uintptr_t)h | (uintptr_t)feats32 | (uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)g8) % 16) return fail(TVR_ERR_INVALID, "h / feats32 / h1 / h2 / g8 must be 16-byte aligned");
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n = m; sa.h_in = h; sa.viewdirs = viewdirs; sa.out = rgb; sa.t_feats = feats32; sa.t_h1 = h1; sa.t_h2 = h2; sa.t_g8 = g8; sa.t_rgbs = rgb_s;
    HIP_TRY(launch_shade(s->dev, SH_SRC_H, SH_DST_TRAIN, sa, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_mlp_train_forward(tvr_scene *s, const float *h, const float *viewdirs, int64_t m, float *rgb, size_t rgb_bytes, float *feats32, size_t feats32_bytes,
                          float *h1, size_t h1_bytes, float *h2, size_t h2_bytes, void *stream)
{
    return mlp_train_forward_impl(s, 0, h, viewdirs, m, rgb, rgb_bytes, feats32, feats32_bytes, h1, h1_bytes, h2, h2_bytes, nullptr, 0, nullptr, 0, stream);
}

int tvr_mlp_train_forward_ref(tvr_scene *s, const float *h, const float *viewdirs, int64_t m, float *rgb, size_t rgb_bytes, float *feats32, size_t feats32_bytes,
                              float *h1, size_t h1_bytes, float *h2, size_t h2_bytes, float *g8, size_t g8_bytes, float *rgb_s, size_t rgb_s_bytes, void *stream)
{
    return mlp_train_forward_impl(s, 1, h, viewdirs, m, rgb, rgb_bytes, feats32, feats32_bytes, h1, h1_bytes, h2, h2_bytes, g8, g8_bytes, rgb_s, rgb_s_bytes, stream);
}

int tvr_mlp_train_backward(const float *W1, const float *W2, const float *W3, const float *basis, const float *grad_rgb, const float *rgb, const float *feats32,
                           const float *h1, const float *h2, int64_t m, const float *gscale_dev, float *d_out4, size_t d_out4_bytes, float *dh2, size_t dh2_bytes,
                           float *dh1, size_t dh1_bytes, float *dfeats32, size_t dfeats32_bytes, float *dh, size_t dh_bytes, uint32_t *sat_flag_dev,
                           void *image, size_t image_bytes, void *stream)
{
    if (m == 0) return TVR_OK;
    if (m > 0) {
        NEED("d_out4 [m,4]", d_out4_bytes, m, 4);
        NEED("dh2 [m,128]", dh2_bytes, m, TVR_FEATC);
        NEED("dh1 [m,128]", dh1_bytes, m, TVR_FEATC);
        NEED("dfeats32 [m,32]", dfeats32_bytes, m, 32);
        NEED("dh [m,144]", dh_bytes, m, TVR_KAPP);
    }
    if (!W1 || !W2 || !W3 || !basis || !grad_rgb || !rgb || !feats32 || !h1 || !h2 || !gscale_dev || !d_out4 || !dh2 || !dh1 || !dfeats32 || !dh || !image || m < 0)
        return fail(TVR_ERR_INVALID, "NULL argument or m < 0");
    if (image_bytes < mlp_train_image_bytes() || (uintptr_t)image % 256) return fail(TVR_ERR_SCRATCH, "training image buffer too small or misaligned");
    if ((uint64_t)m * 576u >= (1ull << 32)) return fail(TVR_ERR_INVALID, "m = %lld: 32-bit row offsets inside the kernels allow 7.4 M entries per call", (long long)m);
    if (((uintptr_t)feats32 | (uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)d_out4 | (uintptr_t)dh2 | (uintptr_t)dh1 | (uintptr_t)dfeats32 | (uintptr_t)dh) % 16)
        return fail(TVR_ERR_INVALID, "activation / gradient matrices must be 16-byte aligned");
    HIP_TRY(launch_pack_train_image(W1, W2, W3, basis, nullptr, image, (hipStream_t)stream));
    HIP_TRY(launch_mlp_train_backward(grad_rgb, rgb, feats32, h1, h2, m, gscale_dev, d_out4, dh2, dh1, dfeats32, dh, sat_flag_dev, image, nullptr, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_mlp_train_backward_ref(const float *W1, const float *W2, const float *W3, const float *basis, const float *const heads_W[4], const float *grad_rgb,
                               const float *grad_in0, const float *rgb_s, const float *feats32, const float *h1, const float *h2, const float *g8, const float *viewdirs,
                               int64_t m, const float *gscale_dev, float *d_out4, size_t d_out4_bytes, float *dh2, size_t dh2_bytes, float *dh1, size_t dh1_bytes,
                               float *dfeats32, size_t dfeats32_bytes, float *dg8, size_t dg8_bytes, float *dh, size_t dh_bytes, uint32_t *sat_flag_dev, void *image,
                               size_t image_bytes, void *stream)
{
    if (m == 0) return TVR_OK;
    if (m > 0) {
        NEED("d_out4 [m,4]", d_out4_bytes, m, 4);
        NEED("dh2 [m,128]", dh2_bytes, m, TVR_FEATC);
        NEED("dh1 [m,128]", dh1_bytes, m, TVR_FEATC);
        NEED("dfeats32 [m,32]", dfeats32_bytes, m, 32);
        NEED("dg8 [m,8]", dg8_bytes, m, 8);
        NEED("dh [m,144]", dh_bytes, m, TVR_KAPP);
    }
    if (!W1 || !W2 || !W3 || !basis || !heads_W || !heads_W[0] || !heads_W[1] || !heads_W[2] || !heads_W[3] || !grad_rgb || !rgb_s || !feats32 || !h1 || !h2 || !g8 ||
        !viewdirs || !gscale_dev || !d_out4 || !dh2 || !dh1 || !dfeats32 || !dg8 || !dh || !image || m < 0)
        return fail(TVR_ERR_INVALID, "NULL argument or m < 0");
    if (image_bytes < mlp_train_image_bytes() || (uintptr_t)image % 256) return fail(TVR_ERR_SCRATCH, "training image buffer too small or misaligned");
    if ((uint64_t)m * 576u >= (1ull << 32)) return fail(TVR_ERR_INVALID, "m = %lld: 32-bit row offsets inside the kernels allow 7.4 M entries per call", (long long)m);
    if (((uintptr_t)feats32 | (uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)g8 | (uintptr_t)d_out4 | (uintptr_t)dh2 | (uintptr_t)dh1 | (uintptr_t)dfeats32 | (uintptr_t)dg8 | (uintptr_t)dh) % 16)
        return fail(TVR_ERR_INVALID, "activation / gradient matrices must be 16-byte aligned");
    HIP_TRY(launch_pack_train_image(W1, W2, W3, basis, heads_W, image, (hipStream_t)stream));
    MlpRefBwd rb;
    rb.g8 = g8; rb.viewdirs = viewdirs; rb.grad_in0 = grad_in0; rb.dg8 = dg8; rb.rays = nullptr; rb.q_ray = nullptr;
    HIP_TRY(launch_mlp_train_backward(grad_rgb, rgb_s, feats32, h1, h2, m, gscale_dev, d_out4, dh2, dh1, dfeats32, dh, sat_flag_dev, image, &rb, (hipStream_t)stream));
    return TVR_OK;
}

}  // extern "C"

static int app_h_backward_impl(tvr_scene *s, const float *xyz, int xyz_stride, int64_t m, const unsigned *m_dev, const float *dh, void *grad_scratch,
                               size_t grad_scratch_bytes, const tvr_vm_grads *out, void *stream_)
{
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (!grad_scratch || !out || m < 0 || (m > 0 && (!xyz || !dh))) return fail(TVR_ERR_INVALID, "NULL argument");
    if (grad_scratch_bytes < s->lay.mlp_image || (uintptr_t)grad_scratch % 256) return fail(TVR_ERR_SCRATCH, "gradient scratch too small or misaligned");
    hipStream_t stream = (hipStream_t)stream_;
    const tvr_scene_desc &d = s->desc;
    TrainGrads tg = carve_grads(s, (char *)grad_scratch);
    for (int i = 0; i < 3; ++i) {
        const size_t W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        HIP_TRY(launch_zero_f32(tg.aplane[i], (long long)((H + 1) * (W + 1) * TVR_CA), stream));
        HIP_TRY(launch_zero_f32(tg.aline[i], (long long)((Ln + 1) * TVR_CA), stream));
    }
    if (m > 0) HIP_TRY(launch_app_h_backward(s->dev, xyz, m, dh, tg, stream, xyz_stride, m_dev));
    for (int i = 0; i < 3; ++i) {
        if (!out->app_plane[i] || !out->app_line[i]) return fail(TVR_ERR_INVALID, "appearance gradient pointer %d is NULL", i);
        const int W = d.grid[kMatH[i][0]], H = d.grid[kMatH[i][1]], Ln = d.grid[kVecH[i]];
        HIP_TRY(launch_unpack_grad(tg.aplane[i], out->app_plane[i], s->desc.app_n_comp[i], TVR_CA, H, W, stream));
        HIP_TRY(launch_unpack_grad(tg.aline[i], out->app_line[i], s->desc.app_n_comp[i], TVR_CA, Ln, 1, stream));
    }
    return TVR_OK;
}

// ---- the training step without a host read (include/tvr.h: tvr_train_forward / tvr_train_backward) ----
struct WorkLayout {
    size_t h, feats32, h1, h2, rgb, g8, rgb_s, pre, grgb, gin0, grad_w, grad_acc, d_out4, dh2, dh1, dfeats32, dg8, dh, X, tmp, gemm, colsum, image, scalars, total;
};
static WorkLayout work_layout(int64_t n_rays, int32_t S, int64_t cap, bool gen = false)
{
    WorkLayout L;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * sizeof(float), 256); return o; };
    const size_t c = (size_t)cap, n = (size_t)n_rays;
    L.h = take(c * TVR_KAPP); L.feats32 = take(c * 32); L.h1 = take(c * TVR_FEATC); L.h2 = take(c * TVR_FEATC); L.rgb = take(c * 3);
    L.g8 = take(c * 8); L.rgb_s = take(c * 3); L.pre = take(n * 3);
    L.grgb = take(c * 3); L.gin0 = take(c); L.grad_w = take(n * (size_t)S); L.grad_acc = take(n);
    L.d_out4 = take(c * 4); L.dh2 = take(c * TVR_FEATC); L.dh1 = take(c * TVR_FEATC); L.dfeats32 = take(c * 32); L.dg8 = take(c * 8); L.dh = take(c * TVR_KAPP);
    L.X = take(c * (gen ? (size_t)TVR_GENX_FLOATS : (size_t)TVR_GENX_W));    // (more than two encoding frequencies: three column blocks of 152, tvr_train.hip pe_concat_gen_kernel; otherwise one block or [cap,150 / 151])
    L.tmp = take((size_t)TVR_FEATC * TVR_GENX_W + 256);       // gemm_tn results that are wider than the gradient they feed ([4,128], [32,144], [8,144]; a [128,152] block of dW1, dW2 of a narrow network) and bias sums
    size_t g = gemm_tn_scratch_bytes(TVR_FEATC, TVR_GENX_W + 1, cap);        // (+ 1: the ones column that carries the bias gradient)
    L.gemm = take(g / sizeof(float) + 64);
    L.colsum = take(colsum_scratch_bytes() / sizeof(float));
    L.image = take(mlp_train_image_bytes() / sizeof(float) + 64);
    L.scalars = take(64);                                     // [0] max |gradient| bits, [1] gscale
    L.total = off;
    return L;
}

static int train_args_ok(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const void *fwd_scratch, size_t fwd_bytes, const void *work, size_t work_bytes,
                         int64_t app_cap)
{
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    const tvr_scene_desc &d = s->desc;
    // round 6: view_pe / fea_pe up to 6 (TensorVMSplit scenes: the lockstep layer 1 forward, the streamed W1^T backward); REFTensoRF keeps 2 / 2
    // and (TensorVMSplit) any component counts the kernels hold, <= 16 / 48 per plane — TensorBase's own defaults are 8 / 24: the packed scene carries zero channels, basis_mat's
    // columns are mapped in pack_train_image_kernel and its gradient is copied back plane by plane
    // — and any hidden width up to 128 and any frequencies 0 .. 6: every TensorVMSplit shape the scene itself accepts (check_desc) trains through the fused step
    bool std_shape = d.variant == 0 ? (d.featureC >= 1 && d.featureC <= TVR_FEATC && d.view_pe >= 0 && d.view_pe <= TVR_GEN_PE && d.fea_pe >= 0 && d.fea_pe <= TVR_GEN_PE)
                                    : (d.featureC == TVR_FEATC && d.view_pe == 2 && d.fea_pe == 2);
    for (int i = 0; i < 3; ++i) std_shape = std_shape && (d.app_n_comp[i] == TVR_CA || (d.variant == 0 && d.app_n_comp[i] >= 1 && d.app_n_comp[i] <= TVR_CA));
    if (!std_shape) return fail(TVR_ERR_UNSUPPORTED, "the fused training step takes REFTensoRF scenes at featureC 128, view_pe = fea_pe = 2, 48 appearance components per plane (TensorVMSplit: any shape the scene accepts)");
    if (!rays || n_rays <= 0 || S <= 0 || S > 4096 || app_cap <= 0) return fail(TVR_ERR_INVALID, "rays NULL, or n_rays / n_samples / app_cap out of range");
    if ((size_t)n_rays * (size_t)S >= (1ull << 32) || (uint64_t)app_cap * 576u >= (1ull << 32)) return fail(TVR_ERR_INVALID, "n_rays * n_samples and app_cap * 576 must be < 2^32");
    if (!fwd_scratch || fwd_bytes < scratch_layout(n_rays, S).total || (uintptr_t)fwd_scratch % 256) return fail(TVR_ERR_SCRATCH, "forward scratch too small or misaligned");
    if (!work || work_bytes < work_layout(n_rays, S, app_cap, s->dev.gen != 0).total || (uintptr_t)work % 256) return fail(TVR_ERR_SCRATCH, "training workspace too small (tvr_train_work_bytes) or misaligned");
    return TVR_OK;
}

extern "C" {

size_t tvr_train_work_bytes(const tvr_scene *s, int64_t n_rays, int32_t n_samples, int64_t app_cap)
{
    if (n_rays <= 0 || n_samples <= 0 || app_cap <= 0) return 0;
    return work_layout(n_rays, n_samples, app_cap, s && s->dev.gen).total;           // (a scene with more than two encoding frequencies has the wider X)
}

int tvr_train_work_describe(const tvr_scene *s, int64_t n_rays, int32_t n_samples, int64_t app_cap, tvr_train_work_layout *out)
{
    if (!s || !out || n_rays <= 0 || n_samples <= 0 || app_cap <= 0) return fail(TVR_ERR_INVALID, "tvr_train_work_describe: NULL argument or a count <= 0");
    const bool gen = s->dev.gen != 0;
    const WorkLayout W = work_layout(n_rays, n_samples, app_cap, gen);
    out->h = W.h; out->feats32 = W.feats32; out->h1 = W.h1; out->h2 = W.h2; out->rgb = W.rgb; out->grgb = W.grgb; out->d_out4 = W.d_out4; out->dh2 = W.dh2; out->dh1 = W.dh1;
    out->dfeats32 = W.dfeats32; out->dh = W.dh; out->X = W.X; out->total = W.total;
    const int n_in = TVR_APPDIM + 3 + 2 * TVR_APPDIM * s->desc.fea_pe + 6 * s->desc.view_pe;
    const bool blocks = s->desc.variant == 0 && (gen || s->desc.fea_pe != 2 || s->desc.view_pe != 2 || s->desc.featureC < TVR_FEATC);
    out->x_blocks = blocks ? (n_in + TVR_GENX_W - 1) / TVR_GENX_W : 1;
    out->x_block_cols = blocks ? TVR_GENX_W : (s->desc.variant == 1 ? TVR_NIN_REF : TVR_NIN);
    return TVR_OK;
}

int tvr_train_forward(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const float *jitter, float eps_T, int32_t white_bg, void *fwd_scratch,
                      size_t fwd_bytes, void *work, size_t work_bytes, int64_t app_cap, float *rgb_map, float *depth, float *pen_ray, void *stream_)
{
    int rc = train_args_ok(s, rays, n_rays, S, fwd_scratch, fwd_bytes, work, work_bytes, app_cap);
    if (rc != TVR_OK) return rc;
    const bool ref = s->desc.variant == 1;
    if (!rgb_map || !depth || (ref && !pen_ray)) return fail(TVR_ERR_INVALID, "rgb_map / depth (/ pen_ray for a REFTensoRF scene) is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    const MarchSampling sm = {jitter, nullptr};
    rc = march_forward_impl(s, rays, n_rays, S, sm, eps_T, depth, nullptr, fwd_scratch, fwd_bytes, stream_);
    if (rc != TVR_OK) return rc;
    const ScratchLayout SL = scratch_layout(n_rays, S);
    const MarchOut mo = carve_scratch((char *)fwd_scratch, SL, depth);
    const WorkLayout W = work_layout(n_rays, S, app_cap, s->dev.gen != 0);
    char *w = (char *)work;
    HIP_TRY(launch_app_h_forward(s->dev, (const float *)mo.q_pos, app_cap, (float *)(w + W.h), stream, 4, mo.counter));
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n = app_cap; sa.counter = mo.counter; sa.h_in = (const float *)(w + W.h); sa.q_ray = mo.q_ray; sa.rays = rays;
    sa.out = (float *)(w + W.rgb); sa.t_feats = (float *)(w + W.feats32); sa.t_h1 = (float *)(w + W.h1); sa.t_h2 = (float *)(w + W.h2);
    sa.t_g8 = (float *)(w + W.g8); sa.t_rgbs = (float *)(w + W.rgb_s);
    HIP_TRY(launch_shade(s->dev, SH_SRC_H, SH_DST_TRAIN, sa, stream));
    HIP_TRY(launch_composite_train_forward(mo, (int)n_rays, app_cap, white_bg, (const float *)(w + W.rgb), (const float *)(w + W.feats32), ref ? 1 : 0, rgb_map,
                                           (float *)(w + W.pre), pen_ray, stream));
    return TVR_OK;
}

int tvr_train_backward(tvr_scene *s, const float *rays, int64_t n_rays, int32_t S, const float *jitter, float eps_T, int32_t white_bg, const void *fwd_scratch,
                       size_t fwd_bytes, void *work, size_t work_bytes, int64_t app_cap, const tvr_train_weights *wt, const float *grad_rgb_map,
                       const float *grad_pen_ray, float grad_scale_target, void *grad_scratch, size_t grad_scratch_bytes, const tvr_vm_grads *vm_out,
                       const tvr_train_mlp_grads *mo_, uint32_t *sat_flag_dev, void *stream_)
{
    int rc = train_args_ok(s, rays, n_rays, S, fwd_scratch, fwd_bytes, work, work_bytes, app_cap);
    if (rc != TVR_OK) return rc;
    const bool ref = s->desc.variant == 1;
    if (!wt || !wt->W1 || !wt->W2 || !wt->W3 || !wt->basis || !grad_rgb_map || !vm_out || !mo_ || !(grad_scale_target > 0.0f)) return fail(TVR_ERR_INVALID, "NULL argument or grad_scale_target <= 0");
    if (!mo_->W1 || !mo_->b1 || !mo_->W2 || !mo_->b2 || !mo_->W3 || !mo_->b3 || !mo_->basis) return fail(TVR_ERR_INVALID, "a network gradient pointer is NULL");
    if (ref)
        for (int i = 0; i < 4; ++i)
            if (!wt->heads_W[i] || !mo_->heads_W[i] || !mo_->heads_b[i]) return fail(TVR_ERR_INVALID, "REFTensoRF head %d (normal, diffuse, specular, rho): weight or gradient pointer is NULL", i);
    hipStream_t stream = (hipStream_t)stream_;
    const ScratchLayout SL = scratch_layout(n_rays, S);
    const MarchOut mo = carve_scratch((char *)fwd_scratch, SL, nullptr);
    const bool gen = s->dev.gen != 0;
    const WorkLayout W = work_layout(n_rays, S, app_cap, gen);
    char *w = (char *)work;
    auto F = [&](size_t off) { return (float *)(w + off); };
    unsigned *amax = (unsigned *)(w + W.scalars);
    float *gscale = (float *)(w + W.scalars) + 1;
    const unsigned *mdev = mo.counter;
    const int nin = ref ? TVR_NIN_REF : TVR_NIN;
    // 1. compositing backward: gradients of the per-sample colours, of the weights and of acc; the scale of the fused backward
    HIP_TRY(launch_composite_train_backward(mo, (int)n_rays, app_cap, white_bg, F(W.rgb), F(W.feats32), ref ? F(W.g8) : nullptr, F(W.pre), grad_rgb_map, ref ? grad_pen_ray : nullptr,
                                            F(W.grgb), F(W.gin0), F(W.grad_w), F(W.grad_acc), amax, grad_scale_target, gscale, stream));
    // 2. the network backward (register-resident MFMA chains), dh through basis_mat (and the heads)
    const int fc = s->desc.featureC;
    // dW1 in column blocks (tvr_train.hip pe_concat_gen_kernel) for every shape but the kernels' own 2 / 2 at width 128: more than two frequencies (three blocks), fewer (one block
    // of 30 .. 144 columns), a narrower network (its gradients are cropped out of the 128-wide products)
    const bool blocks = !ref && (gen || s->desc.fea_pe != 2 || s->desc.view_pe != 2 || fc < TVR_FEATC);
    HIP_TRY(launch_pack_train_image(wt->W1, wt->W2, wt->W3, wt->basis, ref ? wt->heads_W : nullptr, w + W.image, stream, s->desc.fea_pe, s->desc.view_pe, s->desc.app_n_comp, fc));
    MlpRefBwd rb;
    rb.g8 = F(W.g8); rb.viewdirs = nullptr; rb.grad_in0 = F(W.gin0); rb.dg8 = F(W.dg8); rb.rays = rays; rb.q_ray = mo.q_ray;
    HIP_TRY(launch_mlp_train_backward(F(W.grgb), ref ? F(W.rgb_s) : F(W.rgb), F(W.feats32), F(W.h1), F(W.h2), app_cap, gscale, F(W.d_out4), F(W.dh2), F(W.dh1), F(W.dfeats32),
                                      F(W.dh), sat_flag_dev, w + W.image, ref ? &rb : nullptr, stream, mdev, gen ? 1 : 0));
    // 3. weight gradients: dW = dY^T X over the step's appearance samples (tall-skinny reductions, fixed order), bias gradients = column sums
    if (blocks) HIP_TRY(launch_pe_concat_gen(F(W.feats32), 32, rays, mo.q_ray, s->desc.fea_pe, s->desc.view_pe, app_cap, mdev, F(W.X), stream));
    else if (ref) HIP_TRY(launch_pe_concat_strided(F(W.feats32), 32, F(W.feats32) + 27, 32, nullptr, nullptr, F(W.feats32) + 30, 32, app_cap, mdev, F(W.X), stream));
    else HIP_TRY(launch_pe_concat_strided(F(W.feats32), 32, nullptr, 0, rays, mo.q_ray, nullptr, 0, app_cap, mdev, F(W.X), stream));
    float *tmp = F(W.tmp), *gsc = F(W.gemm), *csc = F(W.colsum);
    // (the bias gradients ride along as a virtual ones column of the second operand: no second pass over d_out / dh2 / dh1)
    float *bs3 = tmp + 32 * TVR_KAPP + 16, *btmp = tmp + TVR_FEATC * TVR_GENX_W + 64;          // (btmp: a 128-entry bias sum of a product whose rows are cropped afterwards)
    // The products run on the fp16-split MFMAs at the backward's own gradient scale (the same values went through fp16 at that scale inside
    // mlp_train_backward / basis_backward, which raise the saturation flag if they do not fit): the kernel then runs at its staging rate.
    HIP_TRY(launch_gemm_tn(F(W.d_out4), 4, 4, F(W.h2), TVR_FEATC, TVR_FEATC, app_cap, tmp, gsc, stream, mdev, bs3, gscale));
    if (fc == TVR_FEATC) HIP_TRY(launch_copy_f32(mo_->W3, tmp, 3 * TVR_FEATC, stream));
    else HIP_TRY(launch_copy_cols(mo_->W3, fc, 0, tmp, TVR_FEATC, fc, 3, stream));             // W3's gradient is [3, fc]: the first fc columns of the [4,128] product
    HIP_TRY(launch_copy_f32(mo_->b3, bs3, 3, stream));
    if (fc == TVR_FEATC) HIP_TRY(launch_gemm_tn(F(W.dh2), TVR_FEATC, TVR_FEATC, F(W.h1), TVR_FEATC, TVR_FEATC, app_cap, mo_->W2, gsc, stream, mdev, mo_->b2, gscale));
    else {                                                                                      // [fc, fc] and [fc] out of the 128-wide product (units that do not exist: exact zeros)
        HIP_TRY(launch_gemm_tn(F(W.dh2), TVR_FEATC, TVR_FEATC, F(W.h1), TVR_FEATC, TVR_FEATC, app_cap, tmp, gsc, stream, mdev, btmp, gscale));
        HIP_TRY(launch_copy_cols(mo_->W2, fc, 0, tmp, TVR_FEATC, fc, fc, stream));
        HIP_TRY(launch_copy_f32(mo_->b2, btmp, fc, stream));
    }
    if (blocks) {
        // dW1 [fc, n_in] block by block (tvr_train.hip pe_concat_gen_kernel): each block's product lands in `tmp` and is copied to its columns; the bias rides with block 0
        const int n_in = TVR_APPDIM + 3 + 2 * TVR_APPDIM * s->desc.fea_pe + 6 * s->desc.view_pe, nb = (n_in + TVR_GENX_W - 1) / TVR_GENX_W;
        for (int b = 0; b < nb; ++b) {
            const int cols = b == nb - 1 ? n_in - b * TVR_GENX_W : TVR_GENX_W, wb = (cols + 3) & ~3;
            HIP_TRY(launch_gemm_tn(F(W.dh1), TVR_FEATC, TVR_FEATC, F(W.X) + (size_t)b * (size_t)app_cap * TVR_GENX_W, wb, wb, app_cap, tmp, gsc, stream, mdev,
                                   b == 0 ? (fc == TVR_FEATC ? mo_->b1 : btmp) : nullptr, gscale));
            HIP_TRY(launch_copy_cols(mo_->W1, n_in, b * TVR_GENX_W, tmp, wb, cols, fc, stream));
            if (b == 0 && fc < TVR_FEATC) HIP_TRY(launch_copy_f32(mo_->b1, btmp, fc, stream));
        }
    } else
    HIP_TRY(launch_gemm_tn(F(W.dh1), TVR_FEATC, TVR_FEATC, F(W.X), nin, nin, app_cap, mo_->W1, gsc, stream, mdev, mo_->b1, gscale));
    HIP_TRY(launch_gemm_tn(F(W.dfeats32), 32, 32, F(W.h), TVR_KAPP, TVR_KAPP, app_cap, tmp, gsc, stream, mdev, nullptr, gscale));
    {
        int k_app = 0;
        for (int i = 0; i < 3; ++i) k_app += s->desc.app_n_comp[i];
        if (k_app == TVR_KAPP) HIP_TRY(launch_copy_f32(mo_->basis, tmp, TVR_APPDIM * TVR_KAPP, stream));
        else                                     // fewer components: the gradient's [27, k_app] takes plane p's columns 48 p .. 48 p + n_p of the [32,144] product
            for (int i = 0, off = 0; i < 3; off += s->desc.app_n_comp[i], ++i)
                HIP_TRY(launch_copy_cols(mo_->basis, k_app, off, tmp + i * TVR_CA, TVR_KAPP, s->desc.app_n_comp[i], TVR_APPDIM, stream));
    }
    if (ref) {
        HIP_TRY(launch_gemm_tn(F(W.dg8), 8, 8, F(W.h), TVR_KAPP, TVR_KAPP, app_cap, tmp, gsc, stream, mdev));      // rows: normal 0..2, specular 3, diffuse 4..6, rho 7
        HIP_TRY(launch_copy_f32(mo_->heads_W[0], tmp, 3 * TVR_KAPP, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_W[2], tmp + 3 * TVR_KAPP, TVR_KAPP, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_W[1], tmp + 4 * TVR_KAPP, 3 * TVR_KAPP, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_W[3], tmp + 7 * TVR_KAPP, TVR_KAPP, stream));
        float *bs = tmp + 32 * TVR_KAPP;
        HIP_TRY(launch_colsum(F(W.dg8), 8, 8, app_cap, mdev, bs, csc, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_b[0], bs, 3, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_b[2], bs + 3, 1, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_b[1], bs + 4, 3, stream));
        HIP_TRY(launch_copy_f32(mo_->heads_b[3], bs + 7, 1, stream));
    }
    // 4. scatter dh into the appearance planes / lines, then the march backward for the density factors
    rc = app_h_backward_impl(s, (const float *)mo.q_pos, 4, app_cap, mdev, F(W.dh), grad_scratch, grad_scratch_bytes, vm_out, stream_);
    if (rc != TVR_OK) return rc;
    const MarchSampling sm = {jitter, nullptr};
    return march_backward_impl(s, rays, n_rays, S, sm, eps_T, fwd_scratch, fwd_bytes, F(W.grad_w), F(W.grad_acc), nullptr, nullptr, grad_scratch, grad_scratch_bytes, vm_out,
                               stream_, app_cap);
}

int tvr_app_h_backward(tvr_scene *s, const float *xyz, int64_t m, const float *dh, size_t dh_bytes, void *grad_scratch, size_t grad_scratch_bytes,
                       const tvr_vm_grads *out, void *stream_)
{
    if (m > 0) NEED("dh [m,144]", dh_bytes, m, TVR_KAPP);
    return app_h_backward_impl(s, xyz, 3, m, nullptr, dh, grad_scratch, grad_scratch_bytes, out, stream_);
}

int tvr_pe_concat(const float *features, const float *viewdirs, const float *dot_product, int64_t m, float *X, size_t X_bytes, void *stream)
{
    if (m == 0) return TVR_OK;
    if (m > 0) NEED(dot_product ? "X [m,151]" : "X [m,150]", X_bytes, m, dot_product ? TVR_NIN_REF : TVR_NIN);
    if (!features || !viewdirs || !X || m < 0) return fail(TVR_ERR_INVALID, "features/viewdirs/X NULL or m < 0");
    HIP_TRY(launch_pe_concat(features, viewdirs, dot_product, m, X, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_pe_concat_backward(const float *features, const float *viewdirs, const float *grad_X, int64_t m, int32_t with_dot, float *grad_features,
                           size_t grad_features_bytes, float *grad_viewdirs, float *grad_dot, void *stream)
{
    if (m == 0) return TVR_OK;
    if (m > 0) NEED("grad_features [m,27]", grad_features_bytes, m, TVR_APPDIM);
    if (!features || !viewdirs || !grad_X || !grad_features || m < 0) return fail(TVR_ERR_INVALID, "NULL argument or m < 0");
    HIP_TRY(launch_pe_concat_backward(features, viewdirs, grad_X, m, with_dot ? 1 : 0, grad_features, grad_viewdirs, grad_dot, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_tv_loss(const float *x, int32_t C, int32_t H, int32_t W, float weight, float *value, float *grad, void *scratch, size_t scratch_bytes,
                void *stream)
{
    if (!x || !value || !grad || C < 1 || H < 1 || W < 1) return fail(TVR_ERR_INVALID, "x/value/grad NULL or bad shape");
    if (!scratch || scratch_bytes < 2048) return fail(TVR_ERR_SCRATCH, "tvr_tv_loss needs 2048 bytes of scratch");
    HIP_TRY(launch_tv_loss(x, C, H, W, weight, value, grad, (float *)scratch, (hipStream_t)stream));
    return TVR_OK;
}

static int reg_list(RegList &L, const float *const *xs, float *const *grads, const int64_t *counts, const int32_t *rows, int32_t n, bool need_grads)
{
    if (n < 1 || n > TVR_REG_MAX) return fail(TVR_ERR_INVALID, "1..%d tensors per call", TVR_REG_MAX);
    if (!xs || !counts || (need_grads && !grads)) return fail(TVR_ERR_INVALID, "xs/counts/grads NULL");
    L.n = n;
    for (int t = 0; t < n; ++t) {
        if (!xs[t] || counts[t] < 1 || (need_grads && !grads[t])) return fail(TVR_ERR_INVALID, "tensor %d: NULL pointer or empty", t);
        L.x[t] = xs[t];
        L.grad[t] = need_grads ? grads[t] : nullptr;
        L.count[t] = counts[t];
        L.rows[t] = rows ? rows[t] : 1;
        L.blocks[t] = 0;
    }
    return TVR_OK;
}

size_t tvr_l1_mean_scratch_bytes(const int64_t *counts, int32_t n)
{
    RegList L;
    if (!counts || n < 1 || n > TVR_REG_MAX) return 0;
    L.n = n;
    for (int t = 0; t < n; ++t) L.count[t] = counts[t] < 1 ? 1 : counts[t];
    return reg_l1_scratch_bytes(L);
}

int tvr_l1_mean(const float *const *xs, const int64_t *counts, int32_t n, float *value, void *scratch, size_t scratch_bytes, void *stream)
{
    RegList L;
    int rc = reg_list(L, xs, nullptr, counts, nullptr, n, false);
    if (rc != TVR_OK) return rc;
    if (!value) return fail(TVR_ERR_INVALID, "value NULL");
    if (!scratch || scratch_bytes < reg_l1_scratch_bytes(L)) return fail(TVR_ERR_SCRATCH, "scratch too small (tvr_l1_mean_scratch_bytes)");
    HIP_TRY(launch_l1_forward(L, value, (float *)scratch, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_l1_mean_backward(const float *const *xs, float *const *grads, const int64_t *counts, int32_t n, const float *grad_value, void *stream)
{
    RegList L;
    int rc = reg_list(L, xs, grads, counts, nullptr, n, true);
    if (rc != TVR_OK) return rc;
    if (!grad_value) return fail(TVR_ERR_INVALID, "grad_value NULL");
    HIP_TRY(launch_l1_backward(L, grad_value, (hipStream_t)stream));
    return TVR_OK;
}

static int ortho_list(RegList &L, const float *const *vs, float *const *grads, const int32_t *n_comp, const int32_t *n_size, int32_t n, bool need_grads)
{
    if (!n_comp || !n_size) return fail(TVR_ERR_INVALID, "n_comp/n_size NULL");
    int64_t counts[TVR_REG_MAX];
    if (n < 1 || n > TVR_REG_MAX) return fail(TVR_ERR_INVALID, "1..%d line factors per call", TVR_REG_MAX);
    for (int t = 0; t < n; ++t) {
        if (n_comp[t] < 2 || n_comp[t] > 48 || n_size[t] < 1) return fail(TVR_ERR_UNSUPPORTED, "line factor %d: %d components x %d (2..48 components)", t, n_comp[t], n_size[t]);
        counts[t] = (int64_t)n_comp[t] * n_size[t];
    }
    return reg_list(L, vs, grads, counts, n_comp, n, need_grads);
}

int tvr_line_ortho(const float *const *vs, const int32_t *n_comp, const int32_t *n_size, int32_t n, float *value, void *scratch, size_t scratch_bytes, void *stream)
{
    RegList L;
    int rc = ortho_list(L, vs, nullptr, n_comp, n_size, n, false);
    if (rc != TVR_OK) return rc;
    if (!value) return fail(TVR_ERR_INVALID, "value NULL");
    if (!scratch || scratch_bytes < TVR_LINE_ORTHO_SCRATCH_BYTES) return fail(TVR_ERR_SCRATCH, "tvr_line_ortho needs %d bytes of scratch", (int)TVR_LINE_ORTHO_SCRATCH_BYTES);
    HIP_TRY(launch_ortho(L, nullptr, value, (float *)scratch, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_line_ortho_backward(const float *const *vs, float *const *grads, const int32_t *n_comp, const int32_t *n_size, int32_t n, const float *grad_value, void *stream)
{
    RegList L;
    int rc = ortho_list(L, vs, grads, n_comp, n_size, n, true);
    if (rc != TVR_OK) return rc;
    if (!grad_value) return fail(TVR_ERR_INVALID, "grad_value NULL");
    HIP_TRY(launch_ortho(L, grad_value, nullptr, nullptr, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_linear_dx(const float *dY, int32_t ldy, int32_t N, const float *W, int32_t ldw, int32_t n_valid, int32_t K, const float *mask, int32_t ldm,
                  const uint64_t *mask_bits, float *dX, int32_t ldx, size_t dX_bytes, int64_t M, const float *scale_dev, uint32_t *sat_flag_dev, void *stream)
{
    if (mask && mask_bits) return fail(TVR_ERR_INVALID, "mask and mask_bits are alternatives");
    if ((uintptr_t)mask_bits & 7) return fail(TVR_ERR_INVALID, "mask_bits is not 8-B aligned");
    if (scale_dev && (N & 15)) return fail(TVR_ERR_INVALID, "the fp16-split form (scale_dev) takes N in multiples of 16 (N = %d)", N);
    if (M < 0) return fail(TVR_ERR_INVALID, "M < 0");
    if (N < 8 || N > 128 || (N & 7) || n_valid < 1 || n_valid > N) return fail(TVR_ERR_UNSUPPORTED, "N = %d (a multiple of 8 in [8,128]) / n_valid = %d", N, n_valid);
    if (K != 32 && K != 64 && K != 96 && K != 128) return fail(TVR_ERR_UNSUPPORTED, "K = %d (32, 64, 96 or 128)", K);
    if (ldy < N || ldx < K || ldw < K || (mask && ldm < K) || (ldy & 3) || (ldx & 3) || (mask && (ldm & 3))) return fail(TVR_ERR_INVALID, "row strides: >= the row length and multiples of 4");
    if (M == 0) return TVR_OK;
    if (!dY || !W || !dX || ((uintptr_t)dY & 15) || ((uintptr_t)dX & 15) || ((uintptr_t)mask & 15)) return fail(TVR_ERR_INVALID, "dY / W / dX NULL or not 16-B aligned");
    if (dX_bytes < ((size_t)(M - 1) * ldx + K) * sizeof(float)) return fail(TVR_ERR_SCRATCH, "dX holds fewer than M rows");
    HIP_TRY(launch_linear_dx(dY, ldy, N, W, ldw, n_valid, K, mask, ldm, dX, ldx, M, (hipStream_t)stream, scale_dev, sat_flag_dev,
                             (const unsigned long long *)mask_bits));
    return TVR_OK;
}

size_t tvr_colsum_scratch_bytes(void) { return colsum_scratch_bytes(); }

int tvr_colsum(const float *A, int32_t lda, int32_t K, int64_t M, float *out, void *scratch, size_t scratch_bytes, void *stream)
{
    if (M < 0 || K < 1 || K > 128 || lda < K) return fail(TVR_ERR_INVALID, "bad M / K / lda (K <= 128)");
    if (!out || (M > 0 && !A)) return fail(TVR_ERR_INVALID, "A / out NULL");
    if (!scratch || scratch_bytes < colsum_scratch_bytes()) return fail(TVR_ERR_SCRATCH, "scratch too small (tvr_colsum_scratch_bytes)");
    HIP_TRY(launch_colsum(A, lda, K, M, nullptr, out, (float *)scratch, (hipStream_t)stream));
    return TVR_OK;
}

static int gemm_tn_check(int32_t Ka, int32_t Kb, int64_t M)
{
    if (M < 0 || Ka < 1 || Kb < 1) return fail(TVR_ERR_INVALID, "bad Ka/Kb/M");
    if (((Ka + 31) / 32) * ((Kb + 31) / 32) > 20) return fail(TVR_ERR_UNSUPPORTED, "Ka x Kb = %d x %d exceeds the 20 32x32 tiles of a workgroup", Ka, Kb);
    return TVR_OK;
}

size_t tvr_gemm_tn_scratch_bytes(int32_t Ka, int32_t Kb, int64_t M)
{
    if (gemm_tn_check(Ka, Kb, M) != TVR_OK) return 0;
    return gemm_tn_scratch_bytes(Ka, Kb, M);
}

int tvr_gemm_tn(const float *A, int32_t lda, int32_t Ka, const float *B, int32_t ldb, int32_t Kb, int64_t M, float *C, void *scratch,
                size_t scratch_bytes, void *stream)
{
    int rc = gemm_tn_check(Ka, Kb, M);
    if (rc != TVR_OK) return rc;
    if (!C || (M > 0 && (!A || !B))) return fail(TVR_ERR_INVALID, "A/B/C NULL");
    if (lda < Ka || ldb < Kb) return fail(TVR_ERR_INVALID, "lda/ldb smaller than the row length");
    if (M > 0 && (!scratch || scratch_bytes < gemm_tn_scratch_bytes(Ka, Kb, M))) return fail(TVR_ERR_SCRATCH, "scratch too small (tvr_gemm_tn_scratch_bytes)");
    HIP_TRY(launch_gemm_tn(A, lda, Ka, B, ldb, Kb, M, C, (float *)scratch, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_gemm_tn_scaled(const float *A, int32_t lda, int32_t Ka, const float *B, int32_t ldb, int32_t Kb, int64_t M, float *C, float *colsum_A, const float *scale_dev,
                       void *scratch, size_t scratch_bytes, void *stream)
{
    const int ones = colsum_A ? 1 : 0;
    int rc = gemm_tn_check(Ka, Kb + ones, M);
    if (rc != TVR_OK) return rc;
    if (!C || !scale_dev || (M > 0 && (!A || !B))) return fail(TVR_ERR_INVALID, "A/B/C/scale_dev NULL");
    if (lda < Ka || ldb < Kb) return fail(TVR_ERR_INVALID, "lda/ldb smaller than the row length");
    if (Ka > 128 || 16 * (Ka + Kb) > 256 * 20) return fail(TVR_ERR_UNSUPPORTED, "the fp16-split form takes Ka <= 128 and Ka + Kb <= 320 (Ka = %d, Kb = %d)", Ka, Kb);
    if (M > 0 && (!scratch || scratch_bytes < gemm_tn_scratch_bytes(Ka, Kb + ones, M))) return fail(TVR_ERR_SCRATCH, "scratch too small (tvr_gemm_tn_scratch_bytes(Ka, Kb + 1, M))");
    HIP_TRY(launch_gemm_tn(A, lda, Ka, B, ldb, Kb, M, C, (float *)scratch, (hipStream_t)stream, nullptr, colsum_A, scale_dev));
    return TVR_OK;
}

int tvr_gemm_tn_bias(const float *A, int32_t lda, int32_t Ka, const float *B, int32_t ldb, int32_t Kb, int64_t M, float *C, float *colsum_A, void *scratch,
                     size_t scratch_bytes, void *stream)
{
    int rc = gemm_tn_check(Ka, Kb + 1, M);
    if (rc != TVR_OK) return rc;
    if (!C || !colsum_A || (M > 0 && (!A || !B))) return fail(TVR_ERR_INVALID, "A/B/C/colsum_A NULL");
    if (lda < Ka || ldb < Kb) return fail(TVR_ERR_INVALID, "lda/ldb smaller than the row length");
    if (16 * (Ka + Kb) > 256 * 20) return fail(TVR_ERR_UNSUPPORTED, "Ka + Kb = %d exceeds the 320 staged columns of the kernel that carries the ones column", Ka + Kb);
    if (M > 0 && (!scratch || scratch_bytes < gemm_tn_scratch_bytes(Ka, Kb + 1, M))) return fail(TVR_ERR_SCRATCH, "scratch too small (tvr_gemm_tn_scratch_bytes(Ka, Kb + 1, M))");
    HIP_TRY(launch_gemm_tn(A, lda, Ka, B, ldb, Kb, M, C, (float *)scratch, (hipStream_t)stream, nullptr, colsum_A));
    return TVR_OK;
}

int tvr_density_feature(tvr_scene *s, const float *xyz, int64_t m, float *out, size_t out_bytes, void *stream)
{
    if (m > 0) NEED("out [m]", out_bytes, m, 1);
    if (!s || !s->params_set) return fail(TVR_ERR_INVALID, "scene is NULL or tvr_scene_update has not run");
    if (m == 0) return TVR_OK;
    if (!xyz || !out || m < 0) return fail(TVR_ERR_INVALID, "xyz/out NULL or m < 0");
    HIP_TRY(launch_density_feature(s->dev, xyz, m, out, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_app_feature(tvr_scene *s, const float *xyz, int64_t m, float *out, size_t out_bytes, void *stream)
{
    if (m > 0) NEED("out [m,27]", out_bytes, m, TVR_APPDIM);
    if (!s || !s->params_set) return fail(TVR_ERR_INVALID, "scene is NULL or tvr_scene_update has not run");
    if (m == 0) return TVR_OK;
    if (!xyz || !out || m < 0) return fail(TVR_ERR_INVALID, "xyz/out NULL or m < 0");
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n = m;
    sa.xyz = xyz;
    sa.out = out;
    HIP_TRY(launch_shade(s->dev, SH_SRC_XYZ, SH_DST_FEAT, sa, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_app_feature_ref(tvr_scene *s, const float *xyz, int64_t m, float *features, size_t features_bytes, float *extra, size_t extra_bytes, void *stream)
{
    if (m > 0) {
        NEED("features [m,27]", features_bytes, m, TVR_APPDIM);
        NEED("extra [m,8]", extra_bytes, m, 8);
    }
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (s->desc.variant != 1) return fail(TVR_ERR_INVALID, "tvr_app_feature_ref needs a REFTensoRF (variant 1) scene");
    if (m == 0) return TVR_OK;
    if (!xyz || !features || !extra || m < 0) return fail(TVR_ERR_INVALID, "xyz/features/extra NULL or m < 0");
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n = m;
    sa.xyz = xyz;
    sa.out = features;
    sa.out2 = extra;
    HIP_TRY(launch_shade(s->dev, SH_SRC_XYZ, SH_DST_FEAT, sa, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_mlp_render_ref(tvr_scene *s, const float *viewdirs, const float *features, const float *dot_product, int64_t m, float *rgb,
                       size_t rgb_bytes, void *stream)
{
    if (m > 0) NEED("rgb [m,3]", rgb_bytes, m, 3);
    int rc = scene_ready(s);
    if (rc != TVR_OK) return rc;
    if (s->desc.variant != 1) return fail(TVR_ERR_INVALID, "tvr_mlp_render_ref needs a REFTensoRF (variant 1) scene");
    if (m == 0) return TVR_OK;
    if (!viewdirs || !features || !dot_product || !rgb || m < 0) return fail(TVR_ERR_INVALID, "viewdirs/features/dot_product/rgb NULL or m < 0");
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n = m;
    sa.viewdirs = viewdirs;
    sa.feats = features;
    sa.dots = dot_product;
    sa.out = rgb;
    HIP_TRY(launch_shade(s->dev, SH_SRC_FEAT, SH_DST_RGB, sa, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_mlp_render(tvr_scene *s, const float *viewdirs, const float *features, int64_t m, float *rgb, size_t rgb_bytes, void *stream)
{
    if (m > 0) NEED("rgb [m,3]", rgb_bytes, m, 3);
    if (!s || !s->params_set) return fail(TVR_ERR_INVALID, "scene is NULL or tvr_scene_update has not run");
    if (s->desc.variant != 0) return fail(TVR_ERR_INVALID, "tvr_mlp_render is MLPRender_Fea; a REFTensoRF scene takes tvr_mlp_render_ref");
    if (m == 0) return TVR_OK;
    if (!viewdirs || !features || !rgb || m < 0) return fail(TVR_ERR_INVALID, "viewdirs/features/rgb NULL or m < 0");
    ShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n = m;
    sa.viewdirs = viewdirs;
    sa.feats = features;
    sa.out = rgb;
    HIP_TRY(launch_shade(s->dev, SH_SRC_FEAT, SH_DST_RGB, sa, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_alpha_sample(const float *vol, const int32_t ag[3], const float aabb[6], const float inv[3], const float *xyz,
                     int64_t m, float *out, size_t out_bytes, void *stream)
{
    if (m > 0) NEED("out [m]", out_bytes, m, 1);
    if (!vol || !ag || !aabb || !inv) return fail(TVR_ERR_INVALID, "alpha volume/grid/aabb/inv is NULL");
    if (m == 0) return TVR_OK;
    if (!xyz || !out || m < 0) return fail(TVR_ERR_INVALID, "xyz/out NULL or m < 0");
    SceneDev v;
    memset(&v, 0, sizeof(v));
    for (int k = 0; k < 3; ++k) {
        if (ag[k] < 1) return fail(TVR_ERR_INVALID, "alpha grid[%d]=%d", k, ag[k]);
        v.ag[k] = ag[k];
        v.alo[k] = aabb[k];
        v.ainv[k] = inv[k];
        v.agm1[k] = (float)(ag[k] - 1);
    }
    v.avol = vol;
    HIP_TRY(launch_alpha_sample(v, xyz, m, out, (hipStream_t)stream));
    return TVR_OK;
}

int tvr_profile_create(int32_t max_calls, tvr_profile **out)
{
    if (!out || max_calls <= 0 || max_calls > 100000) return fail(TVR_ERR_INVALID, "bad profile arguments");
    tvr_profile *p = new (std::nothrow) tvr_profile;
    if (!p) return fail(TVR_ERR_INVALID, "out of host memory");
    p->max_calls = max_calls;
    p->n_calls = 0;
    p->n_sets = 0;
    p->ev.reserve((size_t)max_calls * 4);      // events are created as launch sets are recorded (a piecewise call records one set per piece) and kept across resets
    *out = p;
    return TVR_OK;
}

int tvr_profile_reset(tvr_profile *p)
{
    if (!p) return fail(TVR_ERR_INVALID, "profile is NULL");
    p->n_calls = 0;
    p->n_sets = 0;
    return TVR_OK;
}

int tvr_profile_read(tvr_profile *p, float ms[3])
{
    if (!p || !ms) return fail(TVR_ERR_INVALID, "profile/ms is NULL");
    ms[0] = ms[1] = ms[2] = 0.f;
    for (int c = 0; c < p->n_sets; ++c)
        for (int k = 0; k < 3; ++k) {
            float t = 0.f;
            hipError_t rc = hipEventElapsedTime(&t, p->ev[(size_t)c * 4 + k], p->ev[(size_t)c * 4 + k + 1]);
            if (rc != hipSuccess) return fail(TVR_ERR_HIP, "hipEventElapsedTime: %s", hipGetErrorString(rc));
            ms[k] += t;
        }
    return p->n_calls;
}

}  // extern "C"

// four events for the next launch set (created on first use, re-used after tvr_profile_reset)
static hipEvent_t *profile_slot(tvr_profile *prof)
{
    const size_t need = ((size_t)prof->n_sets + 1) * 4;
    while (prof->ev.size() < need) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        prof->ev.push_back(e);
    }
    return &prof->ev[(size_t)prof->n_sets++ * 4];
}

extern "C" {

int tvr_profile_destroy(tvr_profile *p)
{
    if (!p) return TVR_OK;
    for (auto &e : p->ev) (void)hipEventDestroy(e);
    delete p;
    return TVR_OK;
}

}  // extern "C"
