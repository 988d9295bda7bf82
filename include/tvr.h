/* tvr.h — C-ABI of the MI355X-native TensoRF volume renderer (libtvr.so, gfx950).
 *
 * The reference (FREDZEL2020/jittor-MYC-NeRFs, tensorf-myc) has NO FFI on this path: the hot path is
 * Python calling Jittor ops.  The boundary kept is therefore the Python call surface of the model and
 * renderer; this header is the C-ABI underneath it.  Each entry point cites the
 * reference interface (paths relative to /root/reference/) whose work it performs.
 *
 * Conventions
 *   - every function returns 0 on success, a negative tvr_status otherwise; tvr_last_error() gives a
 *     thread-local message.  Nothing throws, nothing calls hipDeviceSynchronize.
 *   - ALL device memory is caller-owned (torch allocations): fp32, contiguous, 16-byte aligned.
 *     The library never allocates device memory; sizes come from the *_bytes() queries.
 *   - all work is enqueued on the caller-supplied stream (a hipStream_t passed as void*).
 *   - every OUTPUT matrix whose row width is the kernels' own (not a shape the reference shows) is passed with its size in bytes
 *     (`*_bytes`) and checked on the host before anything is launched: a buffer that is too small is refused with TVR_ERR_SCRATCH
 *     instead of being overrun on the device.  (Round 2: a caller that allocated tvr_app_h_forward's h as [m, sum(app_n_comp)]
 *     instead of the kernels' [m,144] had 144 - sum columns written past its end — a device fault at the next sync.)
 *     tvr_render's rgb_out [n,3] / depth_out [n] and the tvr_vm_grads tensors have the reference's own shapes and carry no count.
 *   - a tvr_scene may be used from one stream at a time; distinct scenes are independent.
 */
#ifndef TVR_H
#define TVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVR_VERSION 141   /* 141 (round 6): tvr_train_forward / _backward take TensorVMSplit scenes with up to six encoding frequencies; tvr_train_work_describe.
                           * 140 (round 6): tvr_mlpnet_forward / _train_forward take packed_bytes and a caller-owned work buffer (tvr_mlpnet_work_bytes);
                           * tvr_mlpnet_packed_bytes no longer counts a ticket word; no entry point writes through a const pointer */

typedef enum {
    TVR_OK = 0,
    TVR_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    TVR_ERR_HIP = -2,          /* a HIP runtime call failed */
    TVR_ERR_SCRATCH = -3,      /* scratch / packed buffer too small or misaligned */
    TVR_ERR_UNSUPPORTED = -4   /* configuration outside what the kernels are built for */
} tvr_status;

typedef struct tvr_scene tvr_scene;
typedef struct tvr_profile tvr_profile;

/* Hyper-parameters of one field: TensorBase.__init__ (tensorf-myc/models/tensorBase.py:141-176),
 * update_stepSize (:197-209) and TensorVMSplit.init_svd_volume (tensorf-myc/models/tensoRF.py:146-164). */
typedef struct {
    float aabb[6];                 /* lo[3], hi[3] */
    int32_t grid[3];               /* gridSize (x, y, z) */
    int32_t density_n_comp[3];     /* 1..16 per plane (the kernels are built for 16; fewer are packed with zero channels, exact) */
    int32_t app_n_comp[3];         /* 1..48 per plane (built for 48; zero-padded likewise) */
    int32_t app_dim;               /* 27 */
    int32_t featureC;              /* 1..128 (built for 128; hidden units that do not exist are zero weights) */
    int32_t view_pe, fea_pe;       /* 0..6 each (shadingMode MLP_Fea).  0..2: the fused kernels (built for 2, 2; every shipped config).  3..6 — TensorBase's own constructor
                                    * defaults are 6, 6 (tensorBase.py:141-145) — render through the lockstep layer-1 form (+213 KB of packed weights, ~1.6 x the frame time)
                                    * and TRAIN through the eager chain (tvr_train_forward / tvr_mlp_train_forward refuse them).  variant 1 (REFTensoRF): 2, 2 only.
                                    * Anything larger: TVR_ERR_UNSUPPORTED from tvr_scene_packed_bytes / tvr_scene_create */
    float near_, far_;             /* near_far */
    float step_size;               /* stepSize = mean(units)*step_ratio, computed by the host in fp32 */
    float inv_aabb_size[3];        /* invaabbSize = 2/(hi-lo), computed by the host in fp32 (:201) */
    float density_shift;           /* -10 */
    float distance_scale;          /* 25 */
    float weight_thres;            /* rayMarch_weight_thres 1e-4 */
    int32_t fea2dense_act;         /* 0 softplus, 1 relu (:444-448) */
    int32_t variant;               /* 0 TensorVMSplit (models/tensoRF.py:141), 1 REFTensoRF (models/REFTensoRF.py:64) */
} tvr_scene_desc;

/* Device pointers to the parameters in the REFERENCE layout (tensoRF.py:154-164, tensorBase.py:69-71):
 * planes (1,C,H,W) channel-first, lines (1,C,L,1), Linear weights [out,in] row-major. */
typedef struct {
    const float *density_plane[3], *density_line[3];
    const float *app_plane[3], *app_line[3];
    const float *basis_mat;              /* [app_dim, sum(app_n_comp)]  (144 columns at the built-for shape) */
    const float *W1, *b1, *W2, *b2, *W3, *b3;   /* W1 [featureC, 30 + 54 fea_pe + 6 view_pe] = [128,150]; variant 1: [128,151] (MLPRender_Fea_Ref, REFTensoRF.py:9-16);
                                                 * W2 [featureC,featureC], W3 [3,featureC] */
    /* variant 1 only (REFTensoRF.init_svd_volume, REFTensoRF.py:86-96): normal_linear [3,144], diffuse_linear [3,144],
     * specular_linear [1,144], rho_linear [1,144] and their biases, in that order */
    const float *ref_W[4], *ref_b[4];
} tvr_scene_params;

/* Optional per-sample outputs (additional_output=True of TensorBase.execute, tensorBase.py:533-534,
 * plus the intermediates the parity tests compare bit for bit).  Any pointer may be NULL. */
typedef struct {
    float *z;            /* [n,S]   z_vals */
    uint8_t *valid;      /* [n,S]   ray_valid after the alpha mask */
    uint8_t *bbox_valid; /* [n,S]   in-box test only */
    int32_t *cell;       /* [n,S,3] floor of the un-normalised grid coordinate per axis */
    float *sigma_feature;/* [n,S] */
    float *sigma;        /* [n,S] */
    float *alpha;        /* [n,S] */
    float *weight;       /* [n,S] */
    float *rgb;          /* [n,S,3] (zero where weight <= thres) */
    float *bg_weight;    /* [n]     T after the last sample */
    float *acc;          /* [n]     sum of weights */
    float *t_min;        /* [n] */
} tvr_dense_out;

/* Counters added to by tvr_render when `stats` is non-NULL (device memory, 8 x uint64, caller zeroes). */
enum { TVR_STAT_SAMPLES_EVAL = 0,   /* density samples actually gathered (valid, before termination) */
       TVR_STAT_SAMPLES_BBOX = 1,   /* in-box samples visited (alpha-mask lookups when a mask is set) */
       TVR_STAT_APP = 2,            /* appearance samples (weight > thres) */
       TVR_STAT_RAYS_TERMINATED = 3,/* rays stopped early by eps_T */
       /* clock probes, summed over the workgroups of a launch: shader-clock ticks (s_memtime) and 100 MHz reference ticks (s_memrealtime) between
        * a workgroup's first and last instruction -> the clock the kernel really ran at = 0.1 GHz * CLK / REF (bench.py's roofline peaks) */
       TVR_STAT_MARCH_CLK = 4, TVR_STAT_MARCH_REF = 5, TVR_STAT_SHADE_CLK = 6, TVR_STAT_SHADE_REF = 7,
       TVR_STAT_COUNT = 8 };

int tvr_version(void);
const char *tvr_last_error(void);

/* Packed (channels-last, zero-padded) copy of the parameters: size query, create, refresh, destroy.
 * Replaces nothing in the reference — it is the HBM layout behind TensorVMSplit's ParameterLists. */
size_t tvr_scene_packed_bytes(const tvr_scene_desc *desc);
int tvr_scene_create(const tvr_scene_desc *desc, void *packed_dev, size_t packed_bytes, tvr_scene **out);
int tvr_scene_update(tvr_scene *scene, const tvr_scene_params *params, void *stream);
/* CONCURRENCY: a scene's packed images are shared by every call that names the scene.  Any number of tvr_render(_z) calls may be in flight on different streams at once as
 * long as each has its own scratch and output buffers (they only READ the scene; render.py::FrameStream keeps two frames in flight this way).  Whatever WRITES the scene's
 * device state — tvr_scene_update, tvr_scene_set_alpha, tvr_scene_validate_arith, the first TVR_ARITH_F16 render after an update (it converts the fp16 copies) — must be
 * ordered by the caller against every call still reading it (stream waits or events); the library inserts no cross-stream synchronisation. */
/* tvr_scene_update captured into a hipGraph (a whole training step, tvr_train_forward's host does that): every REPLAY re-packs the fp32 images on the device and runs no
 * host code, so whatever the host derived from "the parameters as last packed" is stale afterwards — (a) a range proof that switched the fp16-range check off
 * (tvr_scene_set_range_check(scene, 0)): switch it back on, or prove again, before rendering; (b) the fp16 copies of the appearance factors that TVR_ARITH_F16 gathers
 * (converted by an un-captured update or by the first render in that mode): call tvr_scene_touch() before the next render, which then converts first.
 * tvr_scene_touch: "the packed fp32 images were rewritten behind the host's back" — marks every derived copy stale.  Host-only, no launch. */
int tvr_scene_touch(tvr_scene *scene);
/* AlphaGridMask (tensorBase.py:39-59): volume (gz,gy,gx) fp32 (non-negative) in device memory, kept by reference; NULL clears.
 * bits (optional, tvr_alpha_bits_bytes() of device memory, kept by reference): the march then tests `sample_alpha(p) > 0` (:491-496) on a
 * bit volume built here from the float one — same result, 1/32 of the footprint; NULL keeps the 8-tap float lookup. */
size_t tvr_alpha_bits_bytes(const int32_t agrid_xyz[3]);
int tvr_scene_set_alpha(tvr_scene *scene, const float *alpha_volume_dev, const int32_t agrid_xyz[3],
                        const float alpha_aabb[6], const float alpha_inv_size[3], void *bits, size_t bits_bytes, void *stream);
/* fp16-range check of the inference entry points (tvr_render(_z), tvr_app_feature(_ref), tvr_mlp_render(_ref)); ON when a scene is created.
 * ON: every value that enters a matrix product through the fp16 hi / lo split is held against fp16's largest finite value on the way (one v_max3 per two
 * values, ~1.5 % of the shade kernel's time); an appearance sample with an operand at or beyond 65 504 gets NaN as its colour / features, so its pixel is NaN,
 * never a silently clipped product.  OFF: no check — for hosts that have PROVEN the range from the parameters (the Python host does, by interval bounds, at
 * pack time: field.py::TensorVMSplit._fp16_range_proven) and want the last 1.5 %.  Weights are the host's to check (a bound on max|W| needs no kernel). */
int tvr_scene_set_range_check(tvr_scene *scene, int32_t on);
/* Arithmetic of the appearance network's matrix products (basis 144->27 and REFTensoRF's heads, layers 1 and 2) in tvr_render(_z) and tvr_mlp_render(_ref) of a scene
 * with at most two encoding frequencies.  The reference computes them in fp32 (tensorBase.py:76-86, tensoRF.py:243); fp32 accumulation in every mode:
 *   TVR_ARITH_F32    (default) three fp16 products per fp32 product — weights AND activations as fp16 hi + lo: ~2^-22 per product, fp32-class;
 *   TVR_ARITH_F16ACT layers 1 and 2 take two products — weights keep hi + lo (22 bits), their inputs (features, encoded values, relu outputs) are rounded to
 *                    fp16 (nearest even, 2^-12 relative); the basis product and REFTensoRF's heads keep three (their outputs feed sin / cos, where an error is amplified): 0.70 of the matrix work;
 *   TVR_ARITH_F16    one product — activations as fp16 (nearest even), weights as the HI image of the default mode, i.e. truncated toward zero to fp16 (a one-sided
 *                    2^-11 per weight instead of an unbiased 2^-12: the measured errors below include it): 1/3 of the matrix work; and the gather of tvr_render(_z) reads fp16 COPIES of the appearance
 *                    planes / lines (kept in the packed buffer, converted from the fp32 images with round-to-nearest-even by the first TVR_ARITH_F16 render behind an
 *                    update — tvr_scene_update itself converts nothing): half the bytes through the L1 return path, interpolation still in fp32.
 *                    (A hipGraph that captured a render bakes in the mode and whether a conversion was due: capture again after switching modes.)
 * fp16 rounding is RELATIVE: the reduced modes' absolute error grows with the scale of the features and hidden activations (|feature| <= 23: picture within 6.4e-5 / 3.5e-4
 * of the fp32 path in F16ACT / F16; |feature| ~ 230: 6.5e-4 / 1.5e-3) — a scene with unusually large features keeps the default.
 * The reduced modes are OPT-IN trades inside north_star's parity bar (RGB L-inf 1e-3 against the fp32 path): measured against TVR_ARITH_F32 on the 800x800 bench frame
 * and against the oracle on the fixtures, see DESIGN.md 4.7 and tests/test_gpu_arith.py for the numbers and the bars the tests hold.  Layer 3, encoding, interpolation,
 * density, compositing: fp32 in every mode.  (NerfPlusPlus's background network has the same switch in its descriptor: tvr_mlpnet_desc.arith.)  Every other entry point (tvr_app_feature(_ref), the training forwards, scenes with more than two encoding
 * frequencies) computes with three products whatever the mode says.  Range: as for the default (|x| < 65 504); a rounded activation beyond it becomes inf, and
 * the range check marks its sample NaN as in the default mode.
 * GATE (round 5): a reduced mode never runs on parameters it has not been MEASURED on.  fp16 rounding is relative, so no bound from the parameters alone is useful
 * (interval bounds on |W| |x| overestimate the error a thousandfold and would refuse every scene); instead tvr_scene_validate_arith renders a caller-chosen probe batch
 * of rays in TVR_ARITH_F32 and in the requested mode and compares the pictures:
 *   tvr_scene_set_arith(scene, mode)   records the REQUEST; the kernels keep computing in TVR_ARITH_F32 (tvr_last_error() says so, the call returns TVR_OK) unless
 *                                      the mode is already validated for the parameters as packed;
 *   tvr_scene_validate_arith(...)      max |rgb_mode - rgb_f32| over the probe rays <= tol: the mode is IN EFFECT from here on; otherwise the scene stays in
 *                                      TVR_ARITH_F32 and tvr_last_error() carries the measured difference (TVR_OK either way: *max_diff_out and tvr_scene_get_arith tell).
 *                                      One host synchronisation.  work: (8 n_rays) floats + 256 B of device memory, 16-byte aligned.  The Python host calls it with up to 8192 rays of the
 *                                      first inference batch after every parameter change and a tolerance of 2.5e-4 (a quarter of the parity bar);
 *   tvr_scene_update / tvr_scene_touch void the validation (new parameters): TVR_ARITH_F32 until validated again;
 *   tvr_scene_get_arith                the mode IN EFFECT (what the next render computes in); tvr_scene_get_arith_requested the request.
 * So tvr_render(_z) / tvr_mlp_render(_ref) cannot leave the parity bar silently through a reduced mode: either the mode was measured on this scene's own rays, or it does not run.
 * (tvr_mlpnet_desc.arith, NerfPlusPlus's background network, is a field of a stateless descriptor: the library cannot gate it — the Python host measures it the same way,
 *  variants.py::NerfPlusPlus._settle_bg_arith, and a C host must do likewise.)
 *   A probe that shades fewer than min(TVR_ARITH_MIN_PROBE_SAMPLES, 2 n_rays) appearance samples (rays that miss the box, empty space: both pictures are background, the difference is 0)
 *   has measured nothing: it neither validates nor refuses — the mode stays out of effect, *probe_app_samples_out (optional) tells, and the caller probes again with
 *   rays that hit the scene (round 6; before, such a probe opened the gate for free). */
enum { TVR_ARITH_F32 = 0, TVR_ARITH_F16ACT = 1, TVR_ARITH_F16 = 2 };
#define TVR_ARITH_MIN_PROBE_SAMPLES 2048
int tvr_scene_set_arith(tvr_scene *scene, int32_t mode);
int tvr_scene_get_arith(const tvr_scene *scene);
int tvr_scene_get_arith_requested(const tvr_scene *scene);
int tvr_scene_validate_arith(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, int32_t white_bg, float eps_T, float tol,
                             void *scratch, size_t scratch_bytes, float *work, size_t work_bytes, float *max_diff_out, int64_t *probe_app_samples_out, void *stream);
int tvr_scene_destroy(tvr_scene *scene);

/* TensorBase.execute over a ray batch (tensorBase.py:476-536, ndc_ray=False; variant 1: REFTensoRF.execute,
 * models/REFTensoRF.py:174-256, without the training-only normal penalty) as called by
 * OctreeRender_trilinear_fast (tensorf-myc/renderer.py:12-27).
 *   rays [n,6] (o,d); jitter [n] or NULL (is_train: one u per ray, tensorBase.py:351-353);
 *   eps_T: stop a ray once transmittance < eps_T (0 = exact, never stop); must be <= weight_thres;
 *   rgb_out [n,3], depth_out [n]; scratch of tvr_render_scratch_bytes(); dense/stats/prof may be NULL.
 * Arithmetic and its range: everything is fp32 except the matrix products of the appearance network (basis 144->27, layers 1 and 2), which
 * run on the matrix cores as THREE fp16 products per fp32 product (each operand = fp16 hi + fp16 lo, hi*hi + hi*lo + lo*hi, fp32
 * accumulation): ~2^-22 relative error per product, i.e. fp32-class results, but operands pass through fp16's exponent range —
 * |weight|, |appearance feature|, |activation| must stay below 65 504 (the conversion saturates there) and parts below 6e-8 are
 * flushed.  Leaving the range is REPORTED, not silent: with the scene's range check on (the default, tvr_scene_set_range_check) a sample
 * whose operands reach 65 504 renders as NaN.  Layer 3, the positional encoding, interpolation, density and compositing are plain fp32.  The shipped scenes and the reference's
 * 0.1 * randn initialisation are far inside this range (tests: |feature| up to ~1100, weights at 1e-4 scale).
 * PIECES (round 6): a call of at least 6 x piece_rays rays (tvr_scene_set_render_pieces; default 30 720, so every call of 184 320 rays or more — smaller calls gain
 * nothing from two or three pieces, measured) is rendered as
 * K = round(n / piece_rays) pieces of consecutive rays (equal sizes, multiples of 512 rays, the last one shorter), piece k on library-owned stream k & 1 in that stream's half of `scratch`; the two streams fork from the caller's
 * stream by an event and are joined back into it by two more before the call returns, so for the caller everything is still ordered on ITS stream (and a capture of
 * the caller's stream captures the fork and join).  Why: the kernels of one piece take the CUs the other piece's kernels leave as they drain, and a march beside a shade
 * kernel uses the chip's power budget better than either alone — the 800x800 bench frame takes 2 - 4 % less time (profiles/r06_split_frame.txt), pixels unchanged BIT
 * FOR BIT (a ray's result does not depend on the batch it arrives in).  What a caller can notice: (a) the march's fault flag NaNs the pixels of the PIECE that raised
 * it, not of the whole call; (b) tvr_profile records one launch set per piece and the kernels of two pieces overlap, so its sums are per-launch durations as a
 * profiler would list them, not a partition of the call's wall time; (c) calls with `dense` are never cut; (d) one tvr_render(_z) at a time per scene from ONE host
 * thread (the scene owns the two streams; calls from different caller streams queue their pieces on the same two).  tvr_scene_set_render_pieces(scene, 0) switches it off. */
int tvr_scene_set_render_pieces(tvr_scene *scene, int32_t piece_rays);   /* 0: off; < 0: the library's default; else >= 16 (pieces are rounded up to multiples of 512 rays, of 16 below 512) */
int tvr_scene_get_render_pieces(const tvr_scene *scene);
size_t tvr_render_scratch_bytes(const tvr_scene *scene, int64_t n_rays, int32_t n_samples);
/* enough for a call WITHOUT `dense`: two pieces' worth when the call is rendered in pieces (1.3 GB instead of 13 GB for an 800x800 x 512 frame), else the same as above */
size_t tvr_render_scratch_bytes_min(const tvr_scene *scene, int64_t n_rays, int32_t n_samples);
int tvr_render(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, int32_t white_bg,
               const float *jitter, float eps_T, float *rgb_out, float *depth_out,
               void *scratch, size_t scratch_bytes, const tvr_dense_out *dense, uint64_t *stats,
               tvr_profile *prof, void *stream);

/* The same with EXPLICIT sample depths z_vals [n,S] (ascending per ray) instead of uniform steps from the box entry: the foreground of
 * NerfPlusPlus.execute (tensorf-myc/models/nerfplusplus.py:272-276), whose sample_ray (:239-269) spaces the samples between `near`
 * and the bounding sphere and perturbs every one.  dists[j] = z[j+1]-z[j], 0 for the last (tensorBase.py:488).
 * t_last_tiny_out [n] or NULL: prod_j (1 - alpha_j + 1e-6), the `bg_lambda` of nerfplusplus.py:277-278. */
int tvr_render_z(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, int32_t white_bg,
                 const float *z_vals, float eps_T, float *rgb_out, float *depth_out, float *t_last_tiny_out,
                 void *scratch, size_t scratch_bytes, const tvr_dense_out *dense, uint64_t *stats, void *stream);

/* TensorVMSplit.compute_densityfeature (tensoRF.py:209-225): xyz_norm [m,3] -> out [m]. */
int tvr_density_feature(tvr_scene *scene, const float *xyz_norm, int64_t m, float *out, size_t out_bytes, void *stream);
/* TensorVMSplit.compute_appfeature (tensoRF.py:228-244): xyz_norm [m,3] -> out [m,app_dim]. */
int tvr_app_feature(tvr_scene *scene, const float *xyz_norm, int64_t m, float *out, size_t out_bytes, void *stream);
/* MLPRender_Fea.execute (tensorBase.py:76-86): viewdirs [m,3], features [m,app_dim] -> rgb [m,3]. */
int tvr_mlp_render(tvr_scene *scene, const float *viewdirs, const float *features, int64_t m, float *rgb, size_t rgb_bytes, void *stream);
/* REFTensoRF.compute_appfeature (models/REFTensoRF.py:107-133), variant-1 scenes: xyz_norm [m,3] -> features [m,app_dim] and
 * extra [m,8] = {normal_vector 3 (not normalised), rgb_d 3, relu(specular_tint), relu(rho)}. */
int tvr_app_feature_ref(tvr_scene *scene, const float *xyz_norm, int64_t m, float *features, size_t features_bytes, float *extra, size_t extra_bytes,
                        void *stream);
/* MLPRender_Fea_Ref.execute (models/REFTensoRF.py:18-28), variant-1 scenes: viewdirs [m,3] (the reflection directions),
 * features [m,app_dim], dot_product [m] -> sigmoid rgb [m,3]. */
int tvr_mlp_render_ref(tvr_scene *scene, const float *viewdirs, const float *features, const float *dot_product, int64_t m,
                       float *rgb, size_t rgb_bytes, void *stream);
/* AlphaGridMask.sample_alpha (tensorBase.py:50-56): xyz [m,3] (world) -> out [m].  Stand-alone (the reference's
 * AlphaGridMask is its own module): volume (gz,gy,gx) fp32, grid (gx,gy,gz), aabb, invgridSize = 1/size*2 (:46). */
int tvr_alpha_sample(const float *alpha_volume_dev, const int32_t agrid_xyz[3], const float alpha_aabb[6],
                     const float alpha_inv_size[3], const float *xyz, int64_t m, float *out, size_t out_bytes, void *stream);

/* ---- training step (caller: tensorf-myc/train.py:225-261) ------------------------------------------------
 * The TensoRF-specific halves of forward and backward are HIP kernels; the 144->27 basis, the positional encoding and the
 * three Linears run as library GEMMs under the host's autograd between them. */

/* Byte offsets of the regions of a tvr_render / tvr_march_forward scratch buffer, for hosts that consume the queue:
 * counter u32[4] = {queue length, the march's tile counter, FAULT flag, training-workspace OVERFLOW flag (tvr_train_forward)}: the flag is non-zero if a wave of the march kernel gave up waiting for its
 * tile number (1: overtaken in the 64-slot ring, 2: spin limit; tvr_march.hip) — tvr_render then writes NaN to every pixel and depth of the call,
 * hosts that consume the queue read it with the queue length; ray_off/ray_cnt u32[n]; acc f32[n]; q_pos float4[cap] {xyz_norm, weight}; q_out float4[cap] {rgb, weight};
 * q_ray u32[cap]; q_j u32[cap]; cap = n_rays * n_samples.  Each ray's entries are contiguous and in sample order.
 * The `counter` region is a 256-byte header the library zeroes at the start of every call; beyond the four words above it holds the kernels' own work counters (word 16: the
 * shade kernels' tile tickets, word 32: the march's ray counter) — hosts must not write it between a call's launches. */
typedef struct { size_t counter, ray_off, ray_cnt, acc, q_pos, q_out, q_ray, q_j, total; } tvr_scratch_layout;
int tvr_scratch_describe(int64_t n_rays, int32_t n_samples, tvr_scratch_layout *out);

/* The march alone (sample_ray .. raw2alpha, tensorBase.py:487-513): fills the queue (q_pos, q_ray, ray_off, ray_cnt, counter),
 * acc [n] and depth_out [n].  Same arithmetic as tvr_render.
 * RAY ORDER: for n_rays <= 65 536 (TVR_RAY_ORDER_MAX_RAYS: every training batch of the reference) the queue is put into ray order behind the march — ray_off
 * ascends with the ray index — so that every reduction over the appearance samples (the weight-gradient products) runs in the same order run to run: the network's
 * gradients of tvr_train_backward are bit-identical run to run for such batches, the VM factors' (fp32 atomic scatter) reproducible to rounding.  The pass uses the
 * scratch's q_out and q_j regions as temporaries: their contents are undefined afterwards.  Larger batches keep the order the march kernel's waves finished in
 * (every gradient reproducible to rounding only).  A march that raised its fault flag leaves the queue as it lies (the pass returns at once). */
int tvr_march_forward(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, const float *jitter, float eps_T,
                      float *depth_out, void *scratch, size_t scratch_bytes, void *stream);

/* Explicit-depth variant (NerfPlusPlus foreground under autograd); t_last_tiny_out [n] or NULL as in tvr_render_z. */
int tvr_march_forward_z(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, const float *z_vals, float eps_T,
                        float *depth_out, float *t_last_tiny_out, void *scratch, size_t scratch_bytes, void *stream);

/* TensorBase.filtering_rays (tensorBase.py:411-441; train.py:196-199, 296) in one pass: mask[i] = 1 if ray i is kept.  bbox_only: the slab test against the
 * scene's aabb (t_max > t_min, zero direction components replaced by 1e-6).  Otherwise: any of the N_samples evaluation-mode samples of sample_ray
 * (entry distance clamped to [near, far], j * stepSize) has alpha > 0 in the scene's alpha mask — all samples are looked up, as the reference does. */
int tvr_filter_rays(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t N_samples, int32_t bbox_only, uint8_t *mask, size_t mask_bytes,
                    void *stream);

/* Gradient outputs in the REFERENCE parameter layout ((1,C,H,W) planes, (1,C,L,1) lines); each call overwrites its six. */
typedef struct { float *density_plane[3], *density_line[3], *app_plane[3], *app_line[3]; } tvr_vm_grads;
size_t tvr_grad_scratch_bytes(const tvr_scene *scene);      /* packed gradient images used internally by the two backward calls */

/* d loss / d density factors from grad_w [queue length] (d loss / d weight of each appearance sample, queue order) and
 * grad_acc [n] (d loss / d acc_map).  fwd_scratch is the untouched scratch of the matching tvr_march_forward call. */
int tvr_march_backward(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, const float *jitter, float eps_T,
                       const void *fwd_scratch, size_t fwd_scratch_bytes, const float *grad_w, const float *grad_acc,
                       void *grad_scratch, size_t grad_scratch_bytes, const tvr_vm_grads *out, void *stream);

/* Explicit-depth variant; t_last_tiny [n] (forward values) with grad_t_last_tiny [n] = d loss / d t_last_tiny, or both NULL. */
int tvr_march_backward_z(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, const float *z_vals, float eps_T,
                         const void *fwd_scratch, size_t fwd_scratch_bytes, const float *grad_w, const float *grad_acc,
                         const float *t_last_tiny, const float *grad_t_last_tiny, void *grad_scratch, size_t grad_scratch_bytes,
                         const tvr_vm_grads *out, void *stream);

/* h [m,144] = bilinear(app_plane) * linear(app_line), plane-major (tensoRF.py:235-241, before basis_mat), and its backward.
 * h is ALWAYS 144 columns wide — the kernels' layout, 3 planes x 48 channels: column 48 p + c is component c of plane p, and the columns
 * behind a plane's own app_n_comp[p] components are written as zeros — whatever the scene's component counts (a zero-padded scene's
 * basis_mat is [27, sum(app_n_comp)]; the host gathers the columns it owns).  h_bytes / dh_bytes >= m * 144 * 4. */
int tvr_app_h_forward(tvr_scene *scene, const float *xyz_norm, int64_t m, float *h_out, size_t h_bytes, void *stream);
int tvr_app_h_backward(tvr_scene *scene, const float *xyz_norm, int64_t m, const float *dh, size_t dh_bytes, void *grad_scratch,
                       size_t grad_scratch_bytes, const tvr_vm_grads *out, void *stream);

/* The appearance network of a TRAINING step (TensorVMSplit; train.py:225-261 through tensoRF.py:244 `basis_mat` and tensorBase.py:76-86
 * `MLPRender_Fea.execute`) on the m appearance samples of a batch, forward and backward, as register-resident MFMA chains.
 *   forward : h [m,144] (tvr_app_h_forward), viewdirs [m,3] -> rgb [m,3]; the inference kernel's own instructions, so a training forward and
 *             an evaluation render agree bit for bit.  Saved for the backward: feats32 [m,32] (27 features + 5 zeros), h1 / h2 [m,128] = the
 *             outputs of layers 1 / 2 after their ReLU.  Uses the scene's packed weights (tvr_scene_update first).
 *   backward: grad_rgb [m,3] (w.r.t. the sigmoid output) -> dh [m,144] (feed tvr_app_h_backward), plus the matrices whose products with
 *             the saved activations are the weight gradients (tvr_gemm_tn): d_out4 [m,4] = {grad of layer 3's output, 0}, dh2 / dh1 [m,128] =
 *             gradients of the pre-ReLU outputs of layers 2 / 1, dfeats32 [m,32] = gradient of the 27 features (+ 5 zeros):
 *               dW3 = d_out^T h2, db3 = colsum(d_out); dW2 = dh2^T h1, db2 = colsum(dh2); dW1 = dh1^T X with X = tvr_pe_concat(feats, viewdirs),
 *               db1 = colsum(dh1); d basis_mat = dfeats^T h.
 *             W1 [128,150], W2 [128,128], W3 [3,128], basis [27,144]: the CURRENT parameters in the reference layout; they are packed into
 *             `image` (tvr_mlp_train_image_bytes() of 256-byte aligned device memory, caller-owned) by the call itself.
 *             gscale_dev: device scalar, a power of two.  The MFMA operands pass through fp16 (see tvr_render), and MSE gradients of a
 *             4096-ray batch are O(1e-5): gradients are multiplied by gscale on entry and by 1 / gscale on exit; choose it so that
 *             max |grad_rgb| * gscale is O(100) (results do not depend on it beyond rounding).
 *             sat_flag_dev (optional device uint32, caller zeroes): set to 1 by the backward kernel if a scaled gradient reached fp16's largest
 *             finite value on its way into a matrix product (the split saturates there silently) — the step's gradients are then clipped and
 *             the caller should lower gscale.
 * All matrices row-major, contiguous, 16-byte aligned; m * 576 < 2^32 per call (both directions).  Every output is passed with its size in
 * bytes: rgb m*3*4, feats32 / dfeats32 m*32*4, h1 / h2 / dh1 / dh2 m*128*4, d_out4 m*4*4, dh m*144*4. */
size_t tvr_mlp_train_image_bytes(void);
int tvr_mlp_train_forward(tvr_scene *scene, const float *h, const float *viewdirs, int64_t m, float *rgb, size_t rgb_bytes, float *feats32,
                          size_t feats32_bytes, float *h1, size_t h1_bytes, float *h2, size_t h2_bytes, void *stream);
int tvr_mlp_train_backward(const float *W1, const float *W2, const float *W3, const float *basis, const float *grad_rgb, const float *rgb, const float *feats32,
                           const float *h1, const float *h2, int64_t m, const float *gscale_dev, float *d_out4, size_t d_out4_bytes, float *dh2,
                           size_t dh2_bytes, float *dh1, size_t dh1_bytes, float *dfeats32, size_t dfeats32_bytes, float *dh, size_t dh_bytes,
                           uint32_t *sat_flag_dev, void *image, size_t image_bytes, void *stream);

/* The same for a REFTensoRF scene (what configs/Scar.txt:28 trains; models/REFTensoRF.py:107-133, 174-256): h -> basis_mat and the four heads
 * {normal, specular tint, diffuse, rho} -> normalised normal, d = -view, dot = d.n, reflection = 2 dot n - d -> MLPRender_Fea_Ref([-dot, features, reflection,
 * PE(features), PE(reflection)]) = rgb_s -> rgb = relu(tint) * rgb_s + rgb_d, in one kernel (the inference kernel's instructions).
 *   forward : additionally saves g8 [m,8] = the raw head outputs {normal 3 (not normalised), tint, rgb_d 3, rho} and rgb_s [m,3] (the network's sigmoid output);
 *             feats32 [m,32] = {27 features, reflection 3, -dot, 0}: the 31 base values of layer 1.  -dot (column 30) is a differentiable output of the host's
 *             autograd function: REFTensoRF's normal penalty sum_i w_i relu(-dot_i)^2 (REFTensoRF.py:236-239) is formed from it.
 *   backward: grad_rgb [m,3] w.r.t. the final colour, grad_in0 [m] or NULL = d loss / d(-dot) arriving through that output -> dh [m,144] (through basis_mat AND the
 *             heads), d_out4 / dh2 / dh1 as above, dfeats32 [m,32] = gradients of {features 27, reflection 3, -dot, 0}, dg8 [m,8] = gradients of the raw head
 *             outputs.  Weight gradients: as above with X = tvr_pe_concat(features, reflection, -dot) [m,151] for dW1, d basis_mat = dfeats32[:, :27]^T h, and
 *             d heads = dg8^T h (rows {normal 3, specular, diffuse 3, rho}), bias gradients = column sums.
 *             W1 [128,151]; heads_W = {normal_linear [3,144], diffuse_linear [3,144], specular_linear [1,144], rho_linear [1,144]} weights (the order of
 *             tvr_scene_params.ref_W). */
int tvr_mlp_train_forward_ref(tvr_scene *scene, const float *h, const float *viewdirs, int64_t m, float *rgb, size_t rgb_bytes, float *feats32, size_t feats32_bytes,
                              float *h1, size_t h1_bytes, float *h2, size_t h2_bytes, float *g8, size_t g8_bytes, float *rgb_s, size_t rgb_s_bytes, void *stream);
int tvr_mlp_train_backward_ref(const float *W1, const float *W2, const float *W3, const float *basis, const float *const heads_W[4], const float *grad_rgb,
                               const float *grad_in0, const float *rgb_s, const float *feats32, const float *h1, const float *h2, const float *g8, const float *viewdirs,
                               int64_t m, const float *gscale_dev, float *d_out4, size_t d_out4_bytes, float *dh2, size_t dh2_bytes, float *dh1, size_t dh1_bytes,
                               float *dfeats32, size_t dfeats32_bytes, float *dg8, size_t dg8_bytes, float *dh, size_t dh_bytes, uint32_t *sat_flag_dev, void *image,
                               size_t image_bytes, void *stream);

/* ---- the training step as two calls with NO host read in between (round 3) -------------------------------------------------------------------
 * tvr_train_forward  = tvr_march_forward -> tvr_app_h_forward -> tvr_mlp_train_forward(_ref) -> the compositing tail (tensorBase.py:520-527);
 * tvr_train_backward = its gradient -> tvr_mlp_train_backward(_ref) -> the weight / bias gradients (tvr_gemm_tn, column sums) -> tvr_app_h_backward ->
 *                      tvr_march_backward.
 * Every kernel behind the march takes the number of appearance samples from the device (the queue counter in the forward scratch) and works in `work`,
 * a caller-owned buffer of tvr_train_work_bytes(app_cap) sized for app_cap appearance samples — so the sequence of launches is fixed and the step can be
 * captured in a hipGraph.  If a step's queue is longer than app_cap, word 3 of the scratch header (tvr_scratch_layout.counter) is set and the step's
 * results are void: the caller reads that word where it reads the loss and repeats with a larger capacity.  The compositing sums run in a fixed order:
 * the step is bit-reproducible (torch's index_add is not).
 *   forward : rgb_map [n,3], depth [n]; pen_ray [n] (variant 1 only, else NULL) = per-ray sum_e w_e relu(-dot_e)^2, whose sum over rays is REFTensoRF's
 *             normal penalty (REFTensoRF.py:236-239).
 *   backward: grad_rgb_map [n,3], grad_pen_ray [n] or NULL -> gradients of every parameter: the VM factors through `vm_out` (all twelve), the network
 *             through `mlp_out` (reference layouts; heads: variant 1 only).  `weights`: the CURRENT parameters (reference layout), packed by the call.
 *             grad_scale_target: see tvr_mlp_train_backward's gscale (64 is the tested default); sat_flag_dev as there.
 * SHAPES (round 6): REFTensoRF scenes: featureC 128, view_pe = fea_pe = 2, 48 appearance components per plane.  TensorVMSplit scenes: EVERY shape the scene accepts —
 * 1 .. 16 density and 1 .. 48 appearance components per plane (zero channels in the packed scene; basis_mat's [27, sum n] columns mapped on the way in and out), featureC
 * 1 .. 128 (units that do not exist are zero rows / columns of the packed weights, their gradients are cropped out of the 128-wide products), view_pe / fea_pe 0 .. 6.
 * TensorBase's own defaults — 8 / 24 components, 6 / 6 frequencies, tensorBase.py:141-145 — are among them.  With more than two frequencies the forward is the lockstep
 * layer-1 kernel of tvr_render and the backward takes dX slot by slot over a streamed W1^T image; for every shape but 2 / 2 at width 128 dW1 is reduced in column blocks
 * of 152 columns (csrc/tvr_mlp_train.hip, tvr_train.hip pe_concat_gen_kernel). */
typedef struct { const float *W1, *W2, *W3, *basis; const float *heads_W[4]; } tvr_train_weights;   /* heads_W: normal, diffuse, specular, rho (variant 1) */
typedef struct { float *W1, *b1, *W2, *b2, *W3, *b3, *basis; float *heads_W[4], *heads_b[4]; } tvr_train_mlp_grads;
size_t tvr_train_work_bytes(const tvr_scene *scene, int64_t n_rays, int32_t n_samples, int64_t app_cap);
/* Byte offsets inside `work` of what a step leaves there (version 141), for hosts that inspect a step — the parity tests check every stage of the backward against
 * fp64 arithmetic on these very tensors.  Saved by the forward: h [cap,144], feats32 [cap,32], h1 / h2 [cap,128] (relu outputs), rgb [cap,3]; written by the backward:
 * grgb [cap,3] (gradient of the per-sample colours), d_out4 [cap,4], dh2 / dh1 [cap,128], dfeats32 [cap,32], dh [cap,144], X (the MLP input: [cap,150] / [cap,151], or for
 * scenes with more than two encoding frequencies x_blocks column blocks of x_block_cols columns, block b a contiguous [cap, w_b] matrix at X + b * cap * x_block_cols
 * floats, w_b = the block's columns rounded up to 4).  Rows beyond the step's queue length (scratch header word 0) are undefined. */
typedef struct { size_t h, feats32, h1, h2, rgb, grgb, d_out4, dh2, dh1, dfeats32, dh, X, total; int32_t x_blocks, x_block_cols; } tvr_train_work_layout;
int tvr_train_work_describe(const tvr_scene *scene, int64_t n_rays, int32_t n_samples, int64_t app_cap, tvr_train_work_layout *out);
int tvr_train_forward(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, const float *jitter, float eps_T, int32_t white_bg,
                      void *fwd_scratch, size_t fwd_scratch_bytes, void *work, size_t work_bytes, int64_t app_cap,
                      float *rgb_map, float *depth, float *pen_ray, void *stream);
int tvr_train_backward(tvr_scene *scene, const float *rays, int64_t n_rays, int32_t n_samples, const float *jitter, float eps_T, int32_t white_bg,
                       const void *fwd_scratch, size_t fwd_scratch_bytes, void *work, size_t work_bytes, int64_t app_cap, const tvr_train_weights *weights,
                       const float *grad_rgb_map, const float *grad_pen_ray, float grad_scale_target, void *grad_scratch, size_t grad_scratch_bytes,
                       const tvr_vm_grads *vm_out, const tvr_train_mlp_grads *mlp_out, uint32_t *sat_flag_dev, void *stream);

/* C [Ka,Kb] = A^T B for tall-skinny fp32 operands A [M,Ka] (row stride lda), B [M,Kb] (row stride ldb): the weight gradients dW = dY^T X of
 * the training step's Linears (MLPRender_Fea's three layers tensorBase.py:69-71, basis_mat tensoRF.py:150) over the M appearance samples of
 * a batch.  fp32 semantics (fp32-input MFMA), fixed summation order (bit-reproducible); C is overwritten;
 * (Ka/32 rounded up) * (Kb/32 rounded up) <= 20.  scratch: tvr_gemm_tn_scratch_bytes() of device memory (per-workgroup partial sums). */
size_t tvr_gemm_tn_scratch_bytes(int32_t Ka, int32_t Kb, int64_t M);
int tvr_gemm_tn(const float *A, int32_t lda, int32_t Ka, const float *B, int32_t ldb, int32_t Kb, int64_t M, float *C,
                void *scratch, size_t scratch_bytes, void *stream);
/* The same product plus colsum_A [Ka] = A^T 1 in the same pass (a virtual ones column of B): the weight and the bias gradient of a Linear from one read of dY.
 * Ka x (Kb + 1) must fit the 20 tiles, Ka + Kb <= 320; scratch: tvr_gemm_tn_scratch_bytes(Ka, Kb + 1, M). */
int tvr_gemm_tn_bias(const float *A, int32_t lda, int32_t Ka, const float *B, int32_t ldb, int32_t Kb, int64_t M, float *C, float *colsum_A,
                     void *scratch, size_t scratch_bytes, void *stream);
/* The same on the fp16-split MFMAs with A * s (s = *scale_dev, a power of two; the result is divided by s again): for operands whose range is known to fit fp16
 * at that scale (a caller-chosen gradient scale).  colsum_A optional.  Ka <= 128, Ka + Kb <= 320. */
int tvr_gemm_tn_scaled(const float *A, int32_t lda, int32_t Ka, const float *B, int32_t ldb, int32_t Kb, int64_t M, float *C, float *colsum_A,
                       const float *scale_dev, void *scratch, size_t scratch_bytes, void *stream);

/* The MLP input of the training step in one pass: X [m,150] = [features 27, viewdirs 3, PE(features), PE(viewdirs)] (MLPRender_Fea.execute,
 * tensorBase.py:76-82; positional_encoding :9-15), or X [m,151] with dot_product [m] in front (MLPRender_Fea_Ref, REFTensoRF.py:19-24) when
 * dot_product is non-NULL; and its backward (grad_viewdirs / grad_dot may be NULL). */
int tvr_pe_concat(const float *features, const float *viewdirs, const float *dot_product, int64_t m, float *X, size_t X_bytes, void *stream);
int tvr_pe_concat_backward(const float *features, const float *viewdirs, const float *grad_X, int64_t m, int32_t with_dot,
                           float *grad_features, size_t grad_features_bytes, float *grad_viewdirs, float *grad_dot, void *stream);

/* TVLoss.forward (tensorf-myc/utils.py:123-142) of one plane x (C,H,W), batch 1: value [1] = weight * 2 (h_tv/count_h + w_tv/count_w) and
 * grad (C,H,W) = d value / d x in the same pass; fixed summation order.  scratch: 2048 bytes. */
int tvr_tv_loss(const float *x, int32_t C, int32_t H, int32_t W, float weight, float *value, float *grad, void *scratch, size_t scratch_bytes,
                void *stream);

/* The two parameter-only regularisers of train.py:237-244 that are not TV, each ONE launch forward and ONE backward, fixed summation order; the
 * backward reads the upstream gradient grad_value[0] from the device.  Up to 8 tensors per call (host arrays of device pointers / sizes).
 *   tvr_l1_mean:    value[0] = sum_t mean |xs[t]|                         TensorVMSplit.density_L1, tensoRF.py:190-194 (3 planes + 3 lines)
 *   tvr_line_ortho: value[0] = sum_t mean |offdiag(V_t V_t^T)|, V_t = vs[t] as (n_comp[t], n_size[t]), 2..48 components
 *                                                                          TensorVMSplit.vector_comp_diffs / vectorDiffs, tensoRF.py:178-188
 * scratch: tvr_l1_mean_scratch_bytes(counts, n) / TVR_LINE_ORTHO_SCRATCH_BYTES (256 since version 141: eight workgroups per line factor, one partial sum each; 32 before). */
#define TVR_LINE_ORTHO_SCRATCH_BYTES 256
size_t tvr_l1_mean_scratch_bytes(const int64_t *counts, int32_t n);
int tvr_l1_mean(const float *const *xs, const int64_t *counts, int32_t n, float *value, void *scratch, size_t scratch_bytes, void *stream);
int tvr_l1_mean_backward(const float *const *xs, float *const *grads, const int64_t *counts, int32_t n, const float *grad_value, void *stream);
int tvr_line_ortho(const float *const *vs, const int32_t *n_comp, const int32_t *n_size, int32_t n, float *value, void *scratch, size_t scratch_bytes,
                   void *stream);
int tvr_line_ortho_backward(const float *const *vs, float *const *grads, const int32_t *n_comp, const int32_t *n_size, int32_t n,
                            const float *grad_value, void *stream);

/* Per-kernel HIP-event timing of tvr_render calls (march / shade / composite), for bench.py's roofline.  max_calls bounds the CALLS recorded; a call rendered in
 * pieces records one set of events per piece (created on first use). */
int tvr_profile_create(int32_t max_calls, tvr_profile **out);
int tvr_profile_reset(tvr_profile *prof);
/* After the stream is synchronised: sums over recorded calls, ms[0..2] = march, shade, composite; returns #calls. */
int tvr_profile_read(tvr_profile *prof, float ms[3]);
int tvr_profile_destroy(tvr_profile *prof);

/* ---- NerfPlusPlus background network: `Embedder` + `MLPNet.forward`, models/nerfplusplus.py:7-56, 66-140, as
 * evaluated by `NerfPlusPlus.execute` on the background samples (:280-302).  W = 128, D base layers with one skip, sigma = |Linear|,
 * rgb = sigmoid(Linear(relu(Linear([base_remap, view embedding])))).  `base_remap` (Linear 128->256) has no activation behind it:
 * the caller folds it into the first rgb layer (rgbh_W_base = W_rgb0[:, :256] @ W_remap, rgbh_b = W_rgb0[:, :256] @ b_remap + b_rgb0). */
typedef struct tvr_mlpnet_desc {
    int32_t D;                  /* base layers (bg_D; 2..4) */
    int32_t W;                  /* 128 (nerfplusplus.py:159) */
    int32_t skip;               /* `skips=[int(bg_D/2)]`: after base layer `skip` the point embedding is concatenated in front */
    int32_t pos_freqs;          /* bg_freq: the 4-vector point gets 4 + 8*pos_freqs inputs */
    int32_t view_freqs;         /* bg_view_freq (2): 3 + 6*view_freqs inputs */
    int32_t samples_per_ray;    /* sample s uses viewdirs[s / samples_per_ray] */
    int32_t arith;              /* TVR_ARITH_* (above) of tvr_mlpnet_forward: 0 = three fp16 products per fp32 product (fp32-class, the default); F16ACT = every layer's inputs
                                 * rounded to fp16, weights hi + lo; F16 = plain fp16 operands.  tvr_mlpnet_train_forward computes fp32-class whatever this says */
} tvr_mlpnet_desc;

typedef struct tvr_mlpnet_params {   /* fp32 device pointers, row-major [out,in] as torch / Jittor Linear */
    const void *base_W[4], *base_b[4];
    const void *sigma_W, *sigma_b;   /* [1,128], [1] */
    const void *rgbh_W_base;         /* [64,128]  (folded, see above) */
    const void *rgbh_W_view;         /* [64,15] = W_rgb0[:, 256:] */
    const void *rgbh_b;              /* [64] */
    const void *rgbo_W, *rgbo_b;     /* [3,64], [3] */
} tvr_mlpnet_params;

size_t tvr_mlpnet_packed_bytes(const tvr_mlpnet_desc *desc);
/* Byte offsets of the regions of a packed network: MFMA fragment blocks, the biases, and the block table tvr_mlpnet_pack writes ONCE and tvr_mlpnet_repack walks on the
 * device every training step (pointers to the parameter tensors + the slice each fragment block takes).  Nothing but pack / repack writes any of it. */
typedef struct { size_t fragments, biases, block_table, block_table_bytes, total; } tvr_mlpnet_layout;
int tvr_mlpnet_describe(const tvr_mlpnet_desc *desc, tvr_mlpnet_layout *out);
/* builds the MFMA fragment image (synchronises the stream once: packing is an explicit, rare call) */
int tvr_mlpnet_pack(const tvr_mlpnet_desc *desc, const tvr_mlpnet_params *params, void *packed, size_t packed_bytes, void *stream);
/* pts [n,4] (inverted-sphere points, depth2pts_outside), viewdirs [ceil(n / samples_per_ray), 3] -> rgb [n,3] (sigmoid applied), sigma [n] (abs applied).
 * `packed` (packed_bytes >= tvr_mlpnet_packed_bytes(), checked) is READ-ONLY to every forward: any number of forwards may share it on different streams.
 * `work`: tvr_mlpnet_work_bytes() of caller-owned device memory, 16-byte aligned — the kernel's per-launch ticket word (dynamic hand-out of the sample tiles), zeroed and
 * advanced by the call: one launch per work buffer at a time.  (Round 5 kept that word in 256 extra bytes of `packed` and wrote it through the const pointer without
 * knowing the buffer's size: version 130.  DESIGN.md 11 has the abort that preceded it.) */
size_t tvr_mlpnet_work_bytes(void);
int tvr_mlpnet_forward(const tvr_mlpnet_desc *desc, const void *packed, size_t packed_bytes, const void *pts, const void *viewdirs, int64_t n_samples, void *rgb,
                       void *sigma, void *work, size_t work_bytes, void *stream);
/* Training (`optimizer.backward(loss)` through MLPNet, train.py:258): the same kernel also saves what the backward needs, all fp32,
 * caller-owned, 16-B aligned, with byte counts that are checked before the launch:
 *   act[l] [n,128] = relu output of base layer l; rgb_hidden [n,64] = relu output of rgb_layers[0]; sigma_pre [n] = the sigma head before `abs`;
 *   embed_pos [n, 4 + 8 pos_freqs] and embed_view [n,16] (15 values + a zero) = the two Embedder outputs as matrices (operands of dW = dY^T E).
 * tvr_mlpnet_repack: the fragment image again from the tensors the table inside `packed` already points at (tvr_mlpnet_pack ran once with the same
 * pointers; their values change every optimizer step) — no host synchronisation. */
typedef struct tvr_mlpnet_saved {
    void *act[4];
    size_t act_bytes;                /* of EACH act[l] */
    void *rgb_hidden;
    size_t rgb_hidden_bytes;
    void *sigma_pre;
    size_t sigma_pre_bytes;
    void *embed_pos;
    size_t embed_pos_bytes;
    void *embed_view;
    size_t embed_view_bytes;
    /* optional (mask_bytes = 0: not written): the ReLU masks of act[l] / rgb_hidden as BITS, [n][2] 64-bit words each — word h of sample s holds at bit
     * 16 b + 4 q + i whether unit 32 b + 8 q + 4 h + i is positive, the order tvr_linear_dx's `mask_bits` takes (16 B per sample instead of 512) */
    void *act_mask[4];
    void *rgb_hidden_mask;
    size_t mask_bytes;               /* of EACH mask buffer: >= n x 16 */
} tvr_mlpnet_saved;
int tvr_mlpnet_train_forward(const tvr_mlpnet_desc *desc, const void *packed, size_t packed_bytes, const void *pts, const void *viewdirs, int64_t n_samples, void *rgb,
                             void *sigma, const tvr_mlpnet_saved *saved, void *work, size_t work_bytes, void *stream);
int tvr_mlpnet_repack(const tvr_mlpnet_desc *desc, const tvr_mlpnet_params *params, void *packed, size_t packed_bytes, void *stream);
/* The input gradient of a Linear over a tall batch with the ReLU mask in front of it fused in (fp32-input MFMAs, W staged in LDS):
 *   dX[m, k] = (sum_{n < N} dY[m, n] W[n, k]) * (mask[m, k] > 0 ? 1 : 0)     (mask NULL: no mask)
 * dY [M, ldy] with N a multiple of 8 in [8,128] (columns n_valid..N-1 of dY must be finite, rows n_valid.. of W are taken as zero), W row-major
 * [n_valid, ldw] = a torch / Jittor Linear weight [out, in] (the reduction runs over its rows), K in {32, 64, 96, 128} columns of W / dX / mask;
 * ldy, ldx, ldm multiples of 4 and all pointers 16-B aligned.  What autograd computes for `relu(Linear(x))` chains (MLPNet.forward, nerfplusplus.py:119-140).
 * scale_dev (optional; N then a multiple of 16): a device scalar s, a power of two — the products run on the fp16-split MFMAs with dY * s (three products,
 * fp32-grade) for callers that know dY's range at that scale, ~3x the rate of the fp32 form; a non-finite result raises *sat_flag_dev (optional).
 * mask_bits (optional, instead of mask): the same mask as bits, [M][2] 64-bit words in the layout tvr_mlpnet_train_forward writes (above). */
int tvr_linear_dx(const float *dY, int32_t ldy, int32_t N, const float *W, int32_t ldw, int32_t n_valid, int32_t K, const float *mask, int32_t ldm,
                  const uint64_t *mask_bits, float *dX, int32_t ldx, size_t dX_bytes, int64_t M, const float *scale_dev, uint32_t *sat_flag_dev, void *stream);
/* out[k] = sum_m A[m, k], k < K <= 128, fixed summation order (the bias gradients of the same Linears).  scratch: tvr_colsum_scratch_bytes(). */
size_t tvr_colsum_scratch_bytes(void);
int tvr_colsum(const float *A, int32_t lda, int32_t K, int64_t M, float *out, void *scratch, size_t scratch_bytes, void *stream);

/* The background of NerfPlusPlus.execute around that network (nerfplusplus.py:280-308).
 * tvr_npp_bg_points: z_lin [n_samples] (the `linspace(0, radii, N)` depths), t_rand [n_rays,n_samples] (the `rand_like` draw of
 *   `perturb_samples`, :196-205) -> pts [n_rays,n_samples,4] = `depth2pts_outside` (:207-237) of the perturbed depths and z
 *   [n_rays,n_samples] = those depths, both already FLIPPED along the sample axis (:296-297).
 * tvr_npp_bg_composite: rgb [n_rays,n_samples,3], sigma, z (flipped order) -> rgb_out [n_rays,3] = sum_k alpha_k T_k rgb_k with
 *   alpha = 1 - exp(-sigma * (z_k - z_{k+1})), last distance 1e10, T = cumprod(1 - alpha + 1e-6) shifted by one (:298-308). */
int tvr_npp_bg_points(const void *rays_o, const void *rays_d, int64_t n_rays, const void *z_lin, int32_t n_samples, const void *t_rand, float radii,
                      void *pts, void *z, void *stream);
int tvr_npp_bg_composite(const void *rgb, const void *sigma, const void *z, int64_t n_rays, int32_t n_samples, void *rgb_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TVR_H */
