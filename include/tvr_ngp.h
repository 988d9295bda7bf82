/* tvr_ngp.h — C-ABI of the alt path (BASELINE configs[4]) in libtvr.so: JNeRF Instant-NGP inference
 * (occupancy-bitfield ray march -> multiresolution hash grid + SH-16 -> density / colour networks -> compositing) on MI355X.
 *
 * The reference reaches these steps through Jittor's inline-op API (`jt.code(..., cuda_src=...)`, SURVEY.md §8b "Config-5
 * boundary"); the entry points below are what a binding for that path would call instead.  Conventions are tvr.h's: plain
 * pointers and sizes, `int` return (0 ok, negative tvr_status, text via tvr_last_error()), caller-owned DEVICE buffers
 * (fp32 / int32 / uint8, contiguous, 16-byte aligned), all work enqueued on the caller's hipStream_t (passed as void*),
 * no allocation and no synchronisation inside.  File:line references are under jnerf-myc/python/jnerf/.
 */
#ifndef TVR_NGP_H
#define TVR_NGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVR_NGP_GRIDSIZE 128      /* density_grid_sampler.py:34 NERF_GRIDSIZE */
#define TVR_NGP_CASCADES 5        /* :33 NERF_CASCADES */
#define TVR_NGP_STEPS 1024        /* :37 MAX_STEP */
#define TVR_NGP_LEVELS 16         /* hash_encoder.py:18-19: n_levels 16, 2 features per level */
#define TVR_NGP_BITFIELD_BYTES (TVR_NGP_GRIDSIZE * TVR_NGP_GRIDSIZE * TVR_NGP_GRIDSIZE * TVR_NGP_CASCADES / 8)

/* Sampling configuration = what RaySampler / DensityGridSampler bake into their kernel source
 * (ray_sampler.py:55-58, density_grid_sampler.py:94-113). */
typedef struct tvr_ngp_march_cfg {
    float aabb_lo[3], aabb_hi[3];   /* dataset.py:214-215: 0.5 -+ aabb_scale/2 */
    float near_distance;            /* cfg near_distance */
    float cone_angle;               /* cfg cone_angle_constant */
    int32_t const_dt;               /* cfg const_dt: 1 = dt is MIN_CONE_STEPSIZE/2 everywhere */
    uint64_t rng_state, rng_inc;    /* the process-global pcg32 (`global_vars.py:16`) AS OF THIS CALL; the caller advances it by 2^32
                                       per slab afterwards (`ray_sampler.py:61`) */
    uint32_t slab_rays;             /* 0: one slab (the reference's call: ray i draws from state advanced by 8*i).  k > 0: the call covers
                                       several of the reference's k-ray slabs (`runner.py:209-222`, k = n_rays_per_batch): ray i draws what
                                       it would in slab i/k, i.e. from state advanced by (i/k)*2^32 + 8*(i%k) */
} tvr_ngp_march_cfg;

/* Hash-grid level table (grid_encode.py:24-39; kernel_grid's scale, HashEncode.h:142), evaluated by the host. */
typedef struct tvr_ngp_grid_cfg {
    uint32_t offsets[TVR_NGP_LEVELS + 1];   /* first ENTRY (2 floats) of each level; offsets[16] = total entries */
    float scale[TVR_NGP_LEVELS];            /* exp2(level*log2(per_level_scale))*base_resolution - 1 */
} tvr_ngp_grid_cfg;

/* NGPNetworks, plain-Linear branch (ngp_network.py:60-68): bias-free, row-major [out,in] fp32 device pointers. */
typedef struct tvr_ngp_net_params {
    const void *density0;   /* [64,32] */
    const void *density1;   /* [16,64] */
    const void *rgb0;       /* [64,32]  input = [density_mlp output (16), SH (16)] */
    const void *rgb1;       /* [64,64] */
    const void *rgb2;       /* [3,64]  */
} tvr_ngp_net_params;

/* update_bitfield (update_bitfield.py:14-31, op_header/update_bitfield.h:23-70): density_grid [5*128^3] fp32 (Morton order per
 * cascade) -> bitfield [TVR_NGP_BITFIELD_BYTES] and mean_out[1] = mean(max(cascade-0 density, 0)).  scratch: 4 KiB. */
int tvr_ngp_update_bitfield(const void *density_grid, void *bitfield, void *mean_out, void *scratch, size_t scratch_bytes, void *stream);

/* rays_sampler (ray_sampler.py:20-72, op_header/ray_sampler.h:4-114).  rays_o / rays_d [n_rays,3].
 * Outputs: coords [max_samples,7] = (warped pos, warped dt, warped dir) — rows [0,total) are written, the rest untouched;
 * numsteps [n_rays,2] int32 = (steps, base); ray_index [n_rays] int32 (rank among rays that received a slab, -1 without steps);
 * counter [2] uint32 = (rays with a slab, total steps).  Bases are the exclusive prefix sum of the step counts IN RAY ORDER
 * (the reference draws them from an atomicAdd in arrival order; per-ray contents are identical, the layout here is
 * deterministic).  A ray whose slab would cross max_samples gets (0, base) as in the reference.
 * scratch: tvr_ngp_sample_scratch_bytes(n_rays). */
size_t tvr_ngp_sample_scratch_bytes(int64_t n_rays);
int tvr_ngp_sample(const tvr_ngp_march_cfg *cfg, const void *rays_o, const void *rays_d, int64_t n_rays, const void *bitfield,
                   void *coords, int64_t max_samples, void *numsteps, void *ray_index, void *counter,
                   void *scratch, size_t scratch_bytes, void *stream);

/* HashEncoder.execute (hash_encoder.py:26-30; extract_position + kernel_grid + transpose, HashEncode.h:36-50,117-199,254-268):
 * positions [n, pos_stride floats] (first 3 used) in [0,1] -> out [n,32]. */
int tvr_ngp_hash_encode(const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *positions, int32_t pos_stride, int64_t n,
                        void *out, void *stream);

/* SHEncoder.execute (sh_encoder.py:26-53, SphericalEncode.h:44-100): directions warped to [0,1], [n, dir_stride floats]
 * -> out [n,16]. */
int tvr_ngp_sh_encode(const void *dirs, int32_t dir_stride, int64_t n, void *out, void *stream);

/* Packed MFMA image of the five weight matrices (refresh after the weights change). */
size_t tvr_ngp_net_packed_bytes(void);
int tvr_ngp_net_pack(const tvr_ngp_net_params *params, void *packed, size_t packed_bytes, void *stream);

/* NGPNetworks.execute_ (ngp_network.py:78-85) fused: positions [n, pos_stride floats] in [0,1], directions warped to [0,1]
 * [n, dir_stride floats] (the sampler's rows: coords, stride 7 and coords+4, stride 7) -> out [n,4] = (rgb raw, density raw).
 * n is read from n_dev[0] (uint32, device; e.g. counter+1 of tvr_ngp_sample) when n_dev != NULL, clamped to n_max;
 * otherwise n = n_max. */
int tvr_ngp_network(const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *positions, int32_t pos_stride,
                    const void *dirs, int32_t dir_stride, int64_t n_max, const void *n_dev, void *out, void *stream);

/* CalcRgb.inference (calc_rgb.py:118-150 -> compute_rgbs_inference_fp32, declared calc_rgb.h:45-60; rgb logistic, density exp):
 * net_out [n,4], coords [n,7], numsteps [n_rays,2] -> rgb [n_rays,3]. */
int tvr_ngp_composite(const void *net_out, const void *coords, const void *numsteps, int64_t n_rays, const float background[3],
                      void *rgb, void *stream);

/* One frame = what Runner.render_img (runner.py:195-228) returns for these rays, without rows, network outputs or slabs in between:
 * the march records each step's t, then one kernel walks every ray's steps 32 at a time through the encoders and networks and
 * composites them in order, stopping at the sample where compute_rgbs_inference breaks (T < 1e-4) — the samples behind it
 * contribute nothing, so their gathers and networks are skipped.  cfg->slab_rays = n_rays_per_batch reproduces the slab loop's
 * jitter.  rgb [n_rays,3]; stats (optional, device) uint64[2] = (samples evaluated, samples marched).
 * scratch: tvr_ngp_render_scratch_bytes(n_rays). */
size_t tvr_ngp_render_scratch_bytes(int64_t n_rays);
int tvr_ngp_render(const tvr_ngp_march_cfg *cfg, const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *rays_o,
                   const void *rays_d, int64_t n_rays, const void *bitfield, const float background[3], void *rgb, void *stats,
                   void *scratch, size_t scratch_bytes, void *stream);

/* tvr_ngp_render with HIP events around its two kernels, recorded on `stream`; WAITS for completion (a measurement entry point, unlike
 * everything else here) and returns ms_out = (march ms, render-kernel ms). */
int tvr_ngp_render_profiled(const tvr_ngp_march_cfg *cfg, const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *rays_o,
                            const void *rays_d, int64_t n_rays, const void *bitfield, const float background[3], void *rgb, void *stats,
                            void *scratch, size_t scratch_bytes, void *stream, float ms_out[2]);

#ifdef __cplusplus
}
#endif
#endif
