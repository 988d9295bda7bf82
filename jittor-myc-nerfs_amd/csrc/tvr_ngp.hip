// tvr_ngp.hip — the alt path: JNeRF Instant-NGP inference on MI355X.  Kernels + the C-ABI of include/tvr_ngp.h.
//
//   ngp_march_kernel          one lane per ray: AABB slab test, jittered start (PCG32), occupancy-bitfield march ONCE, recording each
//                             step's t; an exclusive scan hands out bases IN RAY ORDER; ngp_expand_kernel (one wave per ray) writes the rows.
//   ngp_field_kernel          fused NGPNetworks.execute_: per wave 32 samples at a time; each half-wave gathers the even / odd hash
//                             levels of its 32 samples (8 corners x float2), and the five bias-free Linears run as fp32 MFMAs
//                             (v_mfma_f32_32x32x2_f32) with activations register-resident between layers: the accumulator layout
//                             of one layer IS the B-operand layout of the next once the k index is paired (r, r+4), and that
//                             pairing is folded into the packed weight image (LDS, one ds_read_b32 per MFMA).
//   ngp_composite_kernel      compute_rgbs_inference: one lane per ray over its contiguous rows.
//   ngp_render_kernel         the whole frame in one kernel after the march (tvr_ngp_render): a wave walks a ray's recorded steps 32 at a
//                             time through the same field_tile code and composites in order, stopping at the compositor's T < 1e-4 break.
//   + stand-alone hash / SH encoders (the reference's separate ops) and update_bitfield.
//
// Compiled with -ffp-contract=off; FMAs are explicit where the reference's compiler contracts (see oracle/ngp_oracle.c header).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include "tvr_kernels.h"
#include "tvr_mfma.h"
#include "../../include/tvr_ngp.h"

#define NGP_G TVR_NGP_GRIDSIZE
#define NGP_CELLS (NGP_G * NGP_G * NGP_G)

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return tvr_set_error(TVR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)


// ------------------------------------------------------------------------------------------------ PCG32 (pcg32.h)
struct Pcg {
    uint64_t state, inc;
};
__device__ __forceinline__ uint32_t pcg_next(Pcg &r)
{
    const uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
__device__ __forceinline__ void pcg_advance(Pcg &r, uint64_t delta)
{
    uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = r.inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r.state = acc_mult * r.state + acc_plus;
}
__device__ __forceinline__ float pcg_float(Pcg &r) { return __uint_as_float((pcg_next(r) >> 9) | 0x3f800000u) - 1.0f; }

// ------------------------------------------------------------------------------------------------ march helpers
struct MarchCfg {
    float lo[3], hi[3];
    float near_distance, cone_angle;
    int const_dt;
    uint64_t rng_state, rng_inc;
    uint32_t slab_rays;
};

__device__ __forceinline__ float min_cone_step() { return 1.73205080757f / 1024.0f; }
__device__ __forceinline__ float max_cone_step() { return min_cone_step() * 16.0f * 1024.0f / 128.0f; }
__device__ __forceinline__ float calc_dt(const MarchCfg &c, float t)
{
    if (c.const_dt) return min_cone_step() * 0.5f;
    const float v = t * c.cone_angle;
    return v < min_cone_step() ? min_cone_step() : (max_cone_step() < v ? max_cone_step() : v);
}
__device__ __forceinline__ int frexp_exponent(float v)
{
    int e;
    (void)frexpf(v, &e);
    return e;
}
__device__ __forceinline__ int mip_from_pos(float x, float y, float z)
{
    const float m = fmaxf(fmaxf(fabsf(x - 0.5f), fabsf(y - 0.5f)), fabsf(z - 0.5f));
    return min(TVR_NGP_CASCADES - 1, max(0, frexp_exponent(m) + 1));
}
__device__ __forceinline__ int mip_from_dt(float dt, float x, float y, float z)
{
    const int mip = mip_from_pos(x, y, z);
    dt *= (float)(2 * NGP_G);
    if (dt < 1.f) return mip;
    return min(TVR_NGP_CASCADES - 1, max(frexp_exponent(dt), mip));
}
__host__ __device__ __forceinline__ uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
__host__ __device__ __forceinline__ uint32_t morton3d_invert(uint32_t x)
{
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}
__device__ __forceinline__ bool occupied_at(float x, float y, float z, const uint8_t *__restrict__ bits, uint32_t mip)
{
    const float s = ldexpf(1.0f, -(int)mip);
    const int ix = (int)(((x - 0.5f) * s + 0.5f) * (float)NGP_G);
    const int iy = (int)(((y - 0.5f) * s + 0.5f) * (float)NGP_G);
    const int iz = (int)(((z - 0.5f) * s + 0.5f) * (float)NGP_G);
    const uint32_t idx = morton3d((uint32_t)min(max(ix, 0), NGP_G - 1), (uint32_t)min(max(iy, 0), NGP_G - 1), (uint32_t)min(max(iz, 0), NGP_G - 1));
    return bits[idx / 8 + (uint32_t)NGP_CELLS / 8 * mip] & (1u << (idx % 8));
}
__device__ __forceinline__ float dist_axis(float pos, float dir, float idir, float res)
{
    const float p = res * pos;
    return (floorf(p + 0.5f + 0.5f * copysignf(1.0f, dir)) - p) * idir;
}

// `do { t += dt; } while (t < target);` with a constant dt (advance_to_next_voxel, ray_sampler_header.h:739-753, const_dt mode), in closed
// form and bit for bit.  While t stays inside one binade (ulp u, t = M u with 2^23 <= M < 2^24) every fp32 addition of dt adds the same
// integer q = RN(dt / u) to M, so the loop's j-th value is (M + j q) u and the first one >= target follows from one division.  The
// addition that leaves the binade is done by the hardware (it rounds at the coarser ulp), then the next binade is handled the same
// way.  Ties (dt / u ending in exactly .5: round-to-even then depends on M's parity), non-finite or non-positive operands and
// exponents outside the comfortable range take the plain additions.  Empty space costs a few dozen instructions per voxel instead of
// ~150 dependent additions per coarse voxel.
__device__ __forceinline__ float advance_const(float t, float target, float dt)
{
    t += dt;
    const uint32_t db = __float_as_uint(dt), m_dt = (db & 0x7fffffu) | 0x800000u;
    const int e_dt = (int)((db >> 23) & 0xff);
    while (t < target) {
        const uint32_t tb = __float_as_uint(t), gb = __float_as_uint(target);
        const int e = (int)((tb >> 23) & 0xff), eg = (int)((gb >> 23) & 0xff), s = e - e_dt;
        bool fast = (tb >> 31) == 0 && (gb >> 31) == 0 && e > 0 && e < 0xfe && eg != 0xff && s >= 1 && s <= 22 && e_dt > 0;
        uint32_t q = 0;
        if (fast) {
            const uint32_t rem = m_dt & ((1u << s) - 1u), half = 1u << (s - 1);
            q = (m_dt >> s) + (rem > half ? 1u : 0u);
            fast = rem != half && q > 0;
        }
        if (!fast) {
            t += dt;
            continue;
        }
        const uint32_t M = (tb & 0x7fffffu) | 0x800000u;
        const uint32_t j_stay = (0xffffffu - M) / q;               // most additions that keep M + j q <= 2^24 - 1
        uint32_t j = 0xffffffffu;                                   // target beyond this binade
        if (eg == e) j = (((gb & 0x7fffffu) | 0x800000u) - M + q - 1u) / q;
        if (j <= j_stay) return __uint_as_float((tb & 0xff800000u) | ((M + j * q) & 0x7fffffu));
        t = __uint_as_float((tb & 0xff800000u) | ((M + j_stay * q) & 0x7fffffu));
        t += dt;                                                    // the addition that crosses into the next binade
    }
    return t;
}

struct RayState {
    float o[3], d[3], idir[3];
};

// rays_sampler's march (its two while loops are the same walk): every occupied step's t goes to the ray's slab; returns the step count
__device__ __forceinline__ uint32_t march_ray(const MarchCfg &c, const RayState &r, const uint8_t *__restrict__ bits, float startt, float *__restrict__ tslab)
{
    uint32_t j = 0;
    float t = startt;
    for (;;) {
        const float x = __builtin_fmaf(t, r.d[0], r.o[0]), y = __builtin_fmaf(t, r.d[1], r.o[1]), z = __builtin_fmaf(t, r.d[2], r.o[2]);
        const bool inside = x >= c.lo[0] && x <= c.hi[0] && y >= c.lo[1] && y <= c.hi[1] && z >= c.lo[2] && z <= c.hi[2];
        if (!(inside && j < (uint32_t)TVR_NGP_STEPS)) break;
        const float dt = calc_dt(c, t);
        const uint32_t mip = (uint32_t)mip_from_dt(dt, x, y, z);
        if (occupied_at(x, y, z, bits, mip)) {
            tslab[j] = t;
            ++j;
            t += dt;
        } else {
            const float res = (float)(NGP_G >> mip);
            const float tx = dist_axis(x, r.d[0], r.idir[0], res), ty = dist_axis(y, r.d[1], r.idir[1], res), tz = dist_axis(z, r.d[2], r.idir[2], res);
            const float t_target = t + fmaxf(fminf(fminf(tx, ty), tz) / res, 0.0f);
            if (c.const_dt) {
                t = advance_const(t, t_target, calc_dt(c, t));
            } else {
                do {
                    t += calc_dt(c, t);
                } while (t < t_target);
            }
        }
    }
    return j;
}

__device__ __forceinline__ float ray_entry(const MarchCfg &c, const RayState &r)
{
    float tmin = (c.lo[0] - r.o[0]) / r.d[0], tmax = (c.hi[0] - r.o[0]) / r.d[0], s;
    if (tmin > tmax) { s = tmin; tmin = tmax; tmax = s; }
    float tymin = (c.lo[1] - r.o[1]) / r.d[1], tymax = (c.hi[1] - r.o[1]) / r.d[1];
    if (tymin > tymax) { s = tymin; tymin = tymax; tymax = s; }
    if (tmin > tymax || tymin > tmax) return FLT_MAX;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (c.lo[2] - r.o[2]) / r.d[2], tzmax = (c.hi[2] - r.o[2]) / r.d[2];
    if (tzmin > tzmax) { s = tzmin; tzmin = tzmax; tzmax = s; }
    if (tmin > tzmax || tzmin > tmax) return FLT_MAX;
    if (tzmin > tmin) tmin = tzmin;
    return tmin;
}

__device__ __forceinline__ void load_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, long long i, RayState &r)
{
    for (int k = 0; k < 3; ++k) {
        r.o[k] = rays_o[3 * i + k];
        r.d[k] = rays_d[3 * i + k];
        r.idir[k] = 1.0f / r.d[k];
    }
}

// the march: step count of every ray, and the t of every step in the ray's 1024-float slab
__global__ void __launch_bounds__(256) ngp_march_kernel(MarchCfg c, const float *__restrict__ rays_o, const float *__restrict__ rays_d, long long n_rays,
                                                        const uint8_t *__restrict__ bits, uint32_t *__restrict__ counts, float *__restrict__ tslab)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    RayState r;
    load_ray(rays_o, rays_d, i, r);
    Pcg rng = {c.rng_state, c.rng_inc};
    // N_MAX_RANDOM_SAMPLES_PER_RAY = 8 draws reserved per ray; one 2^32 advance of the global generator per slab of rays
    const uint32_t slab = c.slab_rays ? (uint32_t)(i / c.slab_rays) : 0u, in_slab = c.slab_rays ? (uint32_t)(i % c.slab_rays) : (uint32_t)i;
    pcg_advance(rng, ((uint64_t)slab << 32) + (uint64_t)(uint32_t)(in_slab * 8u));
    float t0 = fmaxf(ray_entry(c, r), c.near_distance);
    t0 = __builtin_fmaf(calc_dt(c, t0), pcg_float(rng), t0);
    counts[i] = march_ray(c, r, bits, t0, tslab + (size_t)i * TVR_NGP_STEPS);
}

// exclusive scan of counts in ray order: (a) per 1024-ray block, (b) the block totals, (c) folded into pass 2
__global__ void __launch_bounds__(1024) ngp_scan_block_kernel(const uint32_t *__restrict__ counts, long long n, uint32_t *__restrict__ local, uint32_t *__restrict__ block_sum)
{
    __shared__ uint32_t wsum[16];
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < n ? counts[i] : 0u;
    uint32_t s = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(s, d, 64);
        if (lane >= d) s += u;
    }
    if (lane == 63) wsum[w] = s;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < w; ++k) before += wsum[k];
    if (i < n) local[i] = before + s - v;
    if (threadIdx.x == 1023) block_sum[blockIdx.x] = before + s;
}
__global__ void __launch_bounds__(1024) ngp_scan_top_kernel(uint32_t *__restrict__ block_sum, int n_blocks, uint32_t *__restrict__ total)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sum[i] : 0u;
        uint32_t s = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(s, d, 64);
            if (lane >= d) s += u;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        uint32_t before = carry;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (i < n_blocks) block_sum[i] = before + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// rows from the recorded t: one wave per ray, lane j writes row base+j (consecutive rows: the stores of a wave cover one contiguous
// span).  Also numsteps and the "got a slab" flag whose scan gives ray_index (rank among rays with a slab; informational, unused by
// inference).
__global__ void __launch_bounds__(256) ngp_expand_kernel(MarchCfg c, const float *__restrict__ rays_o, const float *__restrict__ rays_d, long long n_rays,
                                                         const uint32_t *__restrict__ counts, const float *__restrict__ tslab, const uint32_t *__restrict__ local,
                                                         const uint32_t *__restrict__ block_sum, long long max_samples, float *__restrict__ coords,
                                                         int *__restrict__ numsteps, uint32_t *__restrict__ got_slab)
{
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n_rays) return;
    const uint32_t n = counts[i];
    const uint32_t base = local[i] + block_sum[i >> 10];
    const bool over = (long long)base + n > max_samples;
    if (lane == 0) {
        numsteps[2 * i] = over ? 0 : (int)n;
        numsteps[2 * i + 1] = (int)base;
        if (got_slab) got_slab[i] = over ? 0u : 1u;
    }
    if (over || n == 0) return;
    const float o0 = rays_o[3 * i], o1 = rays_o[3 * i + 1], o2 = rays_o[3 * i + 2], d0 = rays_d[3 * i], d1 = rays_d[3 * i + 1], d2 = rays_d[3 * i + 2];
    const float w0 = (d0 + 1.0f) * 0.5f, w1 = (d1 + 1.0f) * 0.5f, w2 = (d2 + 1.0f) * 0.5f;
    const float max_step = min_cone_step() * 16.0f;
    const float *ts = tslab + (size_t)i * TVR_NGP_STEPS;
    for (uint32_t j = lane; j < n; j += 64) {
        const float t = ts[j];
        const float x = __builtin_fmaf(t, d0, o0), y = __builtin_fmaf(t, d1, o1), z = __builtin_fmaf(t, d2, o2);
        float *q = coords + 7 * ((size_t)base + j);
        q[0] = (x - c.lo[0]) / (c.hi[0] - c.lo[0]);
        q[1] = (y - c.lo[1]) / (c.hi[1] - c.lo[1]);
        q[2] = (z - c.lo[2]) / (c.hi[2] - c.lo[2]);
        q[3] = (calc_dt(c, t) - min_cone_step()) / (max_step - min_cone_step());
        q[4] = w0;
        q[5] = w1;
        q[6] = w2;
    }
}
__global__ void __launch_bounds__(256) ngp_ray_index_kernel(const uint32_t *__restrict__ got_slab, const uint32_t *__restrict__ local, const uint32_t *__restrict__ block_sum,
                                                            const uint32_t *__restrict__ counts, const uint32_t *__restrict__ slab_total,
                                                            const uint32_t *__restrict__ step_total, long long n_rays, int *__restrict__ ray_index,
                                                            uint32_t *__restrict__ counter)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        counter[0] = *slab_total;
        counter[1] = *step_total;
    }
    if (i >= n_rays) return;
    ray_index[i] = (got_slab[i] && counts[i]) ? (int)(local[i] + block_sum[i >> 10]) : -1;
}
__global__ void ngp_counter_kernel(const uint32_t *__restrict__ step_total, uint32_t *__restrict__ counter)
{
    counter[0] = 0u;
    counter[1] = *step_total;
}

// ------------------------------------------------------------------------------------------------ update_bitfield
__global__ void __launch_bounds__(1024) ngp_mean_kernel(const float *__restrict__ grid, float *__restrict__ mean_out)
{
    __shared__ float part[1024];
    float s = 0.f;
    for (uint32_t i = threadIdx.x; i < (uint32_t)NGP_CELLS; i += 1024) s += fmaxf(grid[i], 0.f) / (float)NGP_CELLS;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) *mean_out = part[0];
}
__global__ void __launch_bounds__(256) ngp_grid_to_bits_kernel(const float *__restrict__ grid, uint8_t *__restrict__ bitfield, const float *__restrict__ mean)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)NGP_CELLS / 8 * TVR_NGP_CASCADES) return;
    const float thresh = 0.01f < *mean ? 0.01f : *mean;             // NERF_MIN_OPTICAL_THICKNESS
    uint32_t b = 0;
    for (int j = 0; j < 8; ++j) b |= grid[(size_t)i * 8 + j] > thresh ? (1u << j) : 0u;
    bitfield[i] = (uint8_t)b;
}
__global__ void __launch_bounds__(256) ngp_max_pool_kernel(const uint8_t *__restrict__ prev, uint8_t *__restrict__ next)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)NGP_CELLS / 64) return;
    uint32_t b = 0;
    for (int j = 0; j < 8; ++j) b |= prev[(size_t)i * 8 + j] > 0 ? (1u << j) : 0u;
    const uint32_t x = morton3d_invert(i >> 0) + NGP_G / 8, y = morton3d_invert(i >> 1) + NGP_G / 8, z = morton3d_invert(i >> 2) + NGP_G / 8;
    next[morton3d(x, y, z)] |= (uint8_t)b;
}

// ------------------------------------------------------------------------------------------------ encoders
struct GridCfg {
    uint32_t offsets[TVR_NGP_LEVELS + 1];
    float scale[TVR_NGP_LEVELS];
    uint32_t hashed;                      // bit l: level l uses the spatial hash
};

#ifndef TVR_NGP_DIAG
#define TVR_NGP_DIAG 0            // timing stand-ins for the leave-parts-out table of profiles/r06_ngp_bound.txt (never shipped)
#endif
#ifndef TVR_NGP_DIAG_MASK
#define TVR_NGP_DIAG_MASK 0xFFFFu
#endif
#ifndef TVR_NGP_PAIR              // 1: the x neighbours of a corner pair as one 16-byte load where the table layout allows (encode_level_pair).  Round 5: bit-identical,
#define TVR_NGP_PAIR 0            // and SLOWER — ngp_render_kernel 15.0 - 15.7 ms against 13.0 (profiles/r05_ngp_pair_loads.txt): off
#endif
// grid_index (HashEncode.h:75-93).  With the reference's fixed 2^19 table a level is either dense (res^3 fits: index = x + y*res +
// z*res^2, which can exceed the table by less than its size because corner coordinates reach res) or hashed (table size a power of
// two).  to_grid() checks that every level is one of the two, so the generic stride loop and the modulo reduce to this:
__device__ __forceinline__ uint32_t grid_entry(bool hashed, uint32_t size, uint32_t res, uint32_t x, uint32_t y, uint32_t z)
{
    const uint32_t dense = x + y * res + z * res * res;
    const uint32_t hash = x ^ (y * 19349663u) ^ (z * 83492791u);
    // min(): rows the sampler left unwritten (holes of rays that overflowed max_samples) hold arbitrary bits; never index outside the table
    return hashed ? (hash & (size - 1)) : min(dense - (dense >= size ? size : 0u), size - 1);
}

// one level of kernel_grid for one position: the 8 corner entries are fetched first (independent loads), then blended in corner order
__device__ __forceinline__ float2 encode_level(const float2 *__restrict__ tab, bool hashed, uint32_t size, float scale, float px, float py, float pz)
{
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    const float x = __builtin_fmaf(px, scale, 0.5f), y = __builtin_fmaf(py, scale, 0.5f), z = __builtin_fmaf(pz, scale, 0.5f);
    const float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    const uint32_t cx = (uint32_t)(int)fx0, cy = (uint32_t)(int)fy0, cz = (uint32_t)(int)fz0;
    const float fx = x - fx0, fy = y - fy0, fz = z - fz0;
    float2 v[8];
#pragma unroll
    for (int idx = 0; idx < 8; ++idx)
        v[idx] = tab[grid_entry(hashed, size, res, cx + (idx & 1), cy + ((idx >> 1) & 1), cz + ((idx >> 2) & 1))];
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
        float w = 1.0f;
        w *= (idx & 1) ? fx : 1 - fx;
        w *= (idx & 2) ? fy : 1 - fy;
        w *= (idx & 4) ? fz : 1 - fz;
        r0 = __builtin_fmaf(w, v[idx].x, r0);
        r1 = __builtin_fmaf(w, v[idx].y, r1);
    }
    return make_float2(r0, r1);
}

// encode_level for the fused kernels, with the two x neighbours of a (y, z) corner pair fetched as ONE 16-byte block wherever the table layout has them side by side:
// in a dense level always (entries x and x + 1), in a hashed level whenever cx is even (x + 1 = x | 1: the two hashes differ in bit 0 only — the aligned pair idx & ~1).
// The vector-memory path charges a random gather per lane and line; the 8-byte load of the second neighbour stays in the instruction stream (no branch: the loads of
// INFLIGHT levels must stay in flight together) but lanes that already hold it aim it past the end of the buffer, where the hardware returns zero without a fetch.
// `rs`: the whole table as a raw buffer; `o0`: the level's first entry.  Same entries, same blend order as encode_level: bit-identical features.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2 encode_level_pair(__amdgpu_buffer_rsrc_t rs, uint32_t o0, bool hashed, uint32_t size, float scale, float px, float py, float pz)
{
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    const float x = __builtin_fmaf(px, scale, 0.5f), y = __builtin_fmaf(py, scale, 0.5f), z = __builtin_fmaf(pz, scale, 0.5f);
    const float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    const uint32_t cx = (uint32_t)(int)fx0, cy = (uint32_t)(int)fy0, cz = (uint32_t)(int)fz0;
    const float fx = x - fx0, fy = y - fy0, fz = z - fz0;
    u32x4 q[4];
    u32x2 sv[4];
    bool lo0[4], lo1[4], in1[4];
    const uint32_t hm = hashed ? 0xFFFFFFFFu : 0u;              // the level's kind as a bit mask: both index forms are computed and blended (v_bfi), no branch on a lane-half-dependent flag
    const uint32_t res2 = res * res;
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
        const uint32_t yy = cy + (yz & 1), zz = cz + (yz >> 1);
        const uint32_t hyz = (yy * 19349663u) ^ (zz * 83492791u), dyz = yy * res + zz * res2;
        const uint32_t d0 = cx + dyz, d1 = d0 + 1u;
        const uint32_t e0 = min(d0 - (d0 >= size ? size : 0u), size - 1), e1 = min(d1 - (d1 >= size ? size : 0u), size - 1);      // grid_entry's dense form
        const uint32_t h0 = (cx ^ hyz) & (size - 1), h1 = ((cx + 1u) ^ hyz) & (size - 1);                                          // ... and its hashed form
        const uint32_t i0 = (h0 & hm) | (e0 & ~hm), i1 = (h1 & hm) | (e1 & ~hm);
        const uint32_t bd = i0 + 1u < size ? i0 : i0 - 1u;
        const uint32_t base = ((i0 & ~1u) & hm) | (bd & ~hm);                                 // a 16-byte block inside the level that holds entry i0 (level sizes are even)
        lo0[yz] = i0 == base;
        lo1[yz] = i1 == base;
        in1[yz] = (i1 - base) < 2u;
        q[yz] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((o0 + base) << 3), 0, 0);
        sv[yz] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(in1[yz] ? 0xFFFFFFF8u : ((o0 + i1) << 3)), 0, 0);
    }
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
        const int yz = idx >> 1;
        const u32x4 Q = q[yz];
        float vx, vy;
        if (idx & 1) {
            vx = __uint_as_float(in1[yz] ? (lo1[yz] ? Q.x : Q.z) : sv[yz].x);
            vy = __uint_as_float(in1[yz] ? (lo1[yz] ? Q.y : Q.w) : sv[yz].y);
        } else {
            vx = __uint_as_float(lo0[yz] ? Q.x : Q.z);
            vy = __uint_as_float(lo0[yz] ? Q.y : Q.w);
        }
        float w = 1.0f;
        w *= (idx & 1) ? fx : 1 - fx;
        w *= (idx & 2) ? fy : 1 - fy;
        w *= (idx & 4) ? fz : 1 - fz;
        r0 = __builtin_fmaf(w, vx, r0);
        r1 = __builtin_fmaf(w, vy, r1);
    }
    return make_float2(r0, r1);
}

__global__ void __launch_bounds__(256) ngp_hash_encode_kernel(GridCfg g, const float *__restrict__ grid, const float *__restrict__ pos, int pos_stride, long long n,
                                                              float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n) return;
    const float2 r = encode_level(reinterpret_cast<const float2 *>(grid) + g.offsets[l], (g.hashed >> l) & 1u, g.offsets[l + 1] - g.offsets[l], g.scale[l],
                                  pos[i * pos_stride], pos[i * pos_stride + 1], pos[i * pos_stride + 2]);
    reinterpret_cast<float2 *>(out)[i * TVR_NGP_LEVELS + l] = r;
}

__device__ __forceinline__ void sh16(float dx, float dy, float dz, float *__restrict__ o)
{
    const float x = dx * 2.f - 1.f, y = dy * 2.f - 1.f, z = dz * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
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
__global__ void __launch_bounds__(256) ngp_sh_encode_kernel(const float *__restrict__ dirs, int stride, long long n, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[16];
    sh16(dirs[i * stride], dirs[i * stride + 1], dirs[i * stride + 2], o);
    float4 *q = reinterpret_cast<float4 *>(out + 16 * i);
    for (int k = 0; k < 4; ++k) q[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// ------------------------------------------------------------------------------------------------ fused field kernel
// Two arithmetic variants of the five bias-free Linears, same gather, same register-resident chaining:
//   F16 = true  (default) v_mfma_f32_32x32x16_f16 with every fp32 operand split into fp16 hi + lo and three products per k-step
//                         (hi*lo, lo*hi, hi*hi; error ~2^-22 relative, i.e. fp32-grade) — 72 MFMAs x 32 cycles per 32 samples;
//   F16 = false           v_mfma_f32_32x32x2_f32, exact fp32 — 192 MFMAs x 64 cycles per 32 samples (-DTVR_NGP_MLP_F32=1).
// Chaining: the accumulator of a 32-neuron block holds, in lane (sample = lane%32, h = lane/32), register i, neuron
// (i/4)*8 + 4h + i%4.  A k-step of the next layer reads registers straight from there as its B operand, and the k order that implies
// is folded into the packed weight image (LDS), so no activation ever leaves the registers:
//   fp32 image  [layer][m_block][step][64 lanes] floats; step s pairs k = (ka, kb) for the two lane halves
//   fp16 image  [block][hi|lo][64 lanes] uint4 (8 halves); block = (layer, m_block, k-step of 16); lane half h, element j:
//                 hash features (density0)   t: level 2*(4t + j/2) + h, feature j%2
//                 previous accumulators      t: neuron (t/2)*32 + (2*(t%2) + j/4)*8 + 4h + j%4
//                 rgb0 input                 t=0: density_mlp output (j/4)*8 + 4h + j%4;  t=1: SH 8h + j
#ifndef TVR_NGP_MLP_F32
#define TVR_NGP_MLP_F32 0
#endif
#ifndef TVR_NGP_INFLIGHT          // hash levels per lane whose 8 corner loads are issued together (2 -> 16 float2 loads in flight)
#define TVR_NGP_INFLIGHT 2
#endif
#ifndef TVR_NGP_WAVES             // waves per SIMD the field kernel is compiled for
#define TVR_NGP_WAVES 2
#endif
#ifndef TVR_NGP_XCD               // 1: every XCD (blockIdx % 8) works through its own contiguous eighth of the tiles
#define TVR_NGP_XCD 0
#endif
enum { NGP_L_D0 = 0, NGP_L_D1 = 2 * 16 * 64, NGP_L_C0 = NGP_L_D1 + 32 * 64, NGP_L_C1 = NGP_L_C0 + 2 * 16 * 64, NGP_L_C2 = NGP_L_C1 + 2 * 32 * 64,
       NGP_IMAGE_FLOATS = NGP_L_C2 + 32 * 64 };
// fp16 image: first block of each layer (blocks are m_block-major, then k-step)
enum { NGP_H_D0 = 0, NGP_H_D1 = 4, NGP_H_C0 = 8, NGP_H_C1 = 12, NGP_H_C2 = 20, NGP_H_BLOCKS = 24, NGP_HIMAGE_FLOATS = NGP_H_BLOCKS * 2 * 64 * 4 };
static_assert(NGP_HIMAGE_FLOATS == NGP_IMAGE_FLOATS, "both images are 48 KiB");


__device__ __forceinline__ int acc_row(int i) { return (i / 4) * 8 + i % 4; }

__global__ void __launch_bounds__(256) ngp_pack_f32_kernel(const float *__restrict__ d0, const float *__restrict__ d1, const float *__restrict__ c0, const float *__restrict__ c1,
                                                           const float *__restrict__ c2, float *__restrict__ image)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NGP_IMAGE_FLOATS) return;
    const float *W;
    int off, n_in, n_out, steps, kind;
    if (e < NGP_L_D1) { W = d0; off = NGP_L_D0; n_in = 32; n_out = 64; steps = 16; kind = 0; }
    else if (e < NGP_L_C0) { W = d1; off = NGP_L_D1; n_in = 64; n_out = 16; steps = 32; kind = 1; }
    else if (e < NGP_L_C1) { W = c0; off = NGP_L_C0; n_in = 32; n_out = 64; steps = 16; kind = 2; }
    else if (e < NGP_L_C2) { W = c1; off = NGP_L_C1; n_in = 64; n_out = 64; steps = 32; kind = 1; }
    else { W = c2; off = NGP_L_C2; n_in = 64; n_out = 3; steps = 32; kind = 1; }
    const int r = e - off, l = r & 63, s = (r >> 6) % steps, mb = (r >> 6) / steps;
    int ka, kb;
    if (kind == 0) { ka = 4 * (s >> 1) + (s & 1); kb = ka + 2; }
    else if (kind == 1) { ka = (s >> 4) * 32 + acc_row(s & 15); kb = ka + 4; }
    else if (s < 8) { ka = acc_row(s); kb = ka + 4; }
    else { ka = 16 + 2 * (s - 8); kb = ka + 1; }
    const int row = mb * 32 + (l & 31), k = l < 32 ? ka : kb;
    image[e] = row < n_out ? W[row * n_in + k] : 0.f;
}

// one thread per (block, lane): the 8 weights of that lane's A fragment, split into hi / lo
__global__ void __launch_bounds__(256) ngp_pack_f16_kernel(const float *__restrict__ d0, const float *__restrict__ d1, const float *__restrict__ c0, const float *__restrict__ c1,
                                                           const float *__restrict__ c2, uint4 *__restrict__ image)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NGP_H_BLOCKS * 64) return;
    const int blk = e >> 6, l = e & 63, h = l >> 5;
    const float *W;
    int first, n_in, n_out, steps, kind;
    if (blk < NGP_H_D1) { W = d0; first = NGP_H_D0; n_in = 32; n_out = 64; steps = 2; kind = 0; }
    else if (blk < NGP_H_C0) { W = d1; first = NGP_H_D1; n_in = 64; n_out = 16; steps = 4; kind = 1; }
    else if (blk < NGP_H_C1) { W = c0; first = NGP_H_C0; n_in = 32; n_out = 64; steps = 2; kind = 2; }
    else if (blk < NGP_H_C2) { W = c1; first = NGP_H_C1; n_in = 64; n_out = 64; steps = 4; kind = 1; }
    else { W = c2; first = NGP_H_C2; n_in = 64; n_out = 3; steps = 4; kind = 1; }
    const int mb = (blk - first) / steps, t = (blk - first) % steps, row = mb * 32 + (l & 31);
    float v[8];
    for (int j = 0; j < 8; ++j) {
        int k;
        if (kind == 0) k = 2 * (2 * (4 * t + j / 2) + h) + (j & 1);
        else if (kind == 1) k = (t / 2) * 32 + (2 * (t & 1) + j / 4) * 8 + 4 * h + (j & 3);
        else k = t == 0 ? (j / 4) * 8 + 4 * h + (j & 3) : 16 + 8 * h + j;
        v[j] = row < n_out ? W[row * n_in + k] : 0.f;
    }
    const Frag f = split8(v);
    image[(blk * 2 + 0) * 64 + l] = f.hi;
    image[(blk * 2 + 1) * 64 + l] = f.lo;
}


#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// one k-step (16 inputs) into the two 32-neuron blocks of a 64-wide layer; the two blocks alternate so that no MFMA directly follows
// the one it depends on
__device__ __forceinline__ void step2(const uint4 *__restrict__ w4, int blk0, int blk1, const Frag &b, f32x16 &a0, f32x16 &a1)
{
    const uint4 h0 = w4[(blk0 * 2 + 0) * 64], l0 = w4[(blk0 * 2 + 1) * 64], h1 = w4[(blk1 * 2 + 0) * 64], l1 = w4[(blk1 * 2 + 1) * 64];
    a0 = MFMAH(l0, b.hi, a0);
    a1 = MFMAH(l1, b.hi, a1);
    a0 = MFMAH(h0, b.lo, a0);
    a1 = MFMAH(h1, b.lo, a1);
    a0 = MFMAH(h0, b.hi, a0);
    a1 = MFMAH(h1, b.hi, a1);
}
__device__ __forceinline__ void step1(const uint4 *__restrict__ w4, int blk, const Frag &b, f32x16 &a)
{
    const uint4 h0 = w4[(blk * 2 + 0) * 64], l0 = w4[(blk * 2 + 1) * 64];
    a = MFMAH(l0, b.hi, a);
    a = MFMAH(h0, b.lo, a);
    a = MFMAH(h0, b.hi, a);
}
// B fragment of k-step t (0..3) of a 64-wide activation held in two accumulators
__device__ __forceinline__ Frag relu_frag(const f32x16 &a0, const f32x16 &a1, int t)
{
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = relu_f(t < 2 ? a0[8 * (t & 1) + j] : a1[8 * (t & 1) + j]);
    return split8(v);
}

// One 32-sample column block through the encoders and both networks.  Lane (col = lane%32, hh = lane/32) holds sample col's
// position and direction; returns (rgb raw, density raw) of sample col in the lanes with hh == 0.  `lds_lane` = LDS image + lane.
template <bool F16>
__device__ __forceinline__ float4 field_tile(const GridCfg &g, const float2 *__restrict__ tab, const float *__restrict__ lds_lane, int hh, float px, float py, float pz,
                                             float dx, float dy, float dz)
{
    // ---- GATHER phase: lane half hh takes the hash levels of parity hh (8 levels x 8 corners)
#if TVR_NGP_PAIR
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)(g.offsets[TVR_NGP_LEVELS] << 3), 0x00020000);
#endif
    float f0[8], f1[8], shv[8];
    {
        float sh[16];
        sh16(dx, dy, dz, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) shv[j] = F16 ? (hh ? sh[8 + j] : sh[j]) : (hh ? sh[2 * j + 1] : sh[2 * j]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint32_t o0 = hh ? g.offsets[2 * p + 1] : g.offsets[2 * p];
        const uint32_t o1 = hh ? g.offsets[2 * p + 2] : g.offsets[2 * p + 1];
        const float sc = hh ? g.scale[2 * p + 1] : g.scale[2 * p];
        const bool hashed = (g.hashed >> (2 * p + hh)) & 1u;
#if TVR_NGP_DIAG == 1 || TVR_NGP_DIAG == 3        // timing stand-ins (WRONG pictures): 1 = no table gather at all, 3 = only the levels with (mask >> level) & 1 gathered
        float2 r = make_float2(px * sc, py * sc);
        if (TVR_NGP_DIAG == 3 && ((TVR_NGP_DIAG_MASK >> (2 * p)) & 3u)) {
            const bool mine = (TVR_NGP_DIAG_MASK >> (2 * p + hh)) & 1u;
            if (mine) r = encode_level(tab + o0, hashed, o1 - o0, sc, px, py, pz);
        }
#elif TVR_NGP_PAIR
        const float2 r = encode_level_pair(rs, o0, hashed, o1 - o0, sc, px, py, pz);
#else
        const float2 r = encode_level(tab + o0, hashed, o1 - o0, sc, px, py, pz);
#endif
        f0[p] = r.x;
        f1[p] = r.y;
        if ((p + 1) % TVR_NGP_INFLIGHT == 0) __builtin_amdgcn_sched_barrier(0);      // bounds the loads in flight (registers)
    }
#if TVR_NGP_DIAG == 2                              // timing stand-in (WRONG pictures): the gather alone, no network
    {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) { s0 += f0[p]; s1 += f1[p]; }
        return make_float4(s0, s1, s0 - s1, -4.0f + 1e-3f * (s0 + s1 + shv[0]));
    }
#endif
    // ---- MATRIX phase
    float density_raw;
    f32x16 e = {0};
    if (F16) {
        const uint4 *w4 = reinterpret_cast<const uint4 *>(lds_lane);
        f32x16 a0 = {0}, a1 = {0};
        // density_mlp.0: 32 -> 64
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (j & 1) ? f1[4 * t + j / 2] : f0[4 * t + j / 2];
            step2(w4, NGP_H_D0 + t, NGP_H_D0 + 2 + t, split8(v), a0, a1);
        }
        // density_mlp.2: relu, 64 -> 16
        f32x16 d = {0};
#pragma unroll
        for (int t = 0; t < 4; ++t) step1(w4, NGP_H_D1 + t, relu_frag(a0, a1, t), d);
        density_raw = d[0];
        // rgb_mlp.0: [density(16), SH(16)] -> 64
        a0 = (f32x16){0};
        a1 = (f32x16){0};
        {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = d[j];
            step2(w4, NGP_H_C0 + 0, NGP_H_C0 + 2, split8(v), a0, a1);
            step2(w4, NGP_H_C0 + 1, NGP_H_C0 + 3, split8(shv), a0, a1);
        }
        // rgb_mlp.2: relu, 64 -> 64
        f32x16 c0 = {0}, c1 = {0};
#pragma unroll
        for (int t = 0; t < 4; ++t) step2(w4, NGP_H_C1 + t, NGP_H_C1 + 4 + t, relu_frag(a0, a1, t), c0, c1);
        // rgb_mlp.4: relu, 64 -> 3
#pragma unroll
        for (int t = 0; t < 4; ++t) step1(w4, NGP_H_C2 + t, relu_frag(c0, c1, t), e);
    } else {
        const float *wl = lds_lane;
        f32x16 a0 = {0}, a1 = {0};                              // the two 32-neuron blocks of a 64-wide layer
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float bv = (s & 1) ? f1[s >> 1] : f0[s >> 1];
            a0 = MFMA(wl[NGP_L_D0 + (0 * 16 + s) * 64], bv, a0);
            a1 = MFMA(wl[NGP_L_D0 + (1 * 16 + s) * 64], bv, a1);
        }
        f32x16 d = {0};
#pragma unroll
        for (int s = 0; s < 32; ++s) d = MFMA(wl[NGP_L_D1 + s * 64], relu_f(s < 16 ? a0[s & 15] : a1[s & 15]), d);
        a0 = (f32x16){0};
        a1 = (f32x16){0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float bv = s < 8 ? d[s & 7] : shv[s & 7];
            a0 = MFMA(wl[NGP_L_C0 + (0 * 16 + s) * 64], bv, a0);
            a1 = MFMA(wl[NGP_L_C0 + (1 * 16 + s) * 64], bv, a1);
        }
        density_raw = d[0];
        f32x16 c0 = {0}, c1 = {0};
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float bv = relu_f(s < 16 ? a0[s & 15] : a1[s & 15]);
            c0 = MFMA(wl[NGP_L_C1 + (0 * 32 + s) * 64], bv, c0);
            c1 = MFMA(wl[NGP_L_C1 + (1 * 32 + s) * 64], bv, c1);
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) e = MFMA(wl[NGP_L_C2 + s * 64], relu_f(s < 16 ? c0[s & 15] : c1[s & 15]), e);
    }
    // rows 0..2 of the last layer and row 0 of the density head sit in accumulator regs 0..2 / 0 of the lower half-wave
    return make_float4(e[0], e[1], e[2], density_raw);
}

template <bool F16>
__global__ void __launch_bounds__(256, TVR_NGP_WAVES) ngp_field_kernel(GridCfg g, const float *__restrict__ grid, const float *__restrict__ image,
                                                           const float *__restrict__ positions, int pos_stride, const float *__restrict__ dirs, int dir_stride,
                                                           long long n_max, const uint32_t *__restrict__ n_dev, float4 *__restrict__ out)
{
    __shared__ float lds[NGP_IMAGE_FLOATS];
    for (int e = threadIdx.x; e < NGP_IMAGE_FLOATS / 4; e += 256) reinterpret_cast<float4 *>(lds)[e] = reinterpret_cast<const float4 *>(image)[e];
    __syncthreads();
    long long n = n_max;
    if (n_dev) n = min((long long)*n_dev, n_max);
    if (n <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, col = lane & 31;
    const long long n_tiles = (n + 31) / 32;
    const float2 *__restrict__ tab = reinterpret_cast<const float2 *>(grid);

#if TVR_NGP_XCD
    // blocks are dealt round-robin to the 8 XCDs: XCD x = blockIdx % 8 walks tiles [x*T/8, (x+1)*T/8) so that neighbouring rays share an L2
    const long long per_xcd = (n_tiles + 7) / 8, xcd = blockIdx.x & 7, t_end = min(n_tiles, (xcd + 1) * per_xcd);
    for (long long tile = xcd * per_xcd + (long long)(blockIdx.x >> 3) * 4 + wave; tile < t_end; tile += (long long)(gridDim.x >> 3) * 4) {
#else
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
#endif
        int hh = h, lane_off = lane;
        asm volatile("" : "+v"(hh), "+v"(lane_off));                // opaque per tile: nothing below may be hoisted out of the loop (registers)
        const long long sidx = tile * 32 + col;
        const long long srow = min(sidx, n - 1);
        const float *q = positions + pos_stride * srow, *qd = dirs + dir_stride * srow;
        const float4 r = field_tile<F16>(g, tab, lds + (F16 ? 4 : 1) * lane_off, hh, q[0], q[1], q[2], qd[0], qd[1], qd[2]);
        if (h == 0 && sidx < n) out[sidx] = r;
    }
}

// ------------------------------------------------------------------------------------------------ fused frame kernel
// render_img's picture without rows, network outputs or slabs in between: a wave takes a ray (dynamic queue), walks its recorded
// steps 32 at a time through field_tile, composites them in order, and STOPS at the sample where compute_rgbs_inference breaks
// (T < 1e-4): the samples behind it never contribute, so skipping their gathers and networks changes nothing.
// T is advanced serially in sample order (the same multiplications as the reference's loop, so the break lands on the same
// sample); colour is accumulated per lane and reduced once per ray (a different summation order: ~1e-7).
template <bool F16>
__global__ void __launch_bounds__(256, TVR_NGP_WAVES) ngp_render_kernel(MarchCfg c, GridCfg g, const float *__restrict__ grid, const float *__restrict__ image,
                                                            const float *__restrict__ rays_o, const float *__restrict__ rays_d, long long n_rays,
                                                            const uint32_t *__restrict__ counts, const float *__restrict__ tslab, unsigned long long *__restrict__ queue,
                                                            float bg0, float bg1, float bg2, float *__restrict__ rgb, unsigned long long *__restrict__ evaluated)
{
    __shared__ float lds[NGP_IMAGE_FLOATS];
    for (int e = threadIdx.x; e < NGP_IMAGE_FLOATS / 4; e += 256) reinterpret_cast<float4 *>(lds)[e] = reinterpret_cast<const float4 *>(image)[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
    const float2 *__restrict__ tab = reinterpret_cast<const float2 *>(grid);
    const float max_step = min_cone_step() * 16.0f;
    unsigned long long done_samples = 0, all_samples = 0;
    for (;;) {
        // a ticket covers 64 consecutive rays: one atomic and one coalesced read of their step counts per 64 rays; rays without
        // steps (a third of a typical frame) are finished right here
        unsigned long long ticket = 0;
        if (lane == 0) ticket = atomicAdd(queue, 1ull);
        const long long first = (long long)__shfl(ticket, 0, 64) * 64;
        if (first >= n_rays) break;
#ifdef TVR_NGP_TILE_W                 // experiment: a ticket is an 8x8 pixel tile of a TVR_NGP_TILE_W-wide image instead of 64 consecutive rays
        const long long tk = first / 64, tpr = TVR_NGP_TILE_W / 8;
        const long long mine = ((tk / tpr) * 8 + (lane >> 3)) * TVR_NGP_TILE_W + (tk % tpr) * 8 + (lane & 7);
#else
        const long long mine = first + lane;
#endif
        const uint32_t my_n = mine < n_rays ? counts[mine] : 0u;
        if (mine < n_rays && my_n == 0) {
            rgb[3 * mine] = bg0;
            rgb[3 * mine + 1] = bg1;
            rgb[3 * mine + 2] = bg2;
        }
        unsigned long long todo = __ballot(my_n != 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const long long i = __shfl(mine, src, 64);
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)my_n, src);
            all_samples += n;
            const float o0 = rays_o[3 * i], o1 = rays_o[3 * i + 1], o2 = rays_o[3 * i + 2], d0 = rays_d[3 * i], d1 = rays_d[3 * i + 1], d2 = rays_d[3 * i + 2];
            const float w0 = (d0 + 1.0f) * 0.5f, w1 = (d1 + 1.0f) * 0.5f, w2 = (d2 + 1.0f) * 0.5f;
            const float *ts = tslab + (size_t)i * TVR_NGP_STEPS;
            float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
            bool broke = false;
            for (uint32_t s0 = 0; s0 < n && !broke; s0 += 32) {
                int hh = h, lane_off = lane;
                asm volatile("" : "+v"(hh), "+v"(lane_off));        // opaque per tile (see ngp_field_kernel)
                const uint32_t valid = min(32u, n - s0);
                const float t = ts[min(s0 + (uint32_t)col, n - 1)];
                const float x = __builtin_fmaf(t, d0, o0), y = __builtin_fmaf(t, d1, o1), z = __builtin_fmaf(t, d2, o2);
                const float px = (x - c.lo[0]) / (c.hi[0] - c.lo[0]), py = (y - c.lo[1]) / (c.hi[1] - c.lo[1]), pz = (z - c.lo[2]) / (c.hi[2] - c.lo[2]);
                const float4 r = field_tile<F16>(g, tab, lds + (F16 ? 4 : 1) * lane_off, hh, px, py, pz, w0, w1, w2);
                // the row's dt as compute_rgbs_inference sees it: warped by the sampler, unwarped by the compositor
                const float dtw = (calc_dt(c, t) - min_cone_step()) / (max_step - min_cone_step());
                const float dt = dtw * (max_step - min_cone_step()) + min_cone_step();
                const float alpha = 1.f - __expf(-__expf(r.w) * dt);
                const int oma_bits = __float_as_int(1.f - alpha);
                // T before each sample, multiplied up serially in sample order (the reference's products, so the break lands on the
                // same sample); branch-free: the break is located afterwards
                float Tk = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    Tk = lane == k ? T : Tk;
                    T *= (uint32_t)k < valid ? __int_as_float(__builtin_amdgcn_readlane(oma_bits, k)) : 1.0f;   // lanes past the ray's end hold a repeat of its last step
                }
                const unsigned long long dead = __ballot(h == 0 && (uint32_t)col < valid && Tk < 1e-4f);   // T is non-increasing: a prefix survives
                const uint32_t n_live = dead ? (uint32_t)(__ffsll((long long)dead) - 1) : valid;
                broke = dead != 0;
                if (h == 0 && (uint32_t)col < n_live) {
                    const float w = alpha * Tk;
                    c0 += w * (1.0f / (1.0f + __expf(-r.x)));
                    c1 += w * (1.0f / (1.0f + __expf(-r.y)));
                    c2 += w * (1.0f / (1.0f + __expf(-r.z)));
                }
                done_samples += valid;
            }
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {                     // lanes 0..31 hold the partial sums
                c0 += __shfl_xor(c0, m, 64);
                c1 += __shfl_xor(c1, m, 64);
                c2 += __shfl_xor(c2, m, 64);
            }
            if (lane == 0) {
                if (!broke) { c0 += T * bg0; c1 += T * bg1; c2 += T * bg2; }
                rgb[3 * i] = c0;
                rgb[3 * i + 1] = c1;
                rgb[3 * i + 2] = c2;
            }
        }
    }
    if (evaluated && lane == 0) {
        if (done_samples) atomicAdd(evaluated, done_samples);
        if (all_samples) atomicAdd(evaluated + 1, all_samples);
    }
}

// ------------------------------------------------------------------------------------------------ compositing
__global__ void __launch_bounds__(256) ngp_composite_kernel(const float4 *__restrict__ net_out, const float *__restrict__ coords, const int *__restrict__ numsteps,
                                                            long long n_rays, float bg0, float bg1, float bg2, float *__restrict__ rgb)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    const uint32_t n = (uint32_t)numsteps[2 * i], base = (uint32_t)numsteps[2 * i + 1];
    const float max_step = min_cone_step() * 16.0f;
    float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    uint32_t j = 0;
    for (; j < n; ++j) {
        if (T < 1e-4f) break;
        const float4 o = net_out[(size_t)base + j];
        const float dt = coords[7 * ((size_t)base + j) + 3] * (max_step - min_cone_step()) + min_cone_step();
        const float alpha = 1.f - __expf(-__expf(o.w) * dt);
        const float w = alpha * T;
        c0 += w * (1.0f / (1.0f + __expf(-o.x)));
        c1 += w * (1.0f / (1.0f + __expf(-o.y)));
        c2 += w * (1.0f / (1.0f + __expf(-o.z)));
        T *= (1.f - alpha);
    }
    if (j == n) {
        c0 += T * bg0;
        c1 += T * bg1;
        c2 += T * bg2;
    }
    rgb[3 * i] = c0;
    rgb[3 * i + 1] = c1;
    rgb[3 * i + 2] = c2;
}

// ------------------------------------------------------------------------------------------------ C-ABI
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

static int to_cfg(const tvr_ngp_march_cfg *cfg, MarchCfg &c)
{
    if (!cfg) return tvr_set_error(TVR_ERR_INVALID, "march cfg is NULL");
    for (int k = 0; k < 3; ++k) {
        if (!(cfg->aabb_hi[k] > cfg->aabb_lo[k])) return tvr_set_error(TVR_ERR_INVALID, "aabb_hi <= aabb_lo on axis %d", k);
        c.lo[k] = cfg->aabb_lo[k];
        c.hi[k] = cfg->aabb_hi[k];
    }
    if ((cfg->rng_inc & 1u) == 0) return tvr_set_error(TVR_ERR_INVALID, "rng_inc must be odd (pcg32 stream)");
    c.near_distance = cfg->near_distance;
    c.cone_angle = cfg->cone_angle;
    c.const_dt = cfg->const_dt;
    c.rng_state = cfg->rng_state;
    c.rng_inc = cfg->rng_inc;
    c.slab_rays = cfg->slab_rays;
    return TVR_OK;
}
static int to_grid(const tvr_ngp_grid_cfg *cfg, GridCfg &g)
{
    if (!cfg) return tvr_set_error(TVR_ERR_INVALID, "grid cfg is NULL");
    g.hashed = 0;
    for (int l = 0; l < TVR_NGP_LEVELS; ++l) {
        if (cfg->offsets[l + 1] <= cfg->offsets[l]) return tvr_set_error(TVR_ERR_INVALID, "grid offsets not increasing at level %d", l);
        if (!(cfg->scale[l] > 0.f) || cfg->scale[l] > 65000.f) return tvr_set_error(TVR_ERR_INVALID, "grid scale[%d] out of (0, 65000]", l);
        g.scale[l] = cfg->scale[l];
        // replay grid_index's stride loop (uint32 arithmetic) to classify the level
        const uint32_t size = cfg->offsets[l + 1] - cfg->offsets[l], res = (uint32_t)ceilf(cfg->scale[l]) + 1u;
        uint32_t stride = 1;
        int dims = 0;
        for (; dims < 3 && stride <= size; ++dims) stride *= res;
        if (size < stride) {
            if (size & (size - 1)) return tvr_set_error(TVR_ERR_UNSUPPORTED, "hashed level %d has a table of %u entries (not a power of two)", l, size);
            g.hashed |= 1u << l;
        } else if (dims != 3 || (uint64_t)res * res * res + (uint64_t)res * res + res >= 2ull * size) {
            return tvr_set_error(TVR_ERR_UNSUPPORTED, "level %d (res %u, %u entries) is neither dense nor hashed", l, res, size);
        }
    }
    for (int l = 0; l <= TVR_NGP_LEVELS; ++l) g.offsets[l] = cfg->offsets[l];
    return TVR_OK;
}

extern "C" {

int tvr_ngp_update_bitfield(const void *density_grid, void *bitfield, void *mean_out, void *scratch, size_t scratch_bytes, void *stream)
{
    (void)scratch;
    (void)scratch_bytes;
    if (!density_grid || !bitfield || !mean_out) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_update_bitfield: NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ngp_mean_kernel, dim3(1), dim3(1024), 0, st, static_cast<const float *>(density_grid), static_cast<float *>(mean_out));
    const unsigned nb = NGP_CELLS / 8 * TVR_NGP_CASCADES;
    hipLaunchKernelGGL(ngp_grid_to_bits_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, static_cast<const float *>(density_grid),
                       static_cast<uint8_t *>(bitfield), static_cast<const float *>(mean_out));
    for (int level = 1; level < TVR_NGP_CASCADES; ++level)
        hipLaunchKernelGGL(ngp_max_pool_kernel, dim3((NGP_CELLS / 64 + 255) / 256), dim3(256), 0, st,
                           static_cast<const uint8_t *>(bitfield) + (size_t)NGP_CELLS / 8 * (level - 1),
                           static_cast<uint8_t *>(bitfield) + (size_t)NGP_CELLS / 8 * level);
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

// scratch: counts, local, flags, flag-local [n] each + two block-sum arrays + 2 totals + the t slabs [n, 1024]
static size_t scan_blocks(int64_t n) { return (size_t)((n + 1023) / 1024); }
size_t tvr_ngp_sample_scratch_bytes(int64_t n_rays)
{
    if (n_rays < 0) return 0;
    const size_t n = align_up((size_t)n_rays * 4, 256), nb = align_up(scan_blocks(n_rays) * 4 + 4, 256);
    return 4 * n + 2 * nb + 256 + (size_t)n_rays * TVR_NGP_STEPS * sizeof(float);
}

int tvr_ngp_sample(const tvr_ngp_march_cfg *cfg, const void *rays_o, const void *rays_d, int64_t n_rays, const void *bitfield,
                   void *coords, int64_t max_samples, void *numsteps, void *ray_index, void *counter,
                   void *scratch, size_t scratch_bytes, void *stream)
{
    MarchCfg c;
    if (int rc = to_cfg(cfg, c)) return rc;
    if (n_rays < 0 || max_samples < 0 || max_samples > 0xFFFFFFFFll) return tvr_set_error(TVR_ERR_INVALID, "n_rays / max_samples out of range");
    if (!counter) return tvr_set_error(TVR_ERR_INVALID, "counter is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_rays == 0) {
        HIP_TRY(hipMemsetAsync(counter, 0, 8, st));
        return TVR_OK;
    }
    if (!rays_o || !rays_d || !bitfield || !numsteps || (!coords && max_samples > 0)) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_sample: NULL argument");
    if (n_rays > (1ll << 31)) return tvr_set_error(TVR_ERR_INVALID, "n_rays > 2^31");
    if (scratch_bytes < tvr_ngp_sample_scratch_bytes(n_rays) || !scratch || misaligned(scratch))
        return tvr_set_error(TVR_ERR_SCRATCH, "tvr_ngp_sample: scratch too small or misaligned (%zu < %zu)", scratch_bytes, tvr_ngp_sample_scratch_bytes(n_rays));
    const size_t n = align_up((size_t)n_rays * 4, 256), nbb = align_up(scan_blocks(n_rays) * 4 + 4, 256);
    char *p = static_cast<char *>(scratch);
    uint32_t *counts = reinterpret_cast<uint32_t *>(p);
    uint32_t *local = reinterpret_cast<uint32_t *>(p + n);
    uint32_t *flags = reinterpret_cast<uint32_t *>(p + 2 * n);
    uint32_t *flocal = reinterpret_cast<uint32_t *>(p + 3 * n);
    uint32_t *bsum = reinterpret_cast<uint32_t *>(p + 4 * n);
    uint32_t *fsum = reinterpret_cast<uint32_t *>(p + 4 * n + nbb);
    uint32_t *totals = reinterpret_cast<uint32_t *>(p + 4 * n + 2 * nbb);          // [0] steps, [1] rays with a slab
    float *tslab = reinterpret_cast<float *>(p + 4 * n + 2 * nbb + 256);
    const int nb = (int)scan_blocks(n_rays);
    const dim3 rg((unsigned)((n_rays + 255) / 256)), wg((unsigned)((n_rays + 3) / 4));
    const float *o = static_cast<const float *>(rays_o), *d = static_cast<const float *>(rays_d);
    const uint8_t *bits = static_cast<const uint8_t *>(bitfield);
    hipLaunchKernelGGL(ngp_march_kernel, rg, dim3(256), 0, st, c, o, d, (long long)n_rays, bits, counts, tslab);
    hipLaunchKernelGGL(ngp_scan_block_kernel, dim3(nb), dim3(1024), 0, st, counts, (long long)n_rays, local, bsum);
    hipLaunchKernelGGL(ngp_scan_top_kernel, dim3(1), dim3(1024), 0, st, bsum, nb, totals);
    hipLaunchKernelGGL(ngp_expand_kernel, wg, dim3(256), 0, st, c, o, d, (long long)n_rays, counts, tslab, local, bsum, (long long)max_samples,
                       static_cast<float *>(coords), static_cast<int *>(numsteps), ray_index ? flags : nullptr);
    if (ray_index) {
        hipLaunchKernelGGL(ngp_scan_block_kernel, dim3(nb), dim3(1024), 0, st, flags, (long long)n_rays, flocal, fsum);
        hipLaunchKernelGGL(ngp_scan_top_kernel, dim3(1), dim3(1024), 0, st, fsum, nb, totals + 1);
        hipLaunchKernelGGL(ngp_ray_index_kernel, rg, dim3(256), 0, st, flags, flocal, fsum, counts, totals + 1, totals, (long long)n_rays,
                           static_cast<int *>(ray_index), static_cast<uint32_t *>(counter));
    } else {
        hipLaunchKernelGGL(ngp_counter_kernel, dim3(1), dim3(1), 0, st, totals, static_cast<uint32_t *>(counter));
    }
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

int tvr_ngp_hash_encode(const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *positions, int32_t pos_stride, int64_t n, void *out, void *stream)
{
    GridCfg g;
    if (int rc = to_grid(grid_cfg, g)) return rc;
    if (n < 0 || pos_stride < 3) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_hash_encode: n < 0 or pos_stride < 3");
    if (n == 0) return TVR_OK;
    if (!grid || !positions || !out) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_hash_encode: NULL argument");
    hipLaunchKernelGGL(ngp_hash_encode_kernel, dim3((unsigned)((n + 255) / 256), TVR_NGP_LEVELS), dim3(256), 0, static_cast<hipStream_t>(stream), g,
                       static_cast<const float *>(grid), static_cast<const float *>(positions), (int)pos_stride, (long long)n, static_cast<float *>(out));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

int tvr_ngp_sh_encode(const void *dirs, int32_t dir_stride, int64_t n, void *out, void *stream)
{
    if (n < 0 || dir_stride < 3) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_sh_encode: n < 0 or dir_stride < 3");
    if (n == 0) return TVR_OK;
    if (!dirs || !out || misaligned(out)) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_sh_encode: NULL or misaligned argument");
    hipLaunchKernelGGL(ngp_sh_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float *>(dirs),
                       (int)dir_stride, (long long)n, static_cast<float *>(out));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

size_t tvr_ngp_net_packed_bytes(void) { return 2 * (size_t)NGP_IMAGE_FLOATS * sizeof(float); }      // fp32 image, then fp16 hi/lo image

int tvr_ngp_net_pack(const tvr_ngp_net_params *p, void *packed, size_t packed_bytes, void *stream)
{
    if (!p || !p->density0 || !p->density1 || !p->rgb0 || !p->rgb1 || !p->rgb2) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_net_pack: NULL weights");
    if (!packed || misaligned(packed) || packed_bytes < tvr_ngp_net_packed_bytes()) return tvr_set_error(TVR_ERR_SCRATCH, "tvr_ngp_net_pack: packed buffer too small or misaligned");
    const float *d0 = static_cast<const float *>(p->density0), *d1 = static_cast<const float *>(p->density1), *c0 = static_cast<const float *>(p->rgb0),
                *c1 = static_cast<const float *>(p->rgb1), *c2 = static_cast<const float *>(p->rgb2);
    hipLaunchKernelGGL(ngp_pack_f32_kernel, dim3((NGP_IMAGE_FLOATS + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), d0, d1, c0, c1, c2,
                       static_cast<float *>(packed));
    hipLaunchKernelGGL(ngp_pack_f16_kernel, dim3((NGP_H_BLOCKS * 64 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), d0, d1, c0, c1, c2,
                       reinterpret_cast<uint4 *>(static_cast<float *>(packed) + NGP_IMAGE_FLOATS));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

int tvr_ngp_network(const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *positions, int32_t pos_stride,
                    const void *dirs, int32_t dir_stride, int64_t n_max, const void *n_dev, void *out, void *stream)
{
    GridCfg g;
    if (int rc = to_grid(grid_cfg, g)) return rc;
    if (n_max < 0 || pos_stride < 3 || dir_stride < 3) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_network: n_max < 0 or a stride < 3");
    if (n_max == 0) return TVR_OK;
    if (!grid || !net_packed || !positions || !dirs || !out || misaligned(out) || misaligned(net_packed))
        return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_network: NULL or misaligned argument");
    const long long tiles = (n_max + 31) / 32;
    unsigned blocks = (unsigned)(tiles < 4 * 2048 ? (tiles + 3) / 4 : 2048);
    if (TVR_NGP_XCD) blocks = (blocks + 7) / 8 * 8;
    const bool f16 = !TVR_NGP_MLP_F32;
    hipLaunchKernelGGL(f16 ? ngp_field_kernel<true> : ngp_field_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g,
                       static_cast<const float *>(grid), static_cast<const float *>(net_packed) + (f16 ? NGP_IMAGE_FLOATS : 0), static_cast<const float *>(positions),
                       (int)pos_stride, static_cast<const float *>(dirs), (int)dir_stride, (long long)n_max, static_cast<const uint32_t *>(n_dev),
                       static_cast<float4 *>(out));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

int tvr_ngp_composite(const void *net_out, const void *coords, const void *numsteps, int64_t n_rays, const float background[3], void *rgb, void *stream)
{
    if (n_rays < 0 || !background) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_composite: n_rays < 0 or background NULL");
    if (n_rays == 0) return TVR_OK;
    if (!numsteps || !rgb) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_composite: NULL argument");
    hipLaunchKernelGGL(ngp_composite_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float4 *>(net_out),
                       static_cast<const float *>(coords), static_cast<const int *>(numsteps), (long long)n_rays, background[0], background[1], background[2],
                       static_cast<float *>(rgb));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}

// scratch of the fused frame call: step counts [n], the ray queue + 2 statistics words, the t slabs [n, 1024]
size_t tvr_ngp_render_scratch_bytes(int64_t n_rays)
{
    if (n_rays < 0) return 0;
    return align_up((size_t)n_rays * 4, 256) + 256 + (size_t)n_rays * TVR_NGP_STEPS * sizeof(float);
}

}  // extern "C" — closed here: the body shared by tvr_ngp_render / tvr_ngp_render_profiled has internal linkage

static int ngp_render_impl(const tvr_ngp_march_cfg *cfg, const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *rays_o,
                           const void *rays_d, int64_t n_rays, const void *bitfield, const float background[3], void *rgb, void *stats,
                           void *scratch, size_t scratch_bytes, void *stream, hipEvent_t *ev)
{
    MarchCfg c;
    GridCfg g;
    if (int rc = to_cfg(cfg, c)) return rc;
    if (int rc = to_grid(grid_cfg, g)) return rc;
    if (n_rays < 0 || n_rays > (1ll << 31) || !background) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_render: n_rays out of range or background NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stats) HIP_TRY(hipMemsetAsync(stats, 0, 16, st));
    if (n_rays == 0) return TVR_OK;
    if (!grid || !net_packed || !rays_o || !rays_d || !bitfield || !rgb || misaligned(net_packed)) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_render: NULL or misaligned argument");
    if (scratch_bytes < tvr_ngp_render_scratch_bytes(n_rays) || !scratch || misaligned(scratch))
        return tvr_set_error(TVR_ERR_SCRATCH, "tvr_ngp_render: scratch too small or misaligned (%zu < %zu)", scratch_bytes, tvr_ngp_render_scratch_bytes(n_rays));
    char *p = static_cast<char *>(scratch);
    const size_t n = align_up((size_t)n_rays * 4, 256);
    uint32_t *counts = reinterpret_cast<uint32_t *>(p);
    unsigned long long *queue = reinterpret_cast<unsigned long long *>(p + n);
    float *tslab = reinterpret_cast<float *>(p + n + 256);
    HIP_TRY(hipMemsetAsync(queue, 0, 8, st));
    const float *o = static_cast<const float *>(rays_o), *d = static_cast<const float *>(rays_d);
    if (ev) HIP_TRY(hipEventRecord(ev[0], st));
    hipLaunchKernelGGL(ngp_march_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, c, o, d, (long long)n_rays, static_cast<const uint8_t *>(bitfield), counts, tslab);
    if (ev) HIP_TRY(hipEventRecord(ev[1], st));
    const bool f16 = !TVR_NGP_MLP_F32;
    const unsigned blocks = (unsigned)(n_rays < 4 * 512 ? (n_rays + 3) / 4 : 512);         // 256 CUs x 2 blocks (LDS image + 2 waves / SIMD)
    hipLaunchKernelGGL(f16 ? ngp_render_kernel<true> : ngp_render_kernel<false>, dim3(blocks), dim3(256), 0, st, c, g, static_cast<const float *>(grid),
                       static_cast<const float *>(net_packed) + (f16 ? NGP_IMAGE_FLOATS : 0), o, d, (long long)n_rays, counts, tslab, queue, background[0],
                       background[1], background[2], static_cast<float *>(rgb), static_cast<unsigned long long *>(stats));
    if (ev) HIP_TRY(hipEventRecord(ev[2], st));
    HIP_TRY(hipGetLastError());
    return TVR_OK;
}


extern "C" {

int tvr_ngp_render(const tvr_ngp_march_cfg *cfg, const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *rays_o,
                   const void *rays_d, int64_t n_rays, const void *bitfield, const float background[3], void *rgb, void *stats,
                   void *scratch, size_t scratch_bytes, void *stream)
{
    return ngp_render_impl(cfg, grid_cfg, grid, net_packed, rays_o, rays_d, n_rays, bitfield, background, rgb, stats, scratch, scratch_bytes, stream, nullptr);
}

int tvr_ngp_render_profiled(const tvr_ngp_march_cfg *cfg, const tvr_ngp_grid_cfg *grid_cfg, const void *grid, const void *net_packed, const void *rays_o,
                            const void *rays_d, int64_t n_rays, const void *bitfield, const float background[3], void *rgb, void *stats,
                            void *scratch, size_t scratch_bytes, void *stream, float ms_out[2])
{
    if (!ms_out) return tvr_set_error(TVR_ERR_INVALID, "tvr_ngp_render_profiled: ms_out is NULL");
    hipEvent_t ev[3];
    for (int k = 0; k < 3; ++k) HIP_TRY(hipEventCreate(&ev[k]));
    int rc = ngp_render_impl(cfg, grid_cfg, grid, net_packed, rays_o, rays_d, n_rays, bitfield, background, rgb, stats, scratch, scratch_bytes, stream, ev);
    ms_out[0] = ms_out[1] = 0.f;
    if (rc == TVR_OK && n_rays > 0) {
        hipError_t e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_out[0], ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_out[1], ev[1], ev[2]);
        if (e != hipSuccess) rc = tvr_set_error(TVR_ERR_HIP, "tvr_ngp_render_profiled: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < 3; ++k) (void)hipEventDestroy(ev[k]);
    return rc;
}

}  // extern "C"
