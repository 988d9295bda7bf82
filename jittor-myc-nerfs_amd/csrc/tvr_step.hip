// tvr_step.hip — the small kernels that make a TRAINING step (tensorf-myc/train.py:225-261) run without a host read of the queue length:
// every kernel behind the march takes the number of appearance samples from the device (the queue counter the march kernel leaves) and its
// buffers at a host-chosen capacity, so the step is a fixed sequence of launches (hipGraph-capturable; tvr_train_forward / tvr_train_backward,
// tvr_api.hip).  Here: the compositing tail of TensorBase.execute (models/tensorBase.py:520-527) forward and backward over the queue, the
// power-of-two gradient scale of the fused MLP backward chosen on the device, and deterministic column sums (bias gradients).
//
//   forward  rgb_map[r] = clamp(sum_e w_e rgb_e + [white_bg] (1 - acc_r), 0, 1)          e over the ray's contiguous queue segment, fixed order
//            pen_ray[r] = sum_e w_e relu(in0_e)^2                                          REFTensoRF's normal penalty (REFTensoRF.py:236-239), per ray
//   backward grgb_e = w_e gm_r,  grad_w_e = rgb_e . gm_r (+ g_pen_r relu(in0_e)^2),  gin0_e = 2 g_pen_r w_e relu(in0_e),  grad_acc_r = -[white_bg] sum_c gm_r,c
//            with gm_r = grad_rgb_map_r where the pre-clamp value lies in [0,1], else 0 (torch.clamp's gradient)
#include "tvr_device.h"
#include "tvr_kernels.h"

#define CT_LANES 8

__global__ __launch_bounds__(256) void composite_train_forward_kernel(const MarchOut mo, const int n_rays, const long long cap, const int white_bg,
                                                                      const float *__restrict__ rgb, const float *__restrict__ feats32, const int with_pen,
                                                                      float *__restrict__ rgb_map, float *__restrict__ pre, float *__restrict__ pen_ray)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = t / CT_LANES, l = t % CT_LANES;
    const bool live = r < n_rays;
    const unsigned base = live ? mo.ray_off[r] : 0u, cnt = live ? mo.ray_cnt[r] : 0u;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, pn = 0.f;
    bool over = false;
    for (unsigned i = l; i < cnt; i += CT_LANES) {
        const long long e = (long long)base + i;
        if (e >= cap) { over = true; break; }              // the appearance workspace is smaller than this step's queue: flagged, the step is void
        const float w = mo.q_pos[e].w;
        c0 = c0 + w * rgb[e * 3];
        c1 = c1 + w * rgb[e * 3 + 1];
        c2 = c2 + w * rgb[e * 3 + 2];
        if (with_pen) {
            const float q = fmaxf(feats32[e * 32 + 30], 0.0f);
            pn = pn + w * (q * q);
        }
    }
#pragma unroll
    for (int off = 1; off < CT_LANES; off <<= 1) {
        c0 = c0 + __shfl_xor(c0, off);
        c1 = c1 + __shfl_xor(c1, off);
        c2 = c2 + __shfl_xor(c2, off);
        pn = pn + __shfl_xor(pn, off);
    }
    if (over) atomicOr(mo.counter + 3, 1u);
    if (!live || l != 0) return;
    // A VOID step is loud on the device (round 4): the appearance queue outgrew the workspace (counter[0] > cap: entries beyond it were never shaded) or
    // the march raised its fault flag.  Every pixel of the batch — not only the rays that hit the cap — is NaN, so the loss is NaN for whoever reads it; the
    // backward (composite_train_backward, march_backward) writes exact zeros, so an optimizer that does not consult model.training_fault_flag() moves nothing.
    if ((long long)*mo.counter > cap || mo.counter[2] != 0u) {
        const float q = __builtin_nanf("");
        pre[(size_t)r * 3] = q; pre[(size_t)r * 3 + 1] = q; pre[(size_t)r * 3 + 2] = q;
        rgb_map[(size_t)r * 3] = q; rgb_map[(size_t)r * 3 + 1] = q; rgb_map[(size_t)r * 3 + 2] = q;
        if (with_pen) pen_ray[r] = q;
        return;
    }
    if (white_bg) {
        const float bg = 1.0f - mo.acc[r];
        c0 = c0 + bg; c1 = c1 + bg; c2 = c2 + bg;
    }
    pre[(size_t)r * 3] = c0; pre[(size_t)r * 3 + 1] = c1; pre[(size_t)r * 3 + 2] = c2;
    rgb_map[(size_t)r * 3] = clamp01(c0);
    rgb_map[(size_t)r * 3 + 1] = clamp01(c1);
    rgb_map[(size_t)r * 3 + 2] = clamp01(c2);
    if (with_pen) pen_ray[r] = pn;
}

// one thread per queue entry (launch sized for the capacity); amax_bits: max over entries of |gradient entering the network| as a uint (floats >= 0 order as
// their bit patterns), the basis of the power-of-two scale of the fused backward
__global__ __launch_bounds__(256) void composite_train_backward_kernel(const MarchOut mo, const long long cap, const float *__restrict__ rgb, const float *__restrict__ feats32,
                                                                       const float *__restrict__ g8, const float *__restrict__ pre, const float *__restrict__ g_map,
                                                                       const float *__restrict__ g_pen, float *__restrict__ grgb, float *__restrict__ gin0,
                                                                       float *__restrict__ grad_w, unsigned *__restrict__ amax_bits)
{
    const long long mq = (long long)*mo.counter, m = mq < cap ? mq : cap;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.0f;
    const bool void_step = mq > cap || mo.counter[2] != 0u;       // (see composite_train_forward_kernel): the incoming gradients are NaN, the outgoing ones exact zeros
    if (void_step) {
        if (e < cap) {
            grgb[e * 3] = 0.0f; grgb[e * 3 + 1] = 0.0f; grgb[e * 3 + 2] = 0.0f;
            grad_w[e] = 0.0f;
            if (g8) gin0[e] = 0.0f;
        }
    } else if (e < m) {
        const unsigned r = mo.q_ray[e];
        float gm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p = pre[(size_t)r * 3 + c];
            gm[c] = (p >= 0.0f && p <= 1.0f) ? g_map[(size_t)r * 3 + c] : 0.0f;
        }
        const float w = mo.q_pos[e].w;
        const float x0 = rgb[e * 3], x1 = rgb[e * 3 + 1], x2 = rgb[e * 3 + 2];
        const float a0 = w * gm[0], a1 = w * gm[1], a2 = w * gm[2];
        grgb[e * 3] = a0; grgb[e * 3 + 1] = a1; grgb[e * 3 + 2] = a2;
        float gw = (x0 * gm[0] + x1 * gm[1]) + x2 * gm[2];
        float scale = 1.0f;
        if (g8) {                                              // REFTensoRF: the colour is relu(tint) * rgb_s + rgb_d; the penalty term
            scale = fmaxf(g8[e * 8 + 3], 0.0f);
            const float q = fmaxf(feats32[e * 32 + 30], 0.0f);
            const float gp = g_pen ? g_pen[r] : 0.0f;
            gw = gw + gp * (q * q);
            gin0[e] = gp * w * (2.0f * q);
        }
        grad_w[e] = gw;
        v = fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fabsf(a2)) * scale;       // (the heads' transpose product has its own scale: tvr_mlp_train.hip)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0 && v > 0.0f) atomicMax(amax_bits, __float_as_uint(v));
}

__global__ __launch_bounds__(256) void composite_train_backward_rays_kernel(const int n_rays, const int white_bg, const float *__restrict__ pre, const float *__restrict__ g_map,
                                                                            float *__restrict__ grad_acc)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float s = 0.0f;
    if (white_bg) {                                            // (a void step left NaN in `pre`: both comparisons fail, the gradient is zero)
        float gm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p = pre[(size_t)r * 3 + c];
            gm[c] = (p >= 0.0f && p <= 1.0f) ? g_map[(size_t)r * 3 + c] : 0.0f;
        }
        s = -((gm[0] + gm[1]) + gm[2]);
    }
    grad_acc[r] = s;
}

// gscale = 2^floor(log2(target / (0.25 max))) in [2^-60, 2^60] (autograd_ops.py's host formula, on the device); the max word is cleared for the next step
__global__ void grad_scale_kernel(unsigned *__restrict__ amax_bits, const float target, float *__restrict__ gscale)
{
    const float gmax = fmaxf(__uint_as_float(*amax_bits) * 0.25f, 1e-30f);
    float e = floorf(log2f(target / gmax));
    e = fminf(fmaxf(e, -60.0f), 60.0f);
    *gscale = exp2f(e);
    *amax_bits = 0u;
}

// column sums of A [M, K] (row stride lda, K <= 128), M on the device: workgroup b sums its slab of rows — KP = K rounded up to a power of two columns x
// 256 / KP row lanes, four rows in flight per thread, the row lanes added in a fixed order through LDS — partials summed by colsum_final_kernel.
#define CS_GRID 512
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ A, const int lda, const int K, const int KP, const long long m_cap,
                                                             const unsigned *__restrict__ m_dev, float *__restrict__ P)
{
    __shared__ float red[256];
    const long long M = m_dev ? ((long long)*m_dev < m_cap ? (long long)*m_dev : m_cap) : m_cap;
    const long long rpb = (M + CS_GRID - 1) / CS_GRID;
    const long long m0 = (long long)blockIdx.x * rpb, m1 = m0 + rpb < M ? m0 + rpb : M;
    const int RL = 256 / KP, c = threadIdx.x % KP, rl = threadIdx.x / KP;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (c < K) {
        long long r = m0 + rl;
        for (; r + 3LL * RL < m1; r += 4LL * RL) {
            const float a0 = A[r * lda + c], a1 = A[(r + RL) * lda + c], a2 = A[(r + 2LL * RL) * lda + c], a3 = A[(r + 3LL * RL) * lda + c];
            s0 = s0 + a0; s1 = s1 + a1; s2 = s2 + a2; s3 = s3 + a3;
        }
        for (; r < m1; r += RL) s0 = s0 + A[r * lda + c];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < K) {
        float s = red[c];
        for (int j = 1; j < RL; ++j) s = s + red[j * KP + c];
        P[(size_t)blockIdx.x * K + c] = s;
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ P, const int K, float *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t >> 4, l = t & 15;
    float s = 0.0f;
    if (e < K)
        for (int b = l; b < CS_GRID; b += 16) s = s + P[(size_t)b * K + e];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s = s + __shfl_xor(s, off);
    if (e < K && l == 0) out[e] = s;
}

hipError_t launch_composite_train_forward(const MarchOut &mo, int n_rays, long long cap, int white_bg, const float *rgb, const float *feats32, int with_pen, float *rgb_map,
                                          float *pre, float *pen_ray, hipStream_t stream)
{
    const long long threads = (long long)n_rays * CT_LANES;
    hipLaunchKernelGGL(composite_train_forward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, mo, n_rays, cap, white_bg, rgb, feats32, with_pen, rgb_map,
                       pre, pen_ray);
    return hipGetLastError();
}

hipError_t launch_composite_train_backward(const MarchOut &mo, int n_rays, long long cap, int white_bg, const float *rgb, const float *feats32, const float *g8, const float *pre,
                                           const float *g_map, const float *g_pen, float *grgb, float *gin0, float *grad_w, float *grad_acc, unsigned *amax_bits,
                                           float target, float *gscale, hipStream_t stream)
{
    hipLaunchKernelGGL(composite_train_backward_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, mo, cap, rgb, feats32, g8, pre, g_map, g_pen, grgb, gin0,
                       grad_w, amax_bits);
    hipLaunchKernelGGL(composite_train_backward_rays_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, stream, n_rays, white_bg, pre, g_map, grad_acc);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, stream, amax_bits, target, gscale);
    return hipGetLastError();
}

// small device-to-device copies as a kernel (pieces of a gemm_tn / column-sum result into the gradient tensors they belong to): inside a captured graph a
// kernel node; hipMemcpyAsync nodes of 4..12 bytes from unaligned sources crashed hipGraphInstantiate on this ROCm (tests/test_gpu_fused_step.py, REFTensoRF case)
__global__ __launch_bounds__(256) void copy_f32_kernel(float *__restrict__ dst, const float *__restrict__ src, const int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

hipError_t launch_copy_f32(float *dst, const float *src, int n, hipStream_t stream)
{
    int grid = (n + 255) / 256;
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(copy_f32_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

// zero fill as a kernel.  Inside a captured training step every clear is a KERNEL node: hipMemsetAsync nodes of a graph that is replayed back to back
// were observed to run ahead of the previous replay's kernels on this ROCm (the scratch header held garbage, the step diverged), round 3.
__global__ __launch_bounds__(256) void zero_f32_kernel(float4 *__restrict__ p, const long long n4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
hipError_t launch_zero_f32(float *p, long long n, hipStream_t stream)       // p 16-byte aligned, n a multiple of 4
{
    const long long n4 = n / 4;
    long long grid = (n4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(grid > 0 ? grid : 1)), dim3(256), 0, stream, (float4 *)p, n4);
    return hipGetLastError();
}

// the scratch header's four words {queue length, tile counter, fault, overflow} cleared by a kernel (a kernel node inside a captured step)
__global__ void zero_header_kernel(unsigned *__restrict__ c) { c[threadIdx.x] = 0u; }       // the whole 256-B header (64 words; word 32: the march's ray counter)
// TensorBase.filtering_rays (tensorBase.py:411-441) as one pass, one lane per ray.  bbox_only: the slab test with the reference's `1e-6` for zero direction
// components, mask = t_max > t_min.  Otherwise: the evaluation-mode samples of sample_ray (:340-360: entry distance clamped to [near, far], j * stepSize) looked
// up in the alpha volume — ALL of them, as the reference does (it ignores sample_ray's own bbox mask here; a point outside the volume reads zero) —
// mask = any(alpha > 0), with the march kernel's arithmetic and the same bit-volume / float lookup.  The reference's form materialises [chunk, N_samples, 3] points per
// 51 200-ray chunk and copies every chunk's mask to the host: 12 s for a 64 M-ray training set against 2 minutes of training.
__global__ __launch_bounds__(256) void filter_rays_kernel(const SceneDev sc, const float *__restrict__ rays, const long long n, const int S, const int bbox_only,
                                                          unsigned char *__restrict__ mask)
{
    const long long ray = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= n) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = rays[ray * 6 + k];
        d[k] = rays[ray * 6 + 3 + k];
    }
    if (bbox_only) {
        float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = (d[k] == 0.0f) ? 1e-6f : d[k];
            const float ra = (sc.hi[k] - o[k]) / v, rb = (sc.lo[k] - o[k]) / v;
            const float lo = ra < rb ? ra : rb, hi = ra > rb ? ra : rb;
            tmin = lo > tmin ? lo : tmin;
            tmax = hi < tmax ? hi : tmax;
        }
        mask[ray] = tmax > tmin ? 1 : 0;
        return;
    }
    const float tmin = ray_tmin(sc, o, d);
    unsigned char hit = 0;
    for (int j = 0; j < S; ++j) {
        const float z = tmin + sc.step * (float)j;
        float p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = o[k] + d[k] * z;
        if (sc.abits ? alpha_positive(sc, p) : (alpha_lookup(sc, p) > 0.0f)) { hit = 1; break; }
    }
    mask[ray] = hit;
}

hipError_t launch_filter_rays(const SceneDev &sc, const float *rays, long long n, int S, int bbox_only, unsigned char *mask, hipStream_t stream)
{
    hipLaunchKernelGGL(filter_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sc, rays, n, S, bbox_only, mask);
    return hipGetLastError();
}

// fp32 -> fp16 (round to nearest even), four values per thread: the appearance factors' second image (tvr_api.hip refresh_h16)
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float4 *__restrict__ in, uint2 *__restrict__ out, long long n4)
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        uint2 o;
        o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.x, v.y}, f16x2));
        o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v.z, v.w}, f16x2));
        out[i] = o;
    }
}

hipError_t launch_f32_to_f16(const float *in, void *out, long long n, hipStream_t stream)
{
    const long long n4 = n / 4;
    if (n4 <= 0) return hipSuccess;
    unsigned grid = (unsigned)((n4 + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid), dim3(256), 0, stream, (const float4 *)in, (uint2 *)out, n4);
    return hipGetLastError();
}

hipError_t launch_zero_header(unsigned *counter, hipStream_t stream)
{
    hipLaunchKernelGGL(zero_header_kernel, dim3(1), dim3(64), 0, stream, counter);
    return hipGetLastError();
}

size_t colsum_scratch_bytes() { return (size_t)CS_GRID * 128 * sizeof(float); }

hipError_t launch_colsum(const float *A, int lda, int K, long long m_cap, const unsigned *m_dev, float *out, float *scratch, hipStream_t stream)
{
    int KP = 1;
    while (KP < K) KP <<= 1;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(CS_GRID), dim3(256), 0, stream, A, lda, K, KP, m_cap, m_dev, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((K * 16 + 255) / 256), dim3(256), 0, stream, scratch, K, out);
    return hipGetLastError();
}


// ---- the training queue in RAY ORDER (round 4) ---------------------------------------------------------------------------------------------------------
// The march kernel places a ray's entries where one atomicAdd says, so the order of the rays in the queue is the order its waves finish in: every sum over
// all appearance samples (the weight-gradient products: fixed order over the rows as they lie) then runs over a run-dependent row order, and the network's
// gradients are reproducible to rounding only.  For a TRAINING batch (4096 rays, < 1 M entries) putting the queue in ray order afterwards costs three tiny
// launches (~10 us of a 3.4 ms step): an exclusive scan of the per-ray counts, a gather of every ray's segment to its scanned offset (into the queue's unused
// q_out / q_j regions), a copy back.  The inference march is untouched (a frame's 55 M entries would cost 0.4 ms to move, and its picture does not depend on
// the order).  After this pass the weight gradients of two runs on the same batch are bit-identical (tests/test_gpu_fused_step.py).
// A march that raised its fault flag (counter[2] != 0: a wave gave up in the tile wait and never wrote ray_off / ray_cnt for its rays) leaves those words as they
// lay — uninitialised scratch for all this pass knows.  The step is void then (composite_train_forward writes NaN, the backward zeros): all three kernels return
// at once, and the gather clamps every segment to the queue's length besides, so that no count or offset it reads can carry an access out of the buffers.
__global__ __launch_bounds__(1024) void queue_scan_kernel(const unsigned *__restrict__ counter, const unsigned *__restrict__ ray_cnt, unsigned *__restrict__ ray_new, const int n_rays)
{
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (counter[2] != 0u) return;                                  // (uniform over the workgroup: nobody reaches a barrier)
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (int base = 0; base < n_rays; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const unsigned v = i < n_rays ? ray_cnt[i] : 0u;
        unsigned x = v;                                            // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        unsigned before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < n_rays) ray_new[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + x;
        __syncthreads();
    }
}

// one wave per ray: its segment [ray_off, ray_off + ray_cnt) -> [ray_new, ...) of the temporary arrays; then the ray's offset is the new one
__global__ __launch_bounds__(256) void queue_gather_kernel(const MarchOut mo, const unsigned *__restrict__ ray_new, float4 *__restrict__ tmp_pos, unsigned *__restrict__ tmp_ray,
                                                           const int n_rays)
{
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= n_rays || mo.counter[2] != 0u) return;
    const unsigned m = *mo.counter;
    const unsigned from = mo.ray_off[ray], to = ray_new[ray];
    unsigned cnt = mo.ray_cnt[ray];
    if (from > m || to > m || cnt > m - from || cnt > m - to) cnt = 0u;      // never true behind a sound march (the scan of the counts ends at m)
    for (unsigned i = lane; i < cnt; i += 64) {
        tmp_pos[to + i] = mo.q_pos[from + i];
        tmp_ray[to + i] = mo.q_ray[from + i];
    }
    if (lane == 0) mo.ray_off[ray] = to;
}

__global__ __launch_bounds__(256) void queue_copyback_kernel(const MarchOut mo, const float4 *__restrict__ tmp_pos, const unsigned *__restrict__ tmp_ray)
{
    if (mo.counter[2] != 0u) return;
    const unsigned m = *mo.counter;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < m; e += gridDim.x * blockDim.x) {
        mo.q_pos[e] = tmp_pos[e];
        mo.q_ray[e] = tmp_ray[e];
    }
}

hipError_t launch_queue_ray_order(const MarchOut &mo, unsigned *ray_new, float4 *tmp_pos, unsigned *tmp_ray, int n_rays, hipStream_t stream)
{
    hipLaunchKernelGGL(queue_scan_kernel, dim3(1), dim3(1024), 0, stream, mo.counter, mo.ray_cnt, ray_new, n_rays);
    hipLaunchKernelGGL(queue_gather_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, stream, mo, ray_new, tmp_pos, tmp_ray, n_rays);
    hipLaunchKernelGGL(queue_copyback_kernel, dim3(1024), dim3(256), 0, stream, mo, tmp_pos, tmp_ray);
    return hipGetLastError();
}
