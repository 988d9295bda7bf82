// tvr_train.hip — backward kernels for the training step (the caller is tensorf-myc/train.py:225-261).
//
//   march_backward_kernel : d loss / d density planes+lines, given d loss / d w_e (per appearance sample) and d loss / d acc (per ray)
//   app_h_forward_kernel  : h[m,144] = bilinear(app_plane)·linear(app_line) for queue positions (tensoRF.py:235-241; the 144->27 basis,
//                           the PE and the three Linears then run as plain rocBLAS GEMMs under torch autograd in training)
//   app_h_backward_kernel : scatter-add of d loss / d h into the appearance planes+lines
//   unpack_grad_kernel    : packed channels-last gradient image -> the reference parameter layout (1,C,H,W)
//
// Compositing gradient in FORWARD order (no reverse scan): with w_k = alpha_k T_k and T_k = prod_{i<k} (1 - alpha_i + 1e-10),
//   dL/dalpha_j = T_j dL/dw_j - (sum_{k>j} dL/dw_k w_k) / (1 - alpha_j + 1e-10),
// and the suffix sum is (total - inclusive prefix), where total = sum_e grad_w_e w_e + grad_acc * acc is known from the forward
// pass.  dL/dw_j = grad_acc (every sample feeds acc, tensorBase.py:520) + grad_w_e if the sample is an appearance sample.
// The kernel recomputes the forward march bit for bit (same code, same eps_T), so "k-th sample with w > thres" is queue entry
// ray_off + k again.
#include "tvr_device.h"
#include "tvr_kernels.h"

#define TB_WAVES 16                      // march_backward: one ray per wave, 16 rays per workgroup
#define TB_THREADS (64 * TB_WAVES)
#ifndef TB_DIAG
#define TB_DIAG 0                        // timing experiments only: 1 no line atomics, 2 no pass C, 4 no plane atomics, 8 no pass A
#endif
#define TB_STAGE (2 * 3 * 16 * TVR_CD)   // floats per wave: [Q | P] of 3 pairs x 16 samples x 16 channels (pass A -> pass C)
#define AHB_THREADS 1024                 // 16 waves, each walking AHB_ENTRIES / 16 consecutive entries with one lane per channel
#define AHB_ENTRIES 2048                 // app_h_backward: queue entries per workgroup (per plane)

template <int CTRL>
__device__ __forceinline__ int qperm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int qbcast_i(int v, int k)
{
    switch (k) {
    case 0: return qperm_i<0x00>(v);
    case 1: return qperm_i<0x55>(v);
    case 2: return qperm_i<0xAA>(v);
    default: return qperm_i<0xFF>(v);
    }
}
__device__ __forceinline__ float qbcast_f(float v, int k) { return __int_as_float(qbcast_i(__float_as_int(v), k)); }
__device__ __forceinline__ float qxor_sum(float v)
{
    v += __int_as_float(qperm_i<0xB1>(__float_as_int(v)));
    v += __int_as_float(qperm_i<0x4E>(__float_as_int(v)));
    return v;
}

__device__ __forceinline__ void atomic_add4(float *p, float4 v)
{
    atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

// texels + interpolated values of one (plane, line) pair for this lane's 4 channels; optionally scatters gradients
struct VmTerm {
    float4 P, Q;          // bilinear(plane)[4ch], linear(line)[4ch]
};

template <int TPT>
__device__ __forceinline__ VmTerm vm_eval(const float4 *__restrict__ Pl, const float4 *__restrict__ Ln, int W, int x0, int y0, int l0, float wx,
                                          float wy, float wl, int sub)
{
    const float ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
    const int Wp = W + 1;
    const float4 *p = Pl + ((size_t)y0 * Wp + x0) * TPT + sub;
    const float4 t00 = p[0], t01 = p[TPT], t10 = p[(size_t)Wp * TPT], t11 = p[(size_t)Wp * TPT + TPT];
    const float4 *q = Ln + (size_t)l0 * TPT + sub;
    const float4 l0v = q[0], l1v = q[TPT];
    VmTerm r;
    r.P = f4_mul(ux * uy, t00);
    r.P = f4_fma(wx * uy, t01, r.P);
    r.P = f4_fma(ux * wy, t10, r.P);
    r.P = f4_fma(wx * wy, t11, r.P);
    r.Q = f4_mul(ul, l0v);
    r.Q = f4_fma(wl, l1v, r.Q);
    return r;
}

// d(plane taps) += gP * w_tap, d(line taps) += gQ * w_l   (gP = dL/dP[4ch], gQ = dL/dQ[4ch]); packed gradient images
template <int TPT>
__device__ __forceinline__ void vm_scatter(float *__restrict__ gPl, float *__restrict__ gLn, int W, int x0, int y0, int l0, float wx, float wy,
                                           float wl, int sub, float4 gP, float4 gQ)
{
    const float ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
    const int Wp = W + 1;
    float *p = gPl + (((size_t)y0 * Wp + x0) * TPT + sub) * 4;
    atomic_add4(p, f4_mul(ux * uy, gP));
    atomic_add4(p + TPT * 4, f4_mul(wx * uy, gP));
    atomic_add4(p + (size_t)Wp * TPT * 4, f4_mul(ux * wy, gP));
    atomic_add4(p + (size_t)Wp * TPT * 4 + TPT * 4, f4_mul(wx * wy, gP));
    float *q = gLn + ((size_t)l0 * TPT + sub) * 4;
    atomic_add4(q, f4_mul(ul, gQ));
    atomic_add4(q + TPT * 4, f4_mul(wl, gQ));
}

// Line gradients are accumulated in LDS and flushed once per workgroup: every sample of every ray at the same height adds into the
// same (L+1) x C line texels, and same-address global atomics serialise in L2 (measured: 10.1 -> see DESIGN.md ms per 4096-ray step).
// LINE_LDS = false (the three lines do not fit the LDS) keeps the global atomics.
template <int TPT, bool LINE_LDS>
__device__ __forceinline__ void vm_scatter_l(float *__restrict__ gPl, float *__restrict__ gLn, float *gLds, int W, int x0, int y0, int l0,
                                             float wx, float wy, float wl, int sub, float4 gP, float4 gQ)
{
    const float ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
    const int Wp = W + 1;
    float *p = gPl + (((size_t)y0 * Wp + x0) * TPT + sub) * 4;
    atomic_add4(p, f4_mul(ux * uy, gP));
    atomic_add4(p + TPT * 4, f4_mul(wx * uy, gP));
    atomic_add4(p + (size_t)Wp * TPT * 4, f4_mul(ux * wy, gP));
    atomic_add4(p + (size_t)Wp * TPT * 4 + TPT * 4, f4_mul(wx * wy, gP));
    float *q = (LINE_LDS ? gLds : gLn) + ((size_t)l0 * TPT + sub) * 4;
    atomic_add4(q, f4_mul(ul, gQ));
    atomic_add4(q + TPT * 4, f4_mul(wl, gQ));
}

// one density channel of one (plane, line) pair: re-evaluate P, Q and scatter gs*Q into the 4 plane taps, gs*P into the 2 line taps
template <bool LINE_LDS>
__device__ __forceinline__ void scatter_ch(const float *__restrict__ Pl, const float *__restrict__ Ln, float *__restrict__ gPl,
                                           float *__restrict__ gLn, float *gLds, int W, int x0, int y0, int l0, float wx, float wy, float wl,
                                           int c, float gs)
{
    const float ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
    const int Wp = W + 1;
    const size_t t00 = ((size_t)y0 * Wp + x0) * TVR_CD + c, t10 = t00 + (size_t)Wp * TVR_CD;
    const size_t q0 = (size_t)l0 * TVR_CD + c;
    float P = (ux * uy) * Pl[t00];
    P = __builtin_fmaf(wx * uy, Pl[t00 + TVR_CD], P);
    P = __builtin_fmaf(ux * wy, Pl[t10], P);
    P = __builtin_fmaf(wx * wy, Pl[t10 + TVR_CD], P);
    float Q = ul * Ln[q0];
    Q = __builtin_fmaf(wl, Ln[q0 + TVR_CD], Q);
    const float gP = gs * Q, gQ = gs * P;
    atomicAdd(gPl + t00, (ux * uy) * gP);
    atomicAdd(gPl + t00 + TVR_CD, (wx * uy) * gP);
    atomicAdd(gPl + t10, (ux * wy) * gP);
    atomicAdd(gPl + t10 + TVR_CD, (wx * wy) * gP);
    float *q = (LINE_LDS ? gLds : gLn) + q0;
    atomicAdd(q, ul * gQ);
    atomicAdd(q + TVR_CD, wl * gQ);
}

// The plane gradients of consecutive samples of a ray accumulate in a 2 x 2 WINDOW of taps that slides with the ray: a step into the neighbouring cell
// retires the two taps that leave the window and keeps the two that stay (a run that flushed all four taps at every cell change issued twice the atomics).
// What is charged is the 64-B line request, not the dword (scripts/hwprobe/atomic_rate.hip), so every retired tap costs one request per 16 channels.
// One lane per channel c of a C-channel texel; all lanes of a sample take the same branches.
template <int C>
struct TapWin {
    int kx, ky;                  // the window's (x0, y0); an empty window holds zeros, which are never written
    float a00, a01, a10, a11;    // taps (kx, ky), (kx+1, ky), (kx, ky+1), (kx+1, ky+1)
    __device__ __forceinline__ void reset() { kx = 0; ky = 0; a00 = a01 = a10 = a11 = 0.0f; }
    __device__ __forceinline__ static void put(float *__restrict__ g, int Wp, int c, int x, int y, float v)
    {
        if (v != 0.0f) atomicAdd(g + ((long long)y * Wp + x) * C + c, v);
    }
    __device__ __forceinline__ void flush(float *__restrict__ g, int Wp, int c)
    {
        put(g, Wp, c, kx, ky, a00); put(g, Wp, c, kx + 1, ky, a01); put(g, Wp, c, kx, ky + 1, a10); put(g, Wp, c, kx + 1, ky + 1, a11);
        a00 = a01 = a10 = a11 = 0.0f;
    }
    __device__ __forceinline__ void move(float *__restrict__ g, int Wp, int c, int x0, int y0)
    {
        const int dx = x0 - kx;
        if (dx != 0) {
            if (dx == 1) { put(g, Wp, c, kx, ky, a00); put(g, Wp, c, kx, ky + 1, a10); a00 = a01; a10 = a11; a01 = a11 = 0.0f; }
            else if (dx == -1) { put(g, Wp, c, kx + 1, ky, a01); put(g, Wp, c, kx + 1, ky + 1, a11); a01 = a00; a11 = a10; a00 = a10 = 0.0f; }
            else flush(g, Wp, c);
            kx = x0;
        }
        const int dy = y0 - ky;
        if (dy != 0) {
            if (dy == 1) { put(g, Wp, c, kx, ky, a00); put(g, Wp, c, kx + 1, ky, a01); a00 = a10; a01 = a11; a10 = a11 = 0.0f; }
            else if (dy == -1) { put(g, Wp, c, kx, ky + 1, a10); put(g, Wp, c, kx + 1, ky + 1, a11); a10 = a00; a11 = a01; a00 = a01 = 0.0f; }
            else flush(g, Wp, c);
            ky = y0;
        }
    }
};

// the line's two taps (l0, l0 + 1) slide the same way; g is the LDS image of the line (or the global one when it does not fit)
template <int C>
struct LineWin {
    int lk;
    float b0, b1;
    __device__ __forceinline__ void reset() { lk = 0; b0 = b1 = 0.0f; }
    __device__ __forceinline__ void flush(float *g, int c)
    {
        if (b0 != 0.0f) atomicAdd(g + (size_t)lk * C + c, b0);
        if (b1 != 0.0f) atomicAdd(g + (size_t)(lk + 1) * C + c, b1);
        b0 = b1 = 0.0f;
    }
    __device__ __forceinline__ void move(float *g, int c, int l0)
    {
        const int dl = l0 - lk;
        if (dl != 0) {
            if (dl == 1) { if (b0 != 0.0f) atomicAdd(g + (size_t)lk * C + c, b0); b0 = b1; b1 = 0.0f; }
            else if (dl == -1) { if (b1 != 0.0f) atomicAdd(g + (size_t)(lk + 1) * C + c, b1); b1 = b0; b0 = 0.0f; }
            else flush(g, c);
            lk = l0;
        }
    }
};

__device__ __forceinline__ float rdlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// march_backward, pass A: one quad lane's 4 channels of one (plane, line) pair of one sample — P, Q as phase 1 computes them, staged in LDS for pass C
// (sq: this sample's 16-channel row of the pair's Q block; the P block follows TB_STAGE / 2 floats later).
__device__ __forceinline__ void pass_a(const float4 *__restrict__ Pl, const float4 *__restrict__ Ln, int W, int x0, int y0, int l0, float wx, float wy,
                                       float wl, int sub, float *sq)
{
    const VmTerm t = vm_eval<TVR_CD / 4>(Pl, Ln, W, x0, y0, l0, wx, wy, wl, sub);
    *(float4 *)sq = t.Q;
    *(float4 *)(sq + TB_STAGE / 2) = t.P;
}

// march_backward, pass C: the 2 x 2 tap window of one plane, spread over the wave — lane (ti, tj, ch) owns tap (kx + ti, ky + tj), channel ch.  The
// window's position is wave-uniform (scalar registers); a step to the neighbouring cell retires one column or row with ONE atomic instruction (32
// lanes, 2 line requests) and hands the staying column / row over by a lane exchange.
struct RayWin {
    int kx, ky;
    float a;
    __device__ __forceinline__ void reset() { kx = 0; ky = 0; a = 0.0f; }
    __device__ __forceinline__ void put(float *__restrict__ g, int Wp, int ch, int ti, int tj) const
    {
        if (!(TB_DIAG & 4) && a != 0.0f) atomicAdd(g + ((long long)(ky + tj) * Wp + (kx + ti)) * TVR_CD + ch, a);
    }
    __device__ __forceinline__ void flush(float *__restrict__ g, int Wp, int ch, int ti, int tj) { put(g, Wp, ch, ti, tj); a = 0.0f; }
    // x0, y0, wx, wy are wave-uniform; v = gs * Q[ch]
    __device__ __forceinline__ void add(float *__restrict__ g, int Wp, int ch, int ti, int tj, int x0, int y0, float wx, float wy, float v)
    {
        const int dx = x0 - kx;
        if (dx != 0) {
            if (dx == 1 || dx == -1) {
                const bool leaving = ti == (dx == 1 ? 0 : 1);
                if (leaving) put(g, Wp, ch, ti, tj);
                const float o = __shfl_xor(a, 16);
                a = leaving ? o : 0.0f;                              // the staying column moves to the leaving column's lanes
            } else flush(g, Wp, ch, ti, tj);
            kx = x0;
        }
        const int dy = y0 - ky;
        if (dy != 0) {
            if (dy == 1 || dy == -1) {
                const bool leaving = tj == (dy == 1 ? 0 : 1);
                if (leaving) put(g, Wp, ch, ti, tj);
                const float o = __shfl_xor(a, 32);
                a = leaving ? o : 0.0f;
            } else flush(g, Wp, ch, ti, tj);
            ky = y0;
        }
        const float fx = ti ? wx : 1.0f - wx, fy = tj ? wy : 1.0f - wy;
        a += (fx * fy) * v;
    }
};

// the line's two taps the same way: lanes with tj == 0 own tap lk + ti, channel ch.  Consecutive samples of a ray mostly share l0 — as LDS atomics issued per
// sample those were same-address conflicts that serialised (0.93 ms of the kernel's 1.5 ms, measured by leaving them out); the window issues one
// conflict-free 16-lane atomic per tap that leaves.
struct RayLine {
    int lk;
    float b;
    __device__ __forceinline__ void reset() { lk = 0; b = 0.0f; }
    __device__ __forceinline__ void put(float *g, int ch, int ti, int tj) const
    {
        if (!(TB_DIAG & 1) && tj == 0 && b != 0.0f) atomicAdd(g + (size_t)(lk + ti) * TVR_CD + ch, b);
    }
    __device__ __forceinline__ void flush(float *g, int ch, int ti, int tj) { put(g, ch, ti, tj); b = 0.0f; }
    __device__ __forceinline__ void add(float *g, int ch, int ti, int tj, int l0, float wl, float v)
    {
        const int dl = l0 - lk;
        if (dl != 0) {
            if (dl == 1 || dl == -1) {
                const bool leaving = ti == (dl == 1 ? 0 : 1);
                if (leaving) put(g, ch, ti, tj);
                const float o = __shfl_xor(b, 16);
                b = leaving ? o : 0.0f;
            } else flush(g, ch, ti, tj);
            lk = l0;
        }
        b += (ti ? wl : 1.0f - wl) * v;
    }
};

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <bool LINE_LDS>
__global__ __launch_bounds__(TB_THREADS) void march_backward_kernel(const SceneDev sc, const float *__restrict__ rays, const int n_rays, const int S,
                                                                    const MarchSampling sm, const float eps_T, const int rays_per_block,
                                                                    const MarchOut mo, const float *__restrict__ grad_w,
                                                                    const float *__restrict__ grad_acc, const float *__restrict__ lam6,
                                                                    const float *__restrict__ grad_lam6, TrainGrads tg, const long long gw_cap)
{
    // fused training step (gw_cap = the capacity of grad_w): a step whose appearance queue outgrew the workspace, or whose march raised its fault flag, is VOID —
    // its loss is NaN (composite_train_forward) and every gradient is exactly zero; grad_w holds nothing beyond gw_cap, so nothing is read at all
    if (gw_cap >= 0 && ((long long)*mo.counter > gw_cap || mo.counter[2] != 0u)) return;
    // LDS: [per-wave stage: 3 planes x 16 samples x 16 channels of Q][line 0 | line 1 | line 2 gradient accumulators, (L+1) x 16 each]
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane & 3;
    float *const stage = smem_f + wave * TB_STAGE;
    float *const glds = smem_f + TB_WAVES * TB_STAGE;
    const int ln0 = (sc.grid[2] + 1) * TVR_CD, ln1 = (sc.grid[1] + 1) * TVR_CD, ln2 = (sc.grid[0] + 1) * TVR_CD;
    float *const gl0 = glds, *const gl1 = glds + ln0, *const gl2 = glds + ln0 + ln1;
    if (LINE_LDS) {
        for (int i = threadIdx.x; i < ln0 + ln1 + ln2; i += TB_THREADS) glds[i] = 0.0f;
        __syncthreads();
    }
    // pass C's lane: tap (ti, tj) of the 2 x 2 window, channel ch
    const int ch = lane & 15, ti = (lane >> 4) & 1, tj = lane >> 5;
    for (int it = wave; it < rays_per_block; it += TB_WAVES) {
        const int ray = blockIdx.x * rays_per_block + it;
        if (ray >= n_rays) break;
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = rays[(size_t)ray * 6 + k];
            d[k] = rays[(size_t)ray * 6 + 3 + k];
        }
        const float tmin = ray_tmin(sc, o, d);
        const bool has_jit = sm.jitter != nullptr;
        const float u = has_jit ? sm.jitter[ray] : 0.0f;
        const float *__restrict__ zrow = sm.zv ? sm.zv + (size_t)ray * S : nullptr;
        // d lam6 / d alpha_k = -lam6 / (1 - alpha_k + 1e-6)   (lam6 = prod_j (1 - alpha_j + 1e-6), nerfplusplus.py:277-278)
        const float g6 = grad_lam6 ? grad_lam6[ray] * lam6[ray] : 0.0f;
        const unsigned base = mo.ray_off[ray], cnt = mo.ray_cnt[ray];
        const float gacc = grad_acc[ray];
        // total = sum_k dL/dw_k w_k over the whole (possibly early-terminated) ray
        float tot_l = 0.0f;
        for (unsigned i = lane; i < cnt; i += 64) tot_l += grad_w[base + i] * mo.q_pos[base + i].w;
        const float total = wave_sum_f(tot_l) + gacc * mo.acc[ray];

        float T = 1.0f, prefix = 0.0f;
        unsigned napp = 0;
        bool seen = false;
        RayWin win0, win1, win2;                                     // the three pairs' tap windows live for the whole ray
        RayLine lw0, lw1, lw2;
        win0.reset(); win1.reset(); win2.reset();
        lw0.reset(); lw1.reset(); lw2.reset();
        for (int c = 0; c * 64 < S; ++c) {
            const int j = c * 64 + lane;
            const bool inr = j < S;
            float fj = (float)j, fj1 = (float)(j + 1);
            if (has_jit) { fj = fj + u; fj1 = fj1 + u; }
            float z = tmin + sc.step * fj;
            float z1 = tmin + sc.step * fj1;
            if (zrow) {
                z = inr ? zrow[j] : 0.0f;
                z1 = (j < S - 1) ? zrow[j + 1] : z;
            }
            float p[3], n[3], f[3];
            bool bbox = inr;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                p[k] = o[k] + d[k] * z;
                bbox = bbox & !((sc.lo[k] > p[k]) | (p[k] > sc.hi[k]));
            }
            bool valid = bbox;
            if (sc.avol != nullptr) {
                if (bbox) valid = sc.abits ? alpha_positive(sc, p) : (alpha_lookup(sc, p) > 0.0f);
            }
            int i0[3];
            float w[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                n[k] = (p[k] - sc.lo[k]) * sc.inv[k] - 1.0f;
                f[k] = unnorm(n[k], sc.gm1[k]);
                const float fl = floorf(f[k]);
                i0[k] = (int)fl;
                w[k] = f[k] - fl;
            }
            const unsigned long long mb = __ballot(bbox), mv = __ballot(valid);
            if (mv == 0ull) {
                if (mb == 0ull && seen) break;
                continue;
            }
            seen = true;
            // ---- phase 1: forward recomputation of sigma_feature (identical arithmetic to march_kernel) ----
            float sf = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const bool v = qbcast_i((int)valid, k4) != 0;
                if (__ballot(v) == 0ull) continue;
                const int ix = qbcast_i(i0[0], k4), iy = qbcast_i(i0[1], k4), iz = qbcast_i(i0[2], k4);
                const float wx = qbcast_f(w[0], k4), wy = qbcast_f(w[1], k4), wz = qbcast_f(w[2], k4);
                float part = 0.0f;
                if (v) {
                    const float4 a = vm_term<4, false>(sc.dplane[0], sc.dline[0], sc.grid[0], sc.grid[1], sc.grid[2], ix, iy, iz, wx, wy, wz, sub);
                    const float4 b = vm_term<4, false>(sc.dplane[1], sc.dline[1], sc.grid[0], sc.grid[2], sc.grid[1], ix, iz, iy, wx, wz, wy, sub);
                    const float4 cc = vm_term<4, false>(sc.dplane[2], sc.dline[2], sc.grid[1], sc.grid[2], sc.grid[0], iy, iz, ix, wy, wz, wx, sub);
                    part = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((cc.x + cc.y) + (cc.z + cc.w));
                }
                part = qxor_sum(part);
                if (sub == k4) sf = part;
            }
            float sigma = 0.0f, dsig_dsf = 0.0f;
            if (valid) {
                if (sc.act == 0) {
                    const float x = sf + sc.shift;
                    sigma = softplus_f(x);
                    dsig_dsf = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
                } else {
                    sigma = fmaxf(sf, 0.0f);
                    dsig_dsf = sf > 0.0f ? 1.0f : 0.0f;
                }
            }
            float dist = (j < S - 1) ? (z1 - z) : 0.0f;
            dist = dist * sc.scale;
            const float alpha = 1.0f - expf(-sigma * dist);
            const float fT = (1.0f - alpha) + 1e-10f;
            float incl = fT;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_up(incl, off);
                if (lane >= off) incl = incl * t;
            }
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.0f;
            const float Tj = T * excl;
            const float wgt = alpha * Tj;
            const bool app = wgt > sc.thres;
            const unsigned long long ma = __ballot(app);
            float dLdw = gacc;
            if (app) dLdw += grad_w[base + napp + __popcll(ma & ((1ull << lane) - 1ull))];
            napp += __popcll(ma);
            // inclusive prefix of dL/dw_k w_k
            float pin = dLdw * wgt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_up(pin, off);
                if (lane >= off) pin += t;
            }
            const float suffix = total - (prefix + pin);
            prefix += __shfl(pin, 63);
            const float dLda = Tj * dLdw - suffix / fT - g6 / ((1.0f - alpha) + 1e-6f);
            const float dLdsf = valid ? dLda * dist * (1.0f - alpha) * dsig_dsf : 0.0f;
            T = T * __shfl(incl, 63);

            // ---- phase 2: scatter dL/dsf into the density planes / lines (re-gather: the texels are L1/L2 hot) ----
            // Round 3, second form.  The first form walked the samples with one lane per channel, re-gathered with 4-byte loads (288 wave-level loads per
            // chunk against phase 1's 72) and issued the line-tap LDS atomics per sample — consecutive samples share l0, so those were same-address
            // conflicts that serialised: 0.93 ms of a 1.5 ms kernel (0.11 ms forward).  Now, per 16 consecutive samples:
            //   pass A (quad per sample, float4 per lane — phase 1's loads): P, Q of the three pairs, staged in LDS (6 KB per wave);
            //   pass C (ONE sample at a time; lane = (tap of the 2 x 2 window, channel)): every lane owns ONE accumulator per plane and (tj == 0) one per
            //          line; the windows slide with the ray for the WHOLE ray (no flush at chunk ends), a retiring column / row is ONE atomic instruction
            //          of 2 line requests, a retiring line tap one conflict-free 16-lane LDS atomic; cell and weights are wave-uniform (readlane).
            for (int it = 0; it < 4; ++it) {
                if (__builtin_amdgcn_readfirstlane((int)((__ballot(dLdsf != 0.0f) >> (16 * it)) & 0xFFFFull)) == 0) continue;
                {
                    const int src = 16 * it + (lane >> 2);
                    const float gs = __shfl(dLdsf, src);
                    const int ix = __shfl(i0[0], src), iy = __shfl(i0[1], src), iz = __shfl(i0[2], src);
                    const float wx = __shfl(w[0], src), wy = __shfl(w[1], src), wz = __shfl(w[2], src);
                    if (!(TB_DIAG & 8) && gs != 0.0f) {
                        float *sq = stage + (lane >> 2) * TVR_CD + 4 * sub;
                        pass_a(sc.dplane[0], sc.dline[0], sc.grid[0], ix, iy, iz, wx, wy, wz, sub, sq);
                        pass_a(sc.dplane[1], sc.dline[1], sc.grid[0], ix, iz, iy, wx, wz, wy, sub, sq + 16 * TVR_CD);
                        pass_a(sc.dplane[2], sc.dline[2], sc.grid[1], iy, iz, ix, wy, wz, wx, sub, sq + 32 * TVR_CD);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (int u = 0; u < ((TB_DIAG & 2) ? 0 : 16); ++u) {
                    const int src = 16 * it + u;
                    const float gs = rdlane_f(dLdsf, src);
                    if (gs == 0.0f) continue;
                    const int ix = __builtin_amdgcn_readlane(i0[0], src), iy = __builtin_amdgcn_readlane(i0[1], src), iz = __builtin_amdgcn_readlane(i0[2], src);
                    const float wx = rdlane_f(w[0], src), wy = rdlane_f(w[1], src), wz = rdlane_f(w[2], src);
                    const float *sq = stage + u * TVR_CD + ch;
                    // sf = sum_i sum_c P_i[c] Q_i[c]  ->  dP_i[c] = gs Q_i[c] (tap-weighted into the plane), dQ_i[c] = gs P_i[c] (into the line)
                    win0.add(tg.dplane[0], sc.grid[0] + 1, ch, ti, tj, ix, iy, wx, wy, gs * sq[0]);
                    win1.add(tg.dplane[1], sc.grid[0] + 1, ch, ti, tj, ix, iz, wx, wz, gs * sq[16 * TVR_CD]);
                    win2.add(tg.dplane[2], sc.grid[1] + 1, ch, ti, tj, iy, iz, wy, wz, gs * sq[32 * TVR_CD]);
                    lw0.add(LINE_LDS ? gl0 : tg.dline[0], ch, ti, tj, iz, wz, gs * sq[TB_STAGE / 2]);
                    lw1.add(LINE_LDS ? gl1 : tg.dline[1], ch, ti, tj, iy, wy, gs * sq[TB_STAGE / 2 + 16 * TVR_CD]);
                    lw2.add(LINE_LDS ? gl2 : tg.dline[2], ch, ti, tj, ix, wx, gs * sq[TB_STAGE / 2 + 32 * TVR_CD]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (T < eps_T) break;
        }
        win0.flush(tg.dplane[0], sc.grid[0] + 1, ch, ti, tj);
        win1.flush(tg.dplane[1], sc.grid[0] + 1, ch, ti, tj);
        win2.flush(tg.dplane[2], sc.grid[1] + 1, ch, ti, tj);
        lw0.flush(LINE_LDS ? gl0 : tg.dline[0], ch, ti, tj);
        lw1.flush(LINE_LDS ? gl1 : tg.dline[1], ch, ti, tj);
        lw2.flush(LINE_LDS ? gl2 : tg.dline[2], ch, ti, tj);
    }
    if (LINE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < ln0; i += TB_THREADS) { const float v = gl0[i]; if (v != 0.0f) atomicAdd(tg.dline[0] + i, v); }
        for (int i = threadIdx.x; i < ln1; i += TB_THREADS) { const float v = gl1[i]; if (v != 0.0f) atomicAdd(tg.dline[1] + i, v); }
        for (int i = threadIdx.x; i < ln2; i += TB_THREADS) { const float v = gl2[i]; if (v != 0.0f) atomicAdd(tg.dline[2] + i, v); }
    }
}

// one thread per (entry, plane, float4 channel group): h[entry][48*plane + 4*q .. +3]
// xs: floats between consecutive positions (3: xyz [m,3]; 4: the march queue's {xyz, w} entries).  m_dev (optional): the entry count lives on the
// device (the queue length of a training step) and m is the capacity the buffers were sized for: min(*m_dev, m) entries are processed.
__global__ __launch_bounds__(256) void app_h_forward_kernel(const SceneDev sc, const float *__restrict__ xyz, const int xs, const long long m_cap,
                                                            const unsigned *__restrict__ m_dev, float *__restrict__ h)
{
    const long long m = m_dev ? ((long long)*m_dev < m_cap ? (long long)*m_dev : m_cap) : m_cap;
    const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= m * 36) return;
    const long long ent = item / 36;
    const int rem = (int)(item - ent * 36), pl = rem / 12, q = rem - pl * 12;
    const int ax = kMat[pl][0], bx = kMat[pl][1], vx = kVec[pl];
    const float fx = unnorm(xyz[ent * xs + ax], sc.gm1[ax]), fy = unnorm(xyz[ent * xs + bx], sc.gm1[bx]), fl = unnorm(xyz[ent * xs + vx], sc.gm1[vx]);
    const float x0 = floorf(fx), y0 = floorf(fy), l0 = floorf(fl);
    const VmTerm t = vm_eval<12>(sc.aplane[pl], sc.aline[pl], sc.grid[ax], (int)x0, (int)y0, (int)l0, fx - x0, fy - y0, fl - l0, q);
    *(float4 *)(h + ent * TVR_KAPP + pl * TVR_CA + q * 4) = make_float4(t.P.x * t.Q.x, t.P.y * t.Q.y, t.P.z * t.Q.z, t.P.w * t.Q.w);
}

// grid = (entry chunks, 3 planes): a workgroup owns AHB_ENTRIES entries of ONE plane/line pair and keeps that line's gradient in LDS.
// One WAVE per run of AHB_ENTRIES / 16 consecutive queue entries (consecutive appearance samples of a ray), one lane per CHANNEL (48 of the 64 lanes):
//   * an atomic instruction then covers whole 64-B lines of a texel — the fabric charges a no-return atomic per LINE request, 21 G requests/s chip-wide
//     whether the line carries 1, 4 or 16 dwords (scripts/hwprobe/atomic_rate.hip, profiles/r03_atomic_rate_probe.txt); a float4-per-lane layout, whose
//     .x/.y/.z/.w atomics each touch every line of the texel, was measured at 1.47 ms against 0.97;
//   * the entry's position is wave-uniform: scalar loads, scalar branches in the window logic;
//   * the plane contributions accumulate in a 2 x 2 tap WINDOW that slides with the ray (TapWin), the line contributions in a 2-tap window;
//   * 1024-thread workgroups: two fit a CU beside their 58 KB line images = 32 waves per CU (the first form, 192 threads, ran 6 waves per CU and was
//     latency-bound).
template <bool LINE_LDS>
__global__ __launch_bounds__(AHB_THREADS, 2) void app_h_backward_kernel(const SceneDev sc, const float *__restrict__ xyz, const int xs, const long long m_cap,
                                                                        const unsigned *__restrict__ m_dev, const float *__restrict__ dh, TrainGrads tg)
{
    extern __shared__ __attribute__((aligned(16))) float glds[];
    const long long m = m_dev ? ((long long)*m_dev < m_cap ? (long long)*m_dev : m_cap) : m_cap;
    if ((long long)blockIdx.x * AHB_ENTRIES >= m) return;           // (workgroup-uniform: a launch sized for the capacity)
    const int pl = blockIdx.y;
    const int ax = kMat[pl][0], bx = kMat[pl][1], vx = kVec[pl];
    const int ln = (sc.grid[vx] + 1) * TVR_CA;
    if (LINE_LDS) {
        for (int i = threadIdx.x; i < ln; i += AHB_THREADS) glds[i] = 0.0f;
        __syncthreads();
    }
    const long long e0 = (long long)blockIdx.x * AHB_ENTRIES;
    const long long e1 = e0 + AHB_ENTRIES < m ? e0 + AHB_ENTRIES : m;
    const float *__restrict__ Pl = (const float *)sc.aplane[pl];
    const float *__restrict__ Ln = (const float *)sc.aline[pl];
    float *__restrict__ gPl = tg.aplane[pl];
    float *gLn = LINE_LDS ? glds : tg.aline[pl];
    const int Wp = sc.grid[ax] + 1;
    const float gma = sc.gm1[ax], gmb = sc.gm1[bx], gmv = sc.gm1[vx];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = threadIdx.x & 63;
    constexpr long long per = AHB_ENTRIES / (AHB_THREADS / 64);
    const long long g0 = e0 + wave * per, g1 = (g0 + per < e1) ? g0 + per : e1;
    if (c < TVR_CA) {
        TapWin<TVR_CA> win;
        win.reset();
        int lk = 0;
        float b0 = 0.f, b1 = 0.f;
        for (long long ent = g0; ent < g1; ++ent) {
            const float fx = unnorm(xyz[ent * xs + ax], gma), fy = unnorm(xyz[ent * xs + bx], gmb), fl = unnorm(xyz[ent * xs + vx], gmv);
            const float x0f = floorf(fx), y0f = floorf(fy), l0f = floorf(fl);
            const float wx = fx - x0f, wy = fy - y0f, wl = fl - l0f, ux = 1.0f - wx, uy = 1.0f - wy, ul = 1.0f - wl;
            const int x0 = (int)x0f, y0 = (int)y0f, l0 = (int)l0f;
            const long long t00 = ((long long)y0 * Wp + x0) * TVR_CA + c, t10 = t00 + (long long)Wp * TVR_CA;
            const size_t q0 = (size_t)l0 * TVR_CA + c;
            float P = (ux * uy) * Pl[t00];
            P = __builtin_fmaf(wx * uy, Pl[t00 + TVR_CA], P);
            P = __builtin_fmaf(ux * wy, Pl[t10], P);
            P = __builtin_fmaf(wx * wy, Pl[t10 + TVR_CA], P);
            float Q = ul * Ln[q0];
            Q = __builtin_fmaf(wl, Ln[q0 + TVR_CA], Q);
            const float g = dh[ent * TVR_KAPP + pl * TVR_CA + c];
            const float gP = g * Q, gQ = g * P;                       // h = P*Q  ->  dP = g*Q, dQ = g*P
            win.move(gPl, Wp, c, x0, y0);
            win.a00 += (ux * uy) * gP;
            win.a01 += (wx * uy) * gP;
            win.a10 += (ux * wy) * gP;
            win.a11 += (wx * wy) * gP;
            const int dl = l0 - lk;
            if (dl != 0) {                                            // the line's 2-tap window
                if (dl == 1) { if (b0 != 0.0f) atomicAdd(gLn + (size_t)lk * TVR_CA + c, b0); b0 = b1; b1 = 0.0f; }
                else if (dl == -1) { if (b1 != 0.0f) atomicAdd(gLn + (size_t)(lk + 1) * TVR_CA + c, b1); b1 = b0; b0 = 0.0f; }
                else {
                    if (b0 != 0.0f) atomicAdd(gLn + (size_t)lk * TVR_CA + c, b0);
                    if (b1 != 0.0f) atomicAdd(gLn + (size_t)(lk + 1) * TVR_CA + c, b1);
                    b0 = b1 = 0.0f;
                }
                lk = l0;
            }
            b0 += ul * gQ;
            b1 += wl * gQ;
        }
        win.flush(gPl, Wp, c);
        if (b0 != 0.0f) atomicAdd(gLn + (size_t)lk * TVR_CA + c, b0);
        if (b1 != 0.0f) atomicAdd(gLn + (size_t)(lk + 1) * TVR_CA + c, b1);
    }
    if (LINE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < ln; i += AHB_THREADS) { const float v = glds[i]; if (v != 0.0f) atomicAdd(tg.aline[pl] + i, v); }
    }
}

// packed [H+1][Wp][C] gradient image -> reference (Cout,H,W) (a line: W == 1, Wp == 1); Cout <= C: the scene's own component count
__global__ __launch_bounds__(256) void unpack_grad_kernel(const float *__restrict__ in, float *__restrict__ out, int Cout, int C, int H, int W, int Wp)
{
    const long long total = (long long)Cout * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long t = i / W;
        const int y = (int)(t % H), c = (int)(t / H);
        out[i] = in[((size_t)y * Wp + x) * C + c];
    }
}

hipError_t launch_march_backward(const SceneDev &sc, const float *rays, int n_rays, int S, const MarchSampling &sm, float eps_T, const MarchOut &mo,
                                 const float *grad_w, const float *grad_acc, const float *lam6, const float *grad_lam6, const TrainGrads &tg,
                                 hipStream_t stream, long long gw_cap)
{
    const int rpb = TB_WAVES;
    const size_t stage = (size_t)TB_WAVES * TB_STAGE * sizeof(float);
    const size_t lds = stage + ((size_t)sc.grid[0] + sc.grid[1] + sc.grid[2] + 3) * TVR_CD * sizeof(float);
    const unsigned grid = (unsigned)((n_rays + rpb - 1) / rpb);
    if (lds <= 160 * 1024) {
        hipError_t rc = hipFuncSetAttribute((const void *)march_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(march_backward_kernel<true>, dim3(grid), dim3(TB_THREADS), lds, stream, sc, rays, n_rays, S, sm, eps_T, rpb, mo,
                           grad_w, grad_acc, lam6, grad_lam6, tg, gw_cap);
    } else {
        hipLaunchKernelGGL(march_backward_kernel<false>, dim3(grid), dim3(TB_THREADS), stage, stream, sc, rays, n_rays, S, sm, eps_T, rpb, mo,
                           grad_w, grad_acc, lam6, grad_lam6, tg, gw_cap);
    }
    return hipGetLastError();
}

hipError_t launch_app_h_forward(const SceneDev &sc, const float *xyz, long long m, float *h, hipStream_t stream, int xyz_stride, const unsigned *m_dev)
{
    hipLaunchKernelGGL(app_h_forward_kernel, dim3((unsigned)((m * 36 + 255) / 256)), dim3(256), 0, stream, sc, xyz, xyz_stride, m, m_dev, h);
    return hipGetLastError();
}

hipError_t launch_app_h_backward(const SceneDev &sc, const float *xyz, long long m, const float *dh, const TrainGrads &tg, hipStream_t stream, int xyz_stride,
                                 const unsigned *m_dev)
{
    int gmax = sc.grid[0] > sc.grid[1] ? sc.grid[0] : sc.grid[1];
    gmax = gmax > sc.grid[2] ? gmax : sc.grid[2];
    const size_t lds = ((size_t)gmax + 1) * TVR_CA * sizeof(float);
    const dim3 grid((unsigned)((m + AHB_ENTRIES - 1) / AHB_ENTRIES), 3);
    if (lds <= 150 * 1024) {
        hipError_t rc = hipFuncSetAttribute((const void *)app_h_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(app_h_backward_kernel<true>, grid, dim3(AHB_THREADS), lds, stream, sc, xyz, xyz_stride, m, m_dev, dh, tg);
    } else {
        hipLaunchKernelGGL(app_h_backward_kernel<false>, grid, dim3(AHB_THREADS), 0, stream, sc, xyz, xyz_stride, m, m_dev, dh, tg);
    }
    return hipGetLastError();
}

// planes: one workgroup per (row y, 64 columns), through an LDS tile: the reads run along the packed row, the writes along x of each channel (the element-per-thread
// kernel above reads with a stride of C floats per thread: 20 us per plane, 12 per training step)
#define UP_TILE 64
__global__ __launch_bounds__(256) void unpack_grad_tiled_kernel(const float *__restrict__ in, float *__restrict__ out, int Cout, int C, int H, int W, int Wp)
{
    __shared__ float tile[UP_TILE][TVR_CA + 1];
    const int y = blockIdx.y, x0 = blockIdx.x * UP_TILE;
    for (int k = threadIdx.x; k < C * UP_TILE; k += 256) {
        const int xx = k / C, c = k - xx * C, x = x0 + xx;
        tile[xx][c] = x < W ? in[((size_t)y * Wp + x) * C + c] : 0.0f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < Cout * UP_TILE; k += 256) {
        const int c = k / UP_TILE, xx = k - c * UP_TILE, x = x0 + xx;
        if (x < W) out[((size_t)c * H + y) * W + x] = tile[xx][c];
    }
}

hipError_t launch_unpack_grad(const float *in, float *out, int Cout, int C, int H, int W, hipStream_t stream)
{
    const int Wp = (W == 1) ? 1 : W + 1;
    if (W > 1 && C <= TVR_CA) {
        hipLaunchKernelGGL(unpack_grad_tiled_kernel, dim3((unsigned)((W + UP_TILE - 1) / UP_TILE), (unsigned)H), dim3(256), 0, stream, in, out, Cout, C, H, W, Wp);
        return hipGetLastError();
    }
    const long long total = (long long)Cout * H * W;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(unpack_grad_kernel, dim3(grid), dim3(256), 0, stream, in, out, Cout, C, H, W, Wp);
    return hipGetLastError();
}

// ---- fused training helpers: the MLP input assembly and the TV regulariser (each replaces a dozen elementwise library launches) ----------
// MLPRender_Fea's input (tensorBase.py:76-82): X = [f (27), d (3), sin(f 2^k), cos(f 2^k) (k < 2, index 2c + k), sin(d 2^k), cos(d 2^k)];
// MLPRender_Fea_Ref (REFTensoRF.py:19-24) puts dot_product in front (with_dot = 1, 151 columns).  One thread per (entry, base value).
// Strided sources (the fused training step reads them where the forward kernel left them): feat row stride fs, dir row stride ds (or, with q_ray, the
// direction of entry e is rays[q_ray[e]][3..5]), dot row stride dts; m_dev as in app_h_forward_kernel.
struct PeSrc { const float *feat, *dir, *dot, *rays; const unsigned *q_ray, *m_dev; int fs, ds, dts; };
// A workgroup builds the rows of 32 entries in LDS and writes them as ONE contiguous run of float4s (32 rows of 150 / 151 floats start 16-B aligned):
// the first form stored five scattered floats per thread (0.19 ms per step of the real loop, store-path-bound).
// FAST: sin / cos as the shade kernel's layer 1 takes them (sincos of v through the hardware unit behind a Cody-Waite reduction, the double angle by
// 2 s c and 1 - 2 s^2) — the fused training step, whose X must be the forward kernel's own layer-1 input; otherwise libm's sincosf (the C-ABI entry point).
#define PE_TILE 32
template <bool FAST>
__global__ __launch_bounds__(256) void pe_concat_forward_kernel(const PeSrc p, const long long m_cap, const int with_dot, float *__restrict__ X)
{
    __shared__ __attribute__((aligned(16))) float tile[PE_TILE * 152];
    const long long m = p.m_dev ? ((long long)*p.m_dev < m_cap ? (long long)*p.m_dev : m_cap) : m_cap;
    const long long e0 = (long long)blockIdx.x * PE_TILE;
    if (e0 >= m) return;
    const int rows = (int)(m - e0 < PE_TILE ? m - e0 : PE_TILE), n = 150 + with_dot;
    const float *__restrict__ feat = p.feat, *__restrict__ dot = p.dot;
    for (int it = threadIdx.x; it < rows * 30; it += 256) {
        const int r = it / 30, c = it - r * 30;
        const long long ent = e0 + r;
        float *x = tile + r * n + with_dot;
        const float v = c < 27 ? feat[ent * p.fs + c] : (p.q_ray ? p.rays[(size_t)p.q_ray[ent] * 6 + 3 + (c - 27)] : p.dir[ent * p.ds + (c - 27)]);
        float s1, c1, s2, c2;
        if (FAST) {
            const float k = rintf(v * 0.15915494309189535f);
            float rr = __builtin_fmaf(k, -6.2831854820251465f, v);
            rr = __builtin_fmaf(k, 1.7484555e-7f, rr);
            const float t = rr * 0.15915494309189535f;
            s1 = __builtin_amdgcn_sinf(t);
            c1 = __builtin_amdgcn_cosf(t);
            s2 = 2.0f * s1 * c1;
            c2 = __builtin_fmaf(-2.0f * s1, s1, 1.0f);
        } else {
            sincosf(v, &s1, &c1);
            sincosf(v * 2.0f, &s2, &c2);
        }
        x[c] = v;
        if (c < 27) {
            x[30 + 2 * c] = s1; x[31 + 2 * c] = s2; x[84 + 2 * c] = c1; x[85 + 2 * c] = c2;
        } else {
            const int j = c - 27;
            x[138 + 2 * j] = s1; x[139 + 2 * j] = s2; x[144 + 2 * j] = c1; x[145 + 2 * j] = c2;
        }
        if (with_dot && c == 0) tile[r * n] = dot[ent * p.dts];
    }
    __syncthreads();
    float *__restrict__ dst = X + e0 * n;
    const int total = rows * n;
    if (!((uintptr_t)dst & 15)) {
        const int n4 = total >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) ((float4 *)dst)[i] = ((const float4 *)tile)[i];
        for (int i = (n4 << 2) + threadIdx.x; i < total; i += 256) dst[i] = tile[i];
    } else {
        for (int i = threadIdx.x; i < total; i += 256) dst[i] = tile[i];
    }
}

// d/dv of the five columns a base value feeds: gX[v] + sum_k 2^k (gX[sin] cos(v 2^k) - gX[cos] sin(v 2^k))
__global__ __launch_bounds__(256) void pe_concat_backward_kernel(const float *__restrict__ feat, const float *__restrict__ dir,
                                                                 const float *__restrict__ gX, const long long m, const int with_dot,
                                                                 float *__restrict__ gfeat, float *__restrict__ gdir, float *__restrict__ gdot)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * 30) return;
    const long long ent = t / 30;
    const int c = (int)(t - ent * 30);
    const int n = 150 + with_dot;
    const float *g = gX + ent * n + with_dot;
    const float v = c < 27 ? feat[ent * 27 + c] : dir[ent * 3 + (c - 27)];
    float s1, c1, s2, c2;
    sincosf(v, &s1, &c1);
    sincosf(v * 2.0f, &s2, &c2);
    const int is = c < 27 ? 30 + 2 * c : 138 + 2 * (c - 27), ic = c < 27 ? 84 + 2 * c : 144 + 2 * (c - 27);
    const float r = g[c] + (g[is] * c1 - g[ic] * s1) + 2.0f * (g[is + 1] * c2 - g[ic + 1] * s2);
    if (c < 27) gfeat[ent * 27 + c] = r;
    else if (gdir) gdir[ent * 3 + (c - 27)] = r;
    if (with_dot && c == 0 && gdot) gdot[ent] = gX[ent * n];
}

hipError_t launch_pe_concat(const float *feat, const float *dir, const float *dot, long long m, float *X, hipStream_t stream)
{
    PeSrc p = {feat, dir, dot, nullptr, nullptr, nullptr, 27, 3, 1};
    hipLaunchKernelGGL(pe_concat_forward_kernel<false>, dim3((unsigned)((m + PE_TILE - 1) / PE_TILE)), dim3(256), 0, stream, p, m, dot ? 1 : 0, X);
    return hipGetLastError();
}

hipError_t launch_pe_concat_strided(const float *feat, int fs, const float *dir, int ds, const float *rays, const unsigned *q_ray, const float *dot, int dts,
                                    long long m_cap, const unsigned *m_dev, float *X, hipStream_t stream)
{
    PeSrc p = {feat, dir, dot, rays, q_ray, m_dev, fs, ds, dts};
    hipLaunchKernelGGL(pe_concat_forward_kernel<true>, dim3((unsigned)((m_cap + PE_TILE - 1) / PE_TILE)), dim3(256), 0, stream, p, m_cap, dot ? 1 : 0, X);
    return hipGetLastError();
}

// Round 6 — the MLP input of a scene with more than two encoding frequencies (tensorBase.py:76-82 with view_pe / fea_pe up to 6: 30 + 54 fea_pe + 6 view_pe columns, 390
// at the constructor's defaults), for the weight gradient dW1 = dH1^T X of the fused training step.  X is written as COLUMN BLOCKS of TVR_GENX_W columns, block b a
// contiguous [m_cap, w_b] matrix at X + b * m_cap * TVR_GENX_W (w_b = the block's columns rounded up to 4, zero-filled): each block is then one tvr_gemm_tn product in its
// fastest staging mode (contiguous 16-B rows, <= 5 column tiles with the bias column).  sin / cos exactly as the forward kernel's lockstep layer 1 takes them
// (tvr_shade.hip gen_frag): the hardware units on the once-reduced argument times 2^f.
// (PEG_TILE entries per workgroup: 8 — one entry x base value per thread in ONE round, 14.6 KB of LDS, ten workgroups per CU; with the 2 / 2 kernel's 32 entries, 50 KB and four
//  rounds of dependent loads per thread the kernel took 0.27 - 0.30 ms per step for the 0.11 ms its 550 MB of stores need)
#define PEG_TILE 8
__global__ __launch_bounds__(256) void pe_concat_gen_kernel(const PeSrc p, const long long m_cap, const int fea_pe, const int view_pe, float *__restrict__ X)
{
    extern __shared__ __attribute__((aligned(16))) float gtile[];            // [block][PEG_TILE rows][w_b]
    const long long m = p.m_dev ? ((long long)*p.m_dev < m_cap ? (long long)*p.m_dev : m_cap) : m_cap;
    const long long e0 = (long long)blockIdx.x * PEG_TILE;
    if (e0 >= m) return;
    const int rows = (int)(m - e0 < PEG_TILE ? m - e0 : PEG_TILE);
    const int nin = TVR_APPDIM + 3 + 2 * TVR_APPDIM * fea_pe + 6 * view_pe, nb = (nin + TVR_GENX_W - 1) / TVR_GENX_W;
    const int wl = ((nin - (nb - 1) * TVR_GENX_W) + 3) & ~3;                  // width of the last block
    // (every real column of a row is written below; only the last block's zero columns need a value — clearing the whole 50 KB tile first cost as much as filling it)
    const int npad = wl - (nin - (nb - 1) * TVR_GENX_W);
    for (int i = threadIdx.x; i < PEG_TILE * npad; i += 256) gtile[(nb - 1) * PEG_TILE * TVR_GENX_W + (i / npad) * wl + (wl - npad) + (i % npad)] = 0.0f;
    for (int it = threadIdx.x; it < rows * 30; it += 256) {
        const int r = it / 30, c = it - r * 30;
        const long long ent = e0 + r;
        const float v = c < 27 ? p.feat[ent * p.fs + c] : (p.q_ray ? p.rays[(size_t)p.q_ray[ent] * 6 + 3 + (c - 27)] : p.dir[ent * p.ds + (c - 27)]);
        const float k = rintf(v * 0.15915494309189535f);
        float rr = __builtin_fmaf(k, -6.2831854820251465f, v);
        rr = __builtin_fmaf(k, 1.7484555e-7f, rr);
        const float tr = rr * 0.15915494309189535f;
#pragma unroll
        for (int t = 0; t < TVR_GEN_T; ++t) {
            const int idx = gen_in_index(c, t, fea_pe, view_pe);
            if (idx < 0) continue;
            const int f = t == 0 ? 0 : (t <= TVR_GEN_PE ? t - 1 : t - 1 - TVR_GEN_PE);
            const float val = t == 0 ? v : (t <= TVR_GEN_PE ? __builtin_amdgcn_sinf(tr * (float)(1 << f)) : __builtin_amdgcn_cosf(tr * (float)(1 << f)));
            const int b = idx / TVR_GENX_W, col = idx - b * TVR_GENX_W, wb = b == nb - 1 ? wl : TVR_GENX_W;
            gtile[b * PEG_TILE * TVR_GENX_W + r * wb + col] = val;
        }
    }
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int wb = b == nb - 1 ? wl : TVR_GENX_W;
        float4 *__restrict__ dst = (float4 *)(X + (size_t)b * (size_t)m_cap * TVR_GENX_W + (size_t)e0 * wb);       // 16-B aligned: m_cap * 152 * 4 and PEG_TILE * wb * 4 are multiples of 16
        const float4 *src = (const float4 *)(gtile + b * PEG_TILE * TVR_GENX_W);
        for (int i = threadIdx.x; i < rows * wb / 4; i += 256) dst[i] = src[i];
    }
}

hipError_t launch_pe_concat_gen(const float *feat, int fs, const float *rays, const unsigned *q_ray, int fea_pe, int view_pe, long long m_cap, const unsigned *m_dev, float *X,
                                hipStream_t stream)
{
    PeSrc p = {feat, nullptr, nullptr, rays, q_ray, m_dev, fs, 0, 0};
    const int nin = TVR_APPDIM + 3 + 2 * TVR_APPDIM * fea_pe + 6 * view_pe, nb = (nin + TVR_GENX_W - 1) / TVR_GENX_W;
    const int lds = PEG_TILE * nb * TVR_GENX_W * (int)sizeof(float);          // <= 14 592 B
    hipLaunchKernelGGL(pe_concat_gen_kernel, dim3((unsigned)((m_cap + PEG_TILE - 1) / PEG_TILE)), dim3(256), lds, stream, p, m_cap, fea_pe, view_pe, X);
    return hipGetLastError();
}

// dst[r * ldd + col0 + c] = src[r * lds + c], c < ncols, r < nrows (a column block of a weight gradient into its place)
__global__ __launch_bounds__(256) void copy_cols_kernel(float *__restrict__ dst, const int ldd, const int col0, const float *__restrict__ src, const int lds, const int ncols, const int nrows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols * nrows) return;
    const int r = i / ncols, c = i - r * ncols;
    dst[(size_t)r * ldd + col0 + c] = src[(size_t)r * lds + c];
}

hipError_t launch_copy_cols(float *dst, int ldd, int col0, const float *src, int lds, int ncols, int nrows, hipStream_t stream)
{
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((ncols * nrows + 255) / 256)), dim3(256), 0, stream, dst, ldd, col0, src, lds, ncols, nrows);
    return hipGetLastError();
}

hipError_t launch_pe_concat_backward(const float *feat, const float *dir, const float *gX, long long m, int with_dot, float *gfeat, float *gdir,
                                     float *gdot, hipStream_t stream)
{
    hipLaunchKernelGGL(pe_concat_backward_kernel, dim3((unsigned)((m * 30 + 255) / 256)), dim3(256), 0, stream, feat, dir, gX, m, with_dot, gfeat,
                       gdir, gdot);
    return hipGetLastError();
}

// TVLoss (tensorf-myc/utils.py:123-142) of one plane x (C,H,W): value = 2 (h_tv / count_h + w_tv / count_w) with h_tv = sum (x[.,y+1,.] - x[.,y,.])^2,
// w_tv likewise along x, count_h = C (H-1) W, count_w = C H (W-1); and its gradient, in one pass.  Partial sums go to `part` (one float per
// workgroup, summed in order by tv_finish_kernel): deterministic.
#define TV_THREADS 256
// VEC: W is a multiple of 4 — a thread takes four consecutive elements of a row (float4 loads of the row and of its two neighbours, one index division per
// four elements).  Round 3: the first form divided a 64-bit index twice per element and ran at 1.1 TB/s; 32-bit indices (a plane has < 2^31 elements).
template <bool VEC>
__global__ __launch_bounds__(TV_THREADS) void tv_loss_kernel(const float *__restrict__ x, const int C, const int H, const int W,
                                                             const float ch, const float cw, float *__restrict__ grad, float *__restrict__ part)
{
    __shared__ float red[TV_THREADS / 64];
    const int total = C * H * W;
    float acc = 0.0f;
    if (VEC) {
        const int W4 = W >> 2, n4 = total >> 2;
        for (int q = blockIdx.x * TV_THREADS + threadIdx.x; q < n4; q += gridDim.x * TV_THREADS) {
            const int row = q / W4, x4 = q - row * W4, yy = row % H, i = q << 2;
            const float4 v = *(const float4 *)(x + i);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            float dn[4] = {0.f, 0.f, 0.f, 0.f}, up[4] = {0.f, 0.f, 0.f, 0.f};
            const bool hd = yy + 1 < H, hu = yy > 0;
            if (hd) { const float4 t = *(const float4 *)(x + i + W); dn[0] = t.x; dn[1] = t.y; dn[2] = t.z; dn[3] = t.w; }
            if (hu) { const float4 t = *(const float4 *)(x + i - W); up[0] = t.x; up[1] = t.y; up[2] = t.z; up[3] = t.w; }
            const bool hl = x4 > 0, hr = x4 + 1 < W4;
            const float left = hl ? x[i - 1] : 0.0f, right = hr ? x[i + 4] : 0.0f;
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ge = 0.0f;
                if (hd) { const float d = dn[e] - vv[e]; acc += ch * d * d; ge -= 2.0f * ch * d; }
                if (hu) ge += 2.0f * ch * (vv[e] - up[e]);
                const bool has_r = e < 3 || hr, has_l = e > 0 || hl;
                if (has_r) { const float d = (e < 3 ? vv[e < 3 ? e + 1 : 3] : right) - vv[e]; acc += cw * d * d; ge -= 2.0f * cw * d; }
                if (has_l) ge += 2.0f * cw * (vv[e] - (e > 0 ? vv[e > 0 ? e - 1 : 0] : left));
                g[e] = ge;
            }
            *(float4 *)(grad + i) = make_float4(g[0], g[1], g[2], g[3]);
        }
    } else {
        for (int i = blockIdx.x * TV_THREADS + threadIdx.x; i < total; i += gridDim.x * TV_THREADS) {
            const int r = i / W, xx = i - r * W, yy = r % H;
            const float v = x[i];
            float g = 0.0f;
            if (yy + 1 < H) { const float d = x[i + W] - v; acc += ch * d * d; g -= 2.0f * ch * d; }
            if (yy > 0) g += 2.0f * ch * (v - x[i - W]);
            if (xx + 1 < W) { const float d = x[i + 1] - v; acc += cw * d * d; g -= 2.0f * cw * d; }
            if (xx > 0) g += 2.0f * cw * (v - x[i - 1]);
            grad[i] = g;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < TV_THREADS / 64; ++w) s += red[w];
        part[blockIdx.x] = s;
    }
}

// one wave: lane l adds partials l, l+64, ... in order, then a fixed butterfly
__global__ __launch_bounds__(64) void tv_finish_kernel(const float *__restrict__ part, const int n, float *__restrict__ out)
{
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) out[0] = s;
}

#define TV_BLOCKS 512
hipError_t launch_tv_loss(const float *x, int C, int H, int W, float weight, float *value, float *grad, float *part, hipStream_t stream)
{
    const float count_h = (float)C * (float)(H - 1) * (float)W, count_w = (float)C * (float)H * (float)(W - 1);
    const float ch = (H > 1) ? weight * 2.0f / count_h : 0.0f, cw = (W > 1) ? weight * 2.0f / count_w : 0.0f;
    if ((long long)C * H * W >= (1ll << 31)) return hipErrorInvalidValue;
    if (!(W & 3) && !((uintptr_t)x & 15) && !((uintptr_t)grad & 15)) hipLaunchKernelGGL(tv_loss_kernel<true>, dim3(TV_BLOCKS), dim3(TV_THREADS), 0, stream, x, C, H, W, ch, cw, grad, part);
    else hipLaunchKernelGGL(tv_loss_kernel<false>, dim3(TV_BLOCKS), dim3(TV_THREADS), 0, stream, x, C, H, W, ch, cw, grad, part);
    hipLaunchKernelGGL(tv_finish_kernel, dim3(1), dim3(64), 0, stream, part, TV_BLOCKS, value);
    return hipGetLastError();
}
