// tvr_reg.hip — the two parameter-only regularisers of the training loop that are not TV (tensorf-myc/train.py:237-244), each as ONE launch forward and ONE backward:
//   L1 of the density factors   TensorVMSplit.density_L1          (tensoRF.py:190-194):  sum_t mean |x_t|  over the 3 planes and 3 lines
//   line orthogonality          TensorVMSplit.vector_comp_diffs   (tensoRF.py:178-188):  sum_t mean |offdiag(V_t V_t^T)|  over the 6 line factors
// As torch expressions these were ~100 small kernels per step (abs / mean / sign / mul / add chains and the autograd that links them) — a quarter of a
// millisecond of a 4 ms step.  All sums run in a fixed order (deterministic).  The backward takes the upstream gradient from the DEVICE (no host read).
#include "tvr_device.h"
#include "tvr_kernels.h"

#define REG_THREADS 256
#define L1_CHUNK 4096                    // elements per workgroup of the L1 forward (one partial sum each; 16384 until round 6: 264 workgroups for the bench scene, 31 + 24 us)

__device__ __forceinline__ float block_sum_256(float v, float *sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;                            // valid in thread 0
}

// forward, stage 1: partial[b] = sum |x| over chunk b of tensor t (blocks are laid out tensor after tensor)
__global__ __launch_bounds__(REG_THREADS) void l1_partial_kernel(const RegList L, float *__restrict__ partial)
{
    __shared__ float sh[4];
    int b = blockIdx.x, t = 0;
    while (t < L.n - 1 && b >= L.blocks[t]) { b -= L.blocks[t]; ++t; }
    const float *__restrict__ x = L.x[t];
    const long long n = L.count[t], i0 = (long long)b * L1_CHUNK, i1 = i0 + L1_CHUNK < n ? i0 + L1_CHUNK : n;
    float acc = 0.0f;
    for (long long i = i0 + threadIdx.x; i < i1; i += REG_THREADS) acc += fabsf(x[i]);
    const float s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// forward, stage 2 (one workgroup): value = sum_t (sum of tensor t's partials, in order) / count_t
__global__ __launch_bounds__(REG_THREADS) void l1_final_kernel(const RegList L, const float *__restrict__ partial, float *__restrict__ value)
{
    __shared__ float sh[4];
    float total = 0.0f;
    int base = 0;
    for (int t = 0; t < L.n; ++t) {
        float acc = 0.0f;
        for (int i = threadIdx.x; i < L.blocks[t]; i += REG_THREADS) acc += partial[base + i];
        const float s = block_sum_256(acc, sh);
        if (threadIdx.x == 0) total += s / (float)L.count[t];
        base += L.blocks[t];
    }
    if (threadIdx.x == 0) value[0] = total;
}

// backward: grad_t = g * sign(x_t) / count_t   (torch.abs' gradient: sign(0) = 0)
__global__ __launch_bounds__(REG_THREADS) void l1_backward_kernel(const RegList L, const float *__restrict__ g)
{
    int b = blockIdx.x, t = 0;
    while (t < L.n - 1 && b >= L.blocks[t]) { b -= L.blocks[t]; ++t; }
    const float *__restrict__ x = L.x[t];
    float *__restrict__ gr = L.grad[t];
    const long long n = L.count[t], i0 = (long long)b * L1_CHUNK, i1 = i0 + L1_CHUNK < n ? i0 + L1_CHUNK : n;
    const float s = g[0] / (float)n;
    for (long long i = i0 + threadIdx.x; i < i1; i += REG_THREADS) {
        const float v = x[i];
        gr[i] = v > 0.0f ? s : (v < 0.0f ? -s : 0.0f);
    }
}

// line orthogonality: ORTHO_SPLIT workgroups per line factor V [nc, ns] (nc <= 48), workgroup (t, rb) owning rows i = rb, rb + ORTHO_SPLIT, ... of G = V V^T (round 6: one
// workgroup per factor — six workgroups on a 256-CU chip — took 107 us per launch, forward and backward: 0.21 ms of a 3.4 ms training step).  Every G_ij is a fixed-order
// sum over ns (lanes over k, butterfly), so G_ij == G_ji bit for bit whichever workgroup computes it;
// value_t = sum_{i != j} |G_ij| / (nc (nc - 1));  d value_t / d V_i = 2 / (nc (nc - 1)) * sum_{j != i} sign(G_ij) V_j.
// g == nullptr: forward (part[t * ORTHO_SPLIT + rb] = the rows' share of value_t; ortho_final_kernel adds them in order); else backward with the upstream gradient g[0].
#define ORTHO_MAXC 48
#define ORTHO_SPLIT 8
#define ORTHO_ROWS ((ORTHO_MAXC + ORTHO_SPLIT - 1) / ORTHO_SPLIT)
#define ORTHO_LDS_FLOATS (ORTHO_MAXC * 320)                           // V staged in LDS when it fits (48 x 301 does): every element is read ~100 times
__global__ __launch_bounds__(1024) void ortho_kernel(const RegList L, const float *__restrict__ g, float *__restrict__ part)
{
    __shared__ float G[ORTHO_ROWS * ORTHO_MAXC];                      // the rows this workgroup owns: G[q * nc + j], row i = rb + q * ORTHO_SPLIT
    __shared__ float red[16];
    __shared__ float Vs[ORTHO_LDS_FLOATS];
    const int t = blockIdx.x / ORTHO_SPLIT, rb = blockIdx.x - t * ORTHO_SPLIT;
    const float *__restrict__ Vg = L.x[t];
    const int nc = L.rows[t], ns = (int)(L.count[t] / nc);
    const int nq = rb < nc ? (nc - rb + ORTHO_SPLIT - 1) / ORTHO_SPLIT : 0;        // rows of this workgroup
    const bool in_lds = nc * ns <= ORTHO_LDS_FLOATS;                  // (round 3: from global memory the six workgroups took 136 us per launch, latency-bound)
    if (in_lds) {
        for (int e = threadIdx.x; e < nc * ns; e += 1024) Vs[e] = Vg[e];
        __syncthreads();
    }
    const float *V = in_lds ? (const float *)Vs : Vg;
    // G_ij: one wave per (row, j) pair in turn, lanes over ns, fixed butterfly
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int p = wave; p < nq * nc; p += 16) {
        const int q = p / nc, j = p - q * nc, i = rb + q * ORTHO_SPLIT;
        // (the products of pair (i, j) are summed in the order of the SMALLER index's row first — a * b == b * a, so either order of the operands gives the same sum)
        float acc = 0.0f;
        for (int k = lane; k < ns; k += 64) acc += V[(size_t)i * ns + k] * V[(size_t)j * ns + k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) G[q * nc + j] = acc;
    }
    __syncthreads();
    const float inv = 1.0f / (float)(nc * (nc - 1));
    if (g == nullptr) {
        float acc = 0.0f;
        for (int p = threadIdx.x; p < nq * nc; p += 1024) {
            const int q = p / nc, j = p - q * nc;
            if (rb + q * ORTHO_SPLIT != j) acc += fabsf(G[p]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.0f;
            for (int w = 0; w < 16; ++w) s += red[w];
            part[blockIdx.x] = s * inv;
        }
    } else {
        float *__restrict__ gr = L.grad[t];
        const float s = 2.0f * inv * g[0];
        for (int e = threadIdx.x; e < nq * ns; e += 1024) {
            const int q = e / ns, k = e - q * ns, i = rb + q * ORTHO_SPLIT;
            float acc = 0.0f;
            for (int j = 0; j < nc; ++j) {
                if (j == i) continue;
                const float gij = G[q * nc + j];
                const float sg = gij > 0.0f ? 1.0f : (gij < 0.0f ? -1.0f : 0.0f);
                acc += sg * V[(size_t)j * ns + k];
            }
            gr[(size_t)i * ns + k] = s * acc;
        }
    }
}

__global__ void ortho_final_kernel(const float *__restrict__ part, int n, float *__restrict__ value)
{
    float s = 0.0f;
    for (int t = 0; t < n; ++t) s += part[t];
    value[0] = s;
}

static void fill_blocks(RegList &L, int &total)
{
    total = 0;
    for (int t = 0; t < L.n; ++t) {
        L.blocks[t] = (int)((L.count[t] + L1_CHUNK - 1) / L1_CHUNK);
        total += L.blocks[t];
    }
}

size_t reg_l1_scratch_bytes(const RegList &Lin)
{
    RegList L = Lin;
    int total;
    fill_blocks(L, total);
    return (size_t)total * sizeof(float);
}

hipError_t launch_l1_forward(const RegList &Lin, float *value, float *scratch, hipStream_t stream)
{
    RegList L = Lin;
    int total;
    fill_blocks(L, total);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(total), dim3(REG_THREADS), 0, stream, L, scratch);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(REG_THREADS), 0, stream, L, scratch, value);
    return hipGetLastError();
}

hipError_t launch_l1_backward(const RegList &Lin, const float *g, hipStream_t stream)
{
    RegList L = Lin;
    int total;
    fill_blocks(L, total);
    hipLaunchKernelGGL(l1_backward_kernel, dim3(total), dim3(REG_THREADS), 0, stream, L, g);
    return hipGetLastError();
}

hipError_t launch_ortho(const RegList &L, const float *g, float *value, float *scratch, hipStream_t stream)
{
    hipLaunchKernelGGL(ortho_kernel, dim3(L.n * ORTHO_SPLIT), dim3(1024), 0, stream, L, g, scratch);
    if (!g) hipLaunchKernelGGL(ortho_final_kernel, dim3(1), dim3(1), 0, stream, scratch, L.n * ORTHO_SPLIT, value);
    return hipGetLastError();
}
