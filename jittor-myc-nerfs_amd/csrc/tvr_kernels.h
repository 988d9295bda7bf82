// tvr_kernels.h — host-visible launcher declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tvr.h"

struct SceneDev;

// records the text tvr_last_error() returns (thread-local) and hands `code` back; defined in tvr_api.hip
int tvr_set_error(int code, const char *fmt, ...);

// Outputs of the march kernel / inputs of shade + composite.  The "queue" holds one entry per appearance sample
// (weight > thres); each ray's entries are contiguous and in sample order.
struct MarchOut {
    unsigned *counter;        // queue length (zeroed by the host before the march)
    unsigned *ray_off;        // [n_rays] first entry of the ray
    unsigned *ray_cnt;        // [n_rays] entries of the ray
    float *acc;               // [n_rays] sum of ALL weights (tensorBase.py:520)
    float *depth;             // [n_rays] final depth_map (written by the march kernel)
    float4 *q_pos;            // [cap] {xyz_norm, weight}
    float4 *q_out;            // [cap] {rgb, weight} written by the shade kernel (may alias q_pos)
    unsigned *q_ray;          // [cap] ray index (for the view direction)
    unsigned *q_j;            // [cap] sample index, or nullptr
    unsigned long long *stats;// TVR_STAT_* counters or nullptr
    float *lam6;              // [n_rays] prod_j (1 - alpha_j + 1e-6) (NerfPlusPlus bg_lambda, nerfplusplus.py:277-278) or nullptr
};

// how the samples of a ray are placed: uniform steps from the box entry (+ one jitter per ray), or explicit depths [n,S]
struct MarchSampling {
    const float *jitter;      // [n_rays] or nullptr
    const float *zv;          // [n_rays,S] or nullptr (NerfPlusPlus.sample_ray, nerfplusplus.py:239-269)
};

// packed (channels-last, zero-padded) gradient images of the VM factors, same geometry as the packed scene
struct TrainGrads {
    float *dplane[3], *dline[3], *aplane[3], *aline[3];
};

enum { SH_SRC_QUEUE = 0, SH_SRC_XYZ = 1, SH_SRC_FEAT = 2, SH_SRC_H = 3 };       // SRC_H: h [n,144] given (training forward)
enum { SH_DST_QUEUE = 0, SH_DST_FEAT = 1, SH_DST_RGB = 2, SH_DST_TRAIN = 3 };    // DST_TRAIN: rgb + the activations the backward needs

struct ShadeArgs {
    const unsigned *counter;   // SRC_QUEUE: entry count lives on the device
    long long n;               // other modes: entry count
    float4 *q_pos;
    float4 *q_out;
    const unsigned *q_ray;
    const float *rays;         // [n_rays,6]
    const float *xyz;          // SRC_XYZ: xyz_norm [n,3]
    const float *viewdirs;     // SRC_FEAT: [n,3]
    const float *feats;        // SRC_FEAT: [n,27]
    float *out;                // DST_FEAT [n,27] / DST_RGB [n,3]
    const float *dots;         // REFTensoRF, SRC_FEAT: the dot_product input of MLPRender_Fea_Ref [n]
    float *out2;               // REFTensoRF, DST_FEAT: [n,8] {normal 3, rgb_d 3, specular_tint, rho}
    const float *h_in;         // SRC_H: h [n,144] (tvr_app_h_forward)
    float *t_feats, *t_h1, *t_h2;   // DST_TRAIN: features [n,32] (27 + zero pad), relu(layer 1) [n,128], relu(layer 2) [n,128]
    float *t_g8, *t_rgbs;           // DST_TRAIN, REFTensoRF: raw head outputs [n,8] {normal 3, tint, rgb_d 3, rho}, the MLP's sigmoid output [n,3]
    unsigned long long *stats;
};

hipError_t launch_march(const SceneDev &sc, const float *rays, int n_rays, int S, const MarchSampling &sm, float eps_T,
                        const MarchOut &mo, const tvr_dense_out *dense, hipStream_t stream);
hipError_t launch_composite(const MarchOut &mo, int n_rays, int white_bg, float *rgb, hipStream_t stream);
hipError_t launch_scatter_rgb(const MarchOut &mo, int S, float *rgb_dense, hipStream_t stream);
hipError_t launch_density_feature(const SceneDev &sc, const float *xyz, long long m, float *out, hipStream_t stream);
hipError_t launch_alpha_sample(const SceneDev &sc, const float *xyz, long long m, float *out, hipStream_t stream);
hipError_t launch_shade(const SceneDev &sc, int src, int dst, const ShadeArgs &a, hipStream_t stream);
// tvr_shade16.hip: the render path (queue -> queue, TensorVMSplit, default arithmetic, at most two encoding frequencies) on 16x16x32 tiles, and its fragment images
hipError_t launch_shade16(const SceneDev &sc, const ShadeArgs &a, hipStream_t stream);
struct MlpShape;
hipError_t launch_pack16(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3, const float *basis, void *img, void *basg,
                         const MlpShape &sh, hipStream_t stream, const float *const *ref_W = nullptr, const float *const *ref_b = nullptr, void *refg = nullptr);
hipError_t launch_pack_plane(const float *in, float *out, int Cin, int C, int H, int W, hipStream_t stream);
// the scene's MLP_Fea / basis shape as the reference holds it (<= the shape the kernels are built for; packed with zero padding)
struct MlpShape {
    int featureC, fea_pe, view_pe, n_in;          // n_in = 30 + 54 fea_pe + 6 view_pe (+ 1 for REFTensoRF)
    int app_n_comp[3], app_off[3], k_app;         // basis_mat is [27][k_app], k_app = sum app_n_comp
};
hipError_t launch_pack_mlp(const float *W, const float *bias, void *out_hi, void *out_lo, int mode, const MlpShape &sh, hipStream_t stream);
hipError_t launch_pack_ref(const float *const W[4], const float *const b[4], void *rows, float *bias, hipStream_t stream);
hipError_t launch_march_backward(const SceneDev &sc, const float *rays, int n_rays, int S, const MarchSampling &sm, float eps_T, const MarchOut &mo,
                                 const float *grad_w, const float *grad_acc, const float *lam6, const float *grad_lam6, const TrainGrads &tg,
                                 hipStream_t stream, long long gw_cap = -1);    // gw_cap >= 0: grad_w holds gw_cap entries; a step whose queue is longer (or faulted) scatters nothing
// xyz_stride: 3 (xyz [m,3]) or 4 (the march queue's {xyz, w}); m_dev: optional device-side entry count, m is then the capacity (min of the two is processed)
hipError_t launch_app_h_forward(const SceneDev &sc, const float *xyz, long long m, float *h, hipStream_t stream, int xyz_stride = 3, const unsigned *m_dev = nullptr);
hipError_t launch_app_h_backward(const SceneDev &sc, const float *xyz, long long m, const float *dh, const TrainGrads &tg, hipStream_t stream, int xyz_stride = 3,
                                 const unsigned *m_dev = nullptr);
hipError_t launch_unpack_grad(const float *in, float *out, int Cout, int C, int H, int W, hipStream_t stream);
// m_dev: optional device-side row count; M is then the capacity (the grid is sized for it, the slabs are cut from min(*m_dev, M) on the device)
hipError_t launch_gemm_tn(const float *A, int lda, int Ka, const float *B, int ldb, int Kb, long long M, float *C, float *scratch, hipStream_t stream,
                          const unsigned *m_dev = nullptr, float *bias_out = nullptr,       // bias_out [Ka]: also the column sums of A (scratch sized for Kb + 1)
                          const float *scale_f16 = nullptr);   // device scalar s (a power of two): products on the fp16-split MFMAs with A * s (operands known to fit fp16 at that scale)
size_t gemm_tn_scratch_bytes(int Ka, int Kb, long long M);
hipError_t launch_pe_concat(const float *feat, const float *dir, const float *dot, long long m, float *X, hipStream_t stream);
hipError_t launch_pe_concat_strided(const float *feat, int fs, const float *dir, int ds, const float *rays, const unsigned *q_ray, const float *dot, int dts,
                                    long long m_cap, const unsigned *m_dev, float *X, hipStream_t stream);
hipError_t launch_pe_concat_backward(const float *feat, const float *dir, const float *gX, long long m, int with_dot, float *gfeat, float *gdir,
                                     float *gdot, hipStream_t stream);
// more than two encoding frequencies: X as column blocks of TVR_GENX_W columns, block b a contiguous [m_cap, w_b] matrix at X + b * m_cap * TVR_GENX_W (tvr_train.hip)
#define TVR_GENX_W 152
#define TVR_GENX_FLOATS (3 * TVR_GENX_W)          // per entry, at most: 390 columns = 152 + 152 + 88
hipError_t launch_pe_concat_gen(const float *feat, int fs, const float *rays, const unsigned *q_ray, int fea_pe, int view_pe, long long m_cap, const unsigned *m_dev, float *X,
                                hipStream_t stream);
hipError_t launch_copy_cols(float *dst, int ldd, int col0, const float *src, int lds, int ncols, int nrows, hipStream_t stream);
size_t mlp_train_image_bytes();
// heads: REFTensoRF's {normal [3,144], diffuse [3,144], specular [1,144], rho [1,144]} weights, or nullptr (TensorVMSplit)
// fea_pe / view_pe > 2 (TensorVMSplit only): W1 is [128, 30 + 54 fea_pe + 6 view_pe] and is packed into the streamed image behind the LDS image (tvr_mlp_train.hip, TI_W1G)
hipError_t launch_pack_train_image(const float *W1, const float *W2, const float *W3, const float *Bas, const float *const heads[4], void *image, hipStream_t stream,
                                   int fea_pe = 2, int view_pe = 2, const int *app_n_comp = nullptr,      // app_n_comp[3] <= 48 each (nullptr: 48): basis is [27, sum app_n_comp]
                                   int featureC = 128);                                                    // <= 128: W1 [fc, n_in], W2 [fc, fc], W3 [3, fc]
// REFTensoRF's additions to the backward: raw head outputs, view directions, optional gradient of the -dot output; dg8 [m,8] is written
struct MlpRefBwd { const float *g8, *viewdirs, *grad_in0; float *dg8; const float *rays; const unsigned *q_ray; };   // q_ray set: the direction of entry e is rays[q_ray[e]][3..5]
hipError_t launch_mlp_train_backward(const float *grad_rgb, const float *rgb, const float *feats, const float *h1, const float *h2, long long m, const float *gscale,
                                     float *d_out, float *dh2, float *dh1, float *dfeats, float *dh, unsigned *sat_flag, void *image, const MlpRefBwd *ref,
                                     hipStream_t stream, const unsigned *m_dev = nullptr, int gen = 0);
// up to 8 parameter tensors of a regulariser (tvr_reg.hip): x / grad pointers, element counts, rows (line factors: n_comp), launch blocks
#define TVR_REG_MAX 8
struct RegList {
    const float *x[TVR_REG_MAX];
    float *grad[TVR_REG_MAX];
    long long count[TVR_REG_MAX];
    int rows[TVR_REG_MAX];
    int blocks[TVR_REG_MAX];
    int n;
};
size_t reg_l1_scratch_bytes(const RegList &L);
hipError_t launch_l1_forward(const RegList &L, float *value, float *scratch, hipStream_t stream);
hipError_t launch_l1_backward(const RegList &L, const float *g, hipStream_t stream);
hipError_t launch_ortho(const RegList &L, const float *g, float *value, float *scratch, hipStream_t stream);
hipError_t launch_tv_loss(const float *x, int C, int H, int W, float weight, float *value, float *grad, float *part, hipStream_t stream);
hipError_t launch_alpha_bits(const float *vol, long long n, unsigned *bits, hipStream_t stream);

// tvr_step.hip: the training step's compositing tail (forward / backward over the queue, device-side counts), gradient scale, column sums
hipError_t launch_composite_train_forward(const MarchOut &mo, int n_rays, long long cap, int white_bg, const float *rgb, const float *feats32, int with_pen, float *rgb_map,
                                          float *pre, float *pen_ray, hipStream_t stream);
hipError_t launch_composite_train_backward(const MarchOut &mo, int n_rays, long long cap, int white_bg, const float *rgb, const float *feats32, const float *g8, const float *pre,
                                           const float *g_map, const float *g_pen, float *grgb, float *gin0, float *grad_w, float *grad_acc, unsigned *amax_bits,
                                           float target, float *gscale, hipStream_t stream);
// the march queue of a training batch put into RAY ORDER (scan of the per-ray counts, gather into tmp_pos / tmp_ray, copy back; ray_off is updated)
hipError_t launch_queue_ray_order(const MarchOut &mo, unsigned *ray_new, float4 *tmp_pos, unsigned *tmp_ray, int n_rays, hipStream_t stream);
hipError_t launch_linear_dx(const float *dY, int ldy, int N, const float *W, int ldw, int n_valid, int K, const float *mask, int ldm, float *dX, int ldx, long long M,
                            hipStream_t stream, const float *scale = nullptr, unsigned *sat_flag = nullptr, const unsigned long long *mask_bits = nullptr);
hipError_t launch_filter_rays(const SceneDev &sc, const float *rays, long long n, int S, int bbox_only, unsigned char *mask, hipStream_t stream);
size_t colsum_scratch_bytes();
hipError_t launch_colsum(const float *A, int lda, int K, long long m_cap, const unsigned *m_dev, float *out, float *scratch, hipStream_t stream);
hipError_t launch_copy_f32(float *dst, const float *src, int n, hipStream_t stream);
hipError_t launch_zero_header(unsigned *counter, hipStream_t stream);
hipError_t launch_f32_to_f16(const float *in, void *out, long long n, hipStream_t stream);      // n a multiple of 4; round to nearest even
hipError_t launch_zero_f32(float *p, long long n, hipStream_t stream);
