"""torch.autograd glue of the training step (tensorf-myc/train.py:225-261): each Function pairs a forward C-ABI call with its backward one.

  * _MarchFn      tvr_march_forward(_z) / tvr_march_backward(_z)   sample_ray .. raw2alpha (tensorBase.py:487-513) w.r.t. the density factors
  * _AppHFn       tvr_app_h_forward / tvr_app_h_backward           the plane*line products of compute_appfeature (tensoRF.py:235-241)
  * _PEConcatFn   tvr_pe_concat (+ backward)                        the MLP input of MLPRender_Fea(_Ref).execute (tensorBase.py:76-82)
  * _LinearFn     library GEMMs forward / dX, tvr_gemm_tn for dW    the Linears of REFTensoRF / NerfPlusPlus (and of shapes the fused kernels are not built for)
  * _MlpTrainFn   tvr_mlp_train_forward / tvr_mlp_train_backward    basis_mat + MLPRender_Fea of TensorVMSplit (tensoRF.py:244, tensorBase.py:76-86): the
                                                                    inference kernel's forward, register-resident MFMA backward, weight gradients by tvr_gemm_tn
"""
from __future__ import annotations

import ctypes as C

import weakref

import torch

from . import _lib as L


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _hip_mm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x [M, N] @ w [N, K] for a tall fp32 batch on the device, through tvr_linear_dx's fp32-INPUT MFMAs (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
    accumulation — a library GEMM's arithmetic without the library).  The kernel takes a reduction of at most 128 (multiples of 8) and 32 / 64 / 96 / 128 output
    columns per call: wider products are cut into blocks, padded with zeros where a block is ragged, and the reduction blocks are added in a fixed order.
    Round 5: the network of a scene with more than two encoding frequencies (390 inputs at TensorBase's default 6 / 6) trains through this instead of rocBLAS."""
    M, N = x.shape
    K = w.shape[1]
    lib = L.lib()
    Np, Kp = (N + 7) // 8 * 8, (K + 31) // 32 * 32
    xp = x if (Np == N and x.is_contiguous()) else torch.nn.functional.pad(x, (0, Np - N)).contiguous()
    wp = w if (Kp == K and w.is_contiguous()) else torch.nn.functional.pad(w, (0, Kp - K)).contiguous()
    out = None
    st = _stream_ptr(x.device)
    for n0 in range(0, Np, 128):
        nn = min(128, Np - n0)
        part = torch.empty((M, Kp), dtype=torch.float32, device=x.device)
        for k0 in range(0, Kp, 128):
            kk = min(128, Kp - k0)
            L.check(lib.tvr_linear_dx(xp.data_ptr() + 4 * n0, Np, nn, wp.data_ptr() + 4 * (n0 * Kp + k0), Kp, min(nn, N - n0), kk, None, 0, None,
                                      part.data_ptr() + 4 * k0, Kp, (part.numel() - k0) * 4, M, None, None, st), "tvr_linear_dx")
        out = part if out is None else out + part
    return out if Kp == K else out[:, :K]


_HIP_MM_MIN_ROWS = 4096     # below this a library GEMM is as good (and the small fixtures of the gradient tests keep exercising it)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b over the M appearance samples of a batch.  Tall batches on the device (M >= 4096, fp32): forward and dX through tvr_linear_dx's fp32-input
    MFMAs (_hip_mm), the weight gradient dW = dY^T X through the tall-skinny reduction tvr_gemm_tn — no library GEMM in the step (round 5; before: forward and
    dX were rocBLAS calls).  Small batches and CPU tensors: torch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= _HIP_MM_MIN_ROWS:
            y = _hip_mm(x, weight.t().contiguous())
            return y + bias if bias is not None else y
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        tall = gy.is_cuda and gy.dtype == torch.float32 and weight.dtype == torch.float32 and gy.shape[0] >= _HIP_MM_MIN_ROWS
        gx = None
        if ctx.needs_input_grad[0]:
            gx = _hip_mm(gy, weight.contiguous()) if tall else gy @ weight
        gw = None
        if ctx.needs_input_grad[1]:
            gw = _gemm_tn(gy, x.contiguous()) if (x.is_cuda and x.shape[0] >= 4096 and x.dtype == torch.float32) else gy.t() @ x
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # torch's column reduction is fine at the foreground's M ~ 3.5e5 (0.1 ms) and collapses at the background's 2.1e6 (2.3 ms)
            gb = _colsum(gy) if (gy.is_cuda and gy.shape[0] >= 1000000 and gy.dtype == torch.float32) else gy.sum(0)
        return gx, gw, gb


_ONES = {}


def _gemm_tn_call(a, lda, Ka, b, ldb, Kb, M, a_off=0, b_off=0):
    out = L.dev_empty((Ka, Kb), torch.float32, a.device, "tvr_gemm_tn C")
    scratch = L.dev_bytes(L.lib().tvr_gemm_tn_scratch_bytes(Ka, Kb, M), a.device, what="tvr_gemm_tn scratch")
    L.check(L.lib().tvr_gemm_tn(a.data_ptr() + 4 * a_off, lda, Ka, b.data_ptr() + 4 * b_off, ldb, Kb, M, out.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                _stream_ptr(a.device)), "tvr_gemm_tn")
    return out


def _gemm_tn_bias_call(a, lda, Ka, b, ldb, Kb, M, a_off=0, b_off=0):
    """(a^T b [Ka,Kb], colsum(a) [Ka]) from ONE pass over a (tvr_gemm_tn_bias: the bias gradient rides along as a virtual ones column)."""
    out = L.dev_empty((Ka, Kb), torch.float32, a.device, "tvr_gemm_tn_bias C")
    cs = L.dev_empty(Ka, torch.float32, a.device, "tvr_gemm_tn_bias colsum")
    scratch = L.dev_bytes(L.lib().tvr_gemm_tn_scratch_bytes(Ka, Kb + 1, M), a.device, what="tvr_gemm_tn_bias scratch")
    L.check(L.lib().tvr_gemm_tn_bias(a.data_ptr() + 4 * a_off, lda, Ka, b.data_ptr() + 4 * b_off, ldb, Kb, M, out.data_ptr(), cs.data_ptr(), scratch.data_ptr(),
                                     scratch.numel(), _stream_ptr(a.device)), "tvr_gemm_tn_bias")
    return out, cs


def _gemm_tn(gy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW [Ka,Kb] = gy^T x for contiguous fp32 gy [M,Ka], x [M,Kb] through tvr_gemm_tn (deterministic).  The kernel takes at most 20
    32x32 output tiles per call: wider products are cut into column blocks of gy and x (pointer offsets, same row strides)."""
    M, Ka, Kb = gy.shape[0], gy.shape[1], x.shape[1]
    ta, tb = (Ka + 31) // 32, (Kb + 31) // 32
    if ta * tb <= 20:
        return _gemm_tn_call(gy, Ka, Ka, x, Kb, Kb, M)
    ca = min(ta, 4) * 32                                   # rows of dW per call
    cb = max(1, 20 // min(ta, 4)) * 32                     # columns of dW per call
    out = torch.empty((Ka, Kb), dtype=torch.float32, device=gy.device)
    for a0 in range(0, Ka, ca):
        for b0 in range(0, Kb, cb):
            out[a0:a0 + ca, b0:b0 + cb] = _gemm_tn_call(gy, Ka, min(ca, Ka - a0), x, Kb, min(cb, Kb - b0), M, a_off=a0, b_off=b0)
    return out


def _colsum(gy: torch.Tensor) -> torch.Tensor:
    """Bias gradient gy.sum(0) for tall gy [M,K] as gy^T 1 through tvr_gemm_tn: torch's column reduction takes ~2.3 ms for
    M = 2.1e6, K = 128; this reads gy once at memory speed and has a fixed summation order."""
    M, K = gy.shape
    key = (gy.device, M)
    ones = _ONES.get(key)
    if ones is None:
        _ONES.clear()
        ones = _ONES[key] = torch.ones(M, dtype=torch.float32, device=gy.device)
    out = torch.empty(K, dtype=torch.float32, device=gy.device)
    for k0 in range(0, K, 640):                            # 20 tiles of 32 rows
        kk = min(640, K - k0)
        out[k0:k0 + kk] = _gemm_tn_call(gy, K, kk, ones, 1, 1, M, a_off=k0).view(-1)
    return out


def _colsum_call(a, lda, K, M, a_off=0):
    """out[k] = sum_m a[m, a_off + k] through tvr_colsum (fixed order)."""
    out = L.dev_empty(K, torch.float32, a.device, "tvr_colsum out")
    scratch = L.dev_bytes(L.lib().tvr_colsum_scratch_bytes(), a.device, what="tvr_colsum scratch")
    L.check(L.lib().tvr_colsum(a.data_ptr() + 4 * a_off, lda, K, M, out.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream_ptr(a.device)), "tvr_colsum")
    return out


def _linear_dx(dY, ldy, N, W, ldw, n_valid, K, mask, ldm, dX, ldx, M, w_off=0, scale=None, sat=None, mask_bits=None):
    """dX[:, :K] = (dY[:, :N] @ W[:n_valid, w_off : w_off + K]) * (mask > 0) through tvr_linear_dx; `scale` (device scalar, a power of two): the fp16-split form;
    `mask_bits` [M,2] int64: the mask as bits (tvr_mlpnet_train_forward's layout) instead of the float matrix."""
    L.check(L.lib().tvr_linear_dx(dY.data_ptr(), ldy, N, W.data_ptr() + 4 * w_off, ldw, n_valid, K, mask.data_ptr() if mask is not None else None, ldm,
                                  mask_bits.data_ptr() if mask_bits is not None else None,
                                  dX.data_ptr(), ldx, dX.numel() * 4, M, scale.data_ptr() if scale is not None else None,
                                  sat.data_ptr() if sat is not None else None, _stream_ptr(dY.device)), "tvr_linear_dx")


def _gemm_tn_scaled_call(a, lda, Ka, b, ldb, Kb, M, scale, bias=True, a_off=0):
    """(a^T b, colsum(a) or None) on the fp16-split MFMAs with a * scale (tvr_gemm_tn_scaled)."""
    out = L.dev_empty((Ka, Kb), torch.float32, a.device, "tvr_gemm_tn_scaled C")
    cs = L.dev_empty(Ka, torch.float32, a.device, "tvr_gemm_tn_scaled colsum") if bias else None
    scratch = L.dev_bytes(L.lib().tvr_gemm_tn_scratch_bytes(Ka, Kb + 1, M), a.device, what="tvr_gemm_tn_scaled scratch")
    L.check(L.lib().tvr_gemm_tn_scaled(a.data_ptr() + 4 * a_off, lda, Ka, b.data_ptr(), ldb, Kb, M, out.data_ptr(), cs.data_ptr() if bias else None, scale.data_ptr(),
                                       scratch.data_ptr(), scratch.numel(), _stream_ptr(a.device)), "tvr_gemm_tn_scaled")
    return out, cs


class _BgNetFn(torch.autograd.Function):
    """NerfPlusPlus's background network under autograd WITHOUT a library GEMM (nerfplusplus.py:66-140 through train.py:258).
    Forward: the fused inference kernel, which also saves every layer's relu output (tvr_mlpnet_train_forward).  Backward: the heads' elementwise part,
    then one tvr_linear_dx per Linear (input gradient with the ReLU mask of the layer in front fused in), weight gradients by tvr_gemm_tn, bias gradients by
    tvr_colsum.  `base_remap_layers` (Linear 128 -> 256, no activation) stays folded into the first rgb layer in both directions: with
    G1 = dH^T base [64,128] and c = colsum(dH),  dW_rgb0[:, :256] = G1 W_remap^T + c b_remap^T,  dW_remap = W_rgb0[:, :256]^T G1,  db_remap = W_rgb0[:, :256]^T c,
    and the gradient that reaches `base` is dH (W_rgb0[:, :256] W_remap) — no [M,256] tensor exists in either pass.
    params = (base W_0, b_0, ..., W_{D-1}, b_{D-1}, sigma W, b, remap W, b, rgb0 W, b, rgbo W, b)."""

    @staticmethod
    def forward(ctx, owner, desc, pts, viewdirs, *params):
        dev = pts.device
        D = desc.D
        n, N = pts.shape[:2]
        M = n * N
        P = [p.detach() for p in params]
        Wr, br, W0, b0 = P[2 * D + 2], P[2 * D + 3], P[2 * D + 4], P[2 * D + 5]
        st = owner._bg_train_state(desc, P)                       # persistent folded weights + the fragment image (stable pointers: repack, no sync)
        W0b = W0[:, :256].contiguous()
        st["W_eff"].copy_(_gemm_tn_call(W0b.t().contiguous(), 64, 64, Wr, 128, 128, 256))          # [64,128] = W0b @ Wr
        st["b_eff"].copy_((W0b * br.unsqueeze(0)).sum(1) + b0)
        st["W_view"].copy_(W0[:, 256:])
        owner._bg_train_pack(desc, st)
        input_ch = 4 + 8 * desc.pos_freqs
        f = lambda *shape: L.dev_empty(shape, torch.float32, dev, "tvr_mlpnet_train_forward saved / output")
        acts = [f(M, 128) for _ in range(D)]
        Hrgb, sig_pre, Epos, Eview = f(M, 64), f(M), f(M, input_ch), f(M, 16)
        rgb, sigma = f(n, N, 3), f(n, N)
        sv = L.MlpnetSaved()
        for l in range(D):
            sv.act[l] = acts[l].data_ptr()
        sv.act_bytes = M * 128 * 4
        sv.rgb_hidden, sv.rgb_hidden_bytes = Hrgb.data_ptr(), M * 64 * 4
        sv.sigma_pre, sv.sigma_pre_bytes = sig_pre.data_ptr(), M * 4
        sv.embed_pos, sv.embed_pos_bytes = Epos.data_ptr(), M * input_ch * 4
        sv.embed_view, sv.embed_view_bytes = Eview.data_ptr(), M * 16 * 4
        masks = [L.dev_empty((M, 2), torch.int64, dev, "tvr_mlpnet_train_forward mask bits") for _ in range(D + 1)]       # relu masks as bits: 16 B per sample and layer for the backward
        for l in range(D):
            sv.act_mask[l] = masks[l].data_ptr()
        sv.rgb_hidden_mask, sv.mask_bytes = masks[D].data_ptr(), M * 16
        p4 = pts.detach().to(torch.float32).contiguous()
        vd = viewdirs.detach().to(torch.float32).contiguous()
        wk = owner._bg_work()
        L.check(L.lib().tvr_mlpnet_train_forward(C.byref(desc), st["image"].data_ptr(), st["image"].numel(), p4.data_ptr(), vd.data_ptr(), M, rgb.data_ptr(),
                                                 sigma.data_ptr(), C.byref(sv), wk.data_ptr(), wk.numel(), _stream_ptr(dev)), "tvr_mlpnet_train_forward")
        ctx.save_for_backward(rgb, sig_pre, Hrgb, Epos, Eview, W0b, *acts, *masks, *P)
        ctx.meta = (D, M, input_ch, owner._bg_layer_inputs(desc), st, owner.bg_grad_scale_target, owner._get_sat_flag())
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        D, M, input_ch, layer_in, st, owner_scale, sat_flag = ctx.meta
        sv = ctx.saved_tensors
        rgb, sig_pre, Hrgb, Epos, Eview, W0b = sv[:6]
        acts, masks, P = sv[6:6 + D], sv[6 + D:7 + 2 * D], sv[7 + 2 * D:]
        Ws, bs = P[2 * D], P[2 * D + 1]
        Wr, br = P[2 * D + 2], P[2 * D + 3]
        Wo = P[2 * D + 6]
        dev = rgb.device
        # heads, elementwise: sigmoid' on rgb, sign of the sigma head (sigma = |pre|)
        dO = torch.zeros((M, 16), dtype=torch.float32, device=dev)                                   # (row lengths in multiples of 16: one fp16 k-step)
        r = rgb.view(M, 3)
        dO[:, :3] = d_rgb.reshape(M, 3) * r * (1.0 - r)
        dHS = torch.zeros((M, 80), dtype=torch.float32, device=dev)
        dHS[:, 64] = d_sigma.reshape(M) * torch.sign(sig_pre)
        # The products below run on the fp16-split MFMAs (3 products, fp32-grade) at ONE power-of-two scale chosen on the device from the incoming gradients:
        # max |dY| * scale ~ owner.bg_grad_scale_target.  Every kernel multiplies its dY by it and divides its result by it again; a result that leaves fp16's
        # range raises the model's saturation flag (field.training_fault_flag -> the optimizer skips the step, check_training_faults lowers the target).
        scale, sat = None, None
        if owner_scale is not None:
            amax = torch.maximum(dO.abs().max(), dHS[:, 64].abs().max()).clamp_min(1e-30)
            scale = torch.exp2(torch.floor(torch.log2(owner_scale / amax))).clamp(2.0 ** -60, 2.0 ** 60).reshape(1).contiguous()
            sat = sat_flag
        _linear_dx(dO, 16, 16, Wo.contiguous(), 64, 3, 64, None, 0, dHS, 80, M, scale=scale, sat=sat, mask_bits=masks[D])    # dH = (dO W_rgbo) * relu'
        Wcat = st["W_cat"]
        Wcat[:64].copy_(st["W_eff"])
        Wcat[64].copy_(Ws.view(128))
        dP = [None] * D
        dP[D - 1] = torch.empty((M, 128), dtype=torch.float32, device=dev)
        _linear_dx(dHS, 80, 80, Wcat, 128, 65, 128, None, 0, dP[D - 1], 128, M, scale=scale, sat=sat, mask_bits=masks[D - 1])   # d pre_{D-1}
        for l in range(D - 1, 0, -1):
            prev, pe = layer_in[l]
            Wl = P[2 * l].contiguous()
            dP[l - 1] = torch.empty((M, 128), dtype=torch.float32, device=dev)
            _linear_dx(dP[l], 128, 128, Wl, Wl.shape[1], 128, 128, None, 0, dP[l - 1], 128, M, w_off=input_ch if pe else 0, scale=scale, sat=sat,
                       mask_bits=masks[l - 1])

        def gtn(a, lda, Ka, b, ldb, Kb, bias=True, a_off=0):           # a^T b (+ colsum a): the scaled fp16-split form when a scale exists
            if scale is not None:
                return _gemm_tn_scaled_call(a, lda, Ka, b, ldb, Kb, M, scale, bias=bias, a_off=a_off)
            if bias:
                return _gemm_tn_bias_call(a, lda, Ka, b, ldb, Kb, M, a_off=a_off)
            return _gemm_tn_call(a, lda, Ka, b, ldb, Kb, M, a_off=a_off), None
        grads = []
        for l in range(D):
            prev, pe = layer_in[l]
            parts, gb = [], None                                    # (the bias gradient rides along in the first product: no extra pass over dP[l])
            if pe:
                g, gb = gtn(dP[l], 128, 128, Epos, input_ch, input_ch)
                parts.append(g)
            if prev:
                g, gb2 = gtn(dP[l], 128, 128, acts[l - 1], 128, 128, bias=gb is None)
                gb = gb if gb is not None else gb2
                parts.append(g)
            grads += [parts[0] if len(parts) == 1 else torch.cat(parts, dim=1), gb]
        base = acts[D - 1]
        # one pass over [dH | d sigma_pre | 0] and base: rows 0..63 = G1 = dH^T base, row 64 = the sigma head's weight gradient; likewise the column sums
        GS, cS = gtn(dHS, 80, 80, base, 128, 128)                                                      # [80,128], [80]
        G1, g_ws = GS[:64].contiguous(), GS[64:65]
        cH, g_bs = cS[:64].contiguous(), cS[64:65]
        Gv = gtn(dHS, 80, 64, Eview, 16, 16, bias=False)[0][:, :15]                                    # [64,15]
        g_w0_base = _gemm_tn_call(G1.t().contiguous(), 64, 64, Wr.t().contiguous(), 256, 256, 128) + cH.unsqueeze(1) * br.unsqueeze(0)      # [64,256]
        g_wr = torch.cat([_gemm_tn_call(W0b, 256, 128, G1, 128, 128, 64, a_off=o) for o in (0, 128)], dim=0)                              # [256,128]
        g_br = (W0b * cH.unsqueeze(1)).sum(0)
        g_wo, g_bo = gtn(dO, 16, 16, Hrgb, 64, 64)
        g_wo, g_bo = g_wo[:3], g_bo[:3]
        grads += [g_ws, g_bs.view_as(bs), g_wr, g_br, torch.cat([g_w0_base, Gv], dim=1), cH, g_wo, g_bo]
        return (None, None, None, None) + tuple(grads)


def _linear(lin: torch.nn.Linear, x):
    return _LinearFn.apply(x, lin.weight, lin.bias)


def _mlp3(mlp: torch.nn.Sequential, x):
    """Linear-ReLU-Linear-ReLU-Linear of MLPRender_Fea / MLPRender_Fea_Ref (tensorBase.py:69-73) through _LinearFn."""
    return _linear(mlp[4], torch.relu(_linear(mlp[2], torch.relu(_linear(mlp[0], x)))))


class _PEConcatFn(torch.autograd.Function):
    """[ (dot,) features, viewdirs, PE(features), PE(viewdirs) ] in one kernel each way (tvr_pe_concat): the torch formulation is four
    elementwise launches plus a concat forward and a dozen backward, each streaming the [M,150] matrix."""

    @staticmethod
    def forward(ctx, features, viewdirs, dot):
        f, v = features.contiguous(), viewdirs.contiguous()
        d = None if dot is None else dot.contiguous().view(-1)
        m = f.shape[0]
        X = torch.empty((m, 150 + (0 if d is None else 1)), dtype=torch.float32, device=f.device)
        L.check(L.lib().tvr_pe_concat(f.data_ptr(), v.data_ptr(), None if d is None else d.data_ptr(), m, X.data_ptr(), L.nbytes(X), _stream_ptr(f.device)),
                "tvr_pe_concat")
        ctx.save_for_backward(f, v)
        ctx.with_dot = d is not None
        ctx.dot_shape = None if dot is None else dot.shape
        return X

    @staticmethod
    def backward(ctx, gX):
        f, v = ctx.saved_tensors
        gX = gX.contiguous()
        m = f.shape[0]
        gf = torch.empty_like(f)
        gv = torch.empty_like(v) if ctx.needs_input_grad[1] else None
        gd = torch.empty(m, dtype=torch.float32, device=f.device) if (ctx.with_dot and ctx.needs_input_grad[2]) else None
        L.check(L.lib().tvr_pe_concat_backward(f.data_ptr(), v.data_ptr(), gX.data_ptr(), m, int(ctx.with_dot), gf.data_ptr(), L.nbytes(gf),
                                               None if gv is None else gv.data_ptr(), None if gd is None else gd.data_ptr(), _stream_ptr(f.device)),
                "tvr_pe_concat_backward")
        return gf, gv, (None if gd is None else gd.view(ctx.dot_shape))


def _mlp_input(features, viewdirs, feape, viewpe, dot=None):
    """The MLP input of MLPRender_Fea (tensorBase.py:76-82) / MLPRender_Fea_Ref (REFTensoRF.py:19-24)."""
    if features.is_cuda and feape == 2 and viewpe == 2 and features.shape[-1] == 27 and features.dtype == torch.float32:
        return _PEConcatFn.apply(features, viewdirs, dot)
    indata = ([] if dot is None else [dot.view(-1, 1)]) + [features, viewdirs]
    if feape > 0:
        indata += [_pe(features, feape)]
    if viewpe > 0:
        indata += [_pe(viewdirs, viewpe)]
    return torch.cat(indata, dim=-1)


def _pe(x, freqs):                                                                            # tensorBase.py:9-15
    fb = 2 ** torch.arange(freqs, device=x.device, dtype=torch.float32)
    pts = (x[..., None] * fb).reshape(x.shape[:-1] + (freqs * x.shape[-1],))
    return torch.cat([torch.sin(pts), torch.cos(pts)], dim=-1)


class _MarchFn(torch.autograd.Function):
    """Training forward/backward of the march (tvr_march_forward / tvr_march_backward, or their explicit-depth _z forms when z_vals is
    given).  Differentiable outputs: the weights of the appearance samples (queue order), acc_map and — z mode only — t_last_tiny =
    prod_j (1 - alpha_j + 1e-6) (NerfPlusPlus's bg_lambda), w.r.t. the six density factors."""

    @staticmethod
    def forward(ctx, model, rays, jitter, S, eps_T, z_vals, *density_params):
        lib = L.lib()
        sc = model._ensure_scene(force=True)          # a training step: the optimizer has just written the parameters (0.1 ms for 70 MB)
        n = rays.shape[0]
        lay = L.ScratchLayout()
        L.check(lib.tvr_scratch_describe(n, S, C.byref(lay)), "tvr_scratch_describe")
        scratch = L.dev_bytes(lay.total, model.device, what="tvr_march_forward scratch")      # owned by this call: backward needs it intact
        depth = torch.empty(n, dtype=torch.float32, device=model.device)
        lam = torch.ones(n, dtype=torch.float32, device=model.device)
        if z_vals is None:
            L.check(lib.tvr_march_forward(sc, rays.data_ptr(), n, S, None if jitter is None else jitter.data_ptr(), float(eps_T),
                                          depth.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream_ptr(model.device)), "tvr_march_forward")
        else:
            L.check(lib.tvr_march_forward_z(sc, rays.data_ptr(), n, S, z_vals.data_ptr(), float(eps_T), depth.data_ptr(), lam.data_ptr(),
                                            scratch.data_ptr(), scratch.numel(), _stream_ptr(model.device)), "tvr_march_forward_z")
        hdr = scratch[lay.counter:lay.counter + 16].view(torch.int32).tolist()           # host sync: the queue length sizes what follows
        if hdr[2] != 0:
            raise L.TvrError(f"tvr_march_forward: the march kernel raised its fault flag ({hdr[2]}): a wave gave up waiting for its tile number "
                             f"(include/tvr.h, tvr_scratch_layout); the queue of this call is incomplete")
        M = int(hdr[0])
        q_pos = scratch[lay.q_pos:lay.q_pos + M * 16].view(torch.float32).view(M, 4)
        w = q_pos[:, 3].clone()
        xyz = q_pos[:, :3].contiguous()
        ray_id = scratch[lay.q_ray:lay.q_ray + M * 4].view(torch.int32).long()
        acc = scratch[lay.acc:lay.acc + n * 4].view(torch.float32).clone()
        ctx.model, ctx.rays, ctx.jitter, ctx.S, ctx.eps_T, ctx.scratch, ctx.M = model, rays, jitter, S, eps_T, scratch, M
        ctx.z_vals, ctx.lam = z_vals, lam.clone()          # (a copy: keeping the returned tensor itself would tie ctx to its own output)
        ctx.shapes = [p.shape for p in density_params]
        ctx.mark_non_differentiable(xyz, ray_id, depth)
        return w, acc, xyz, ray_id, depth, lam

    @staticmethod
    def backward(ctx, gw, gacc, _gx, _gr, _gd, glam):
        model, lib = ctx.model, L.lib()
        sc = model._ensure_scene()
        grads = [torch.empty(sh, dtype=torch.float32, device=model.device) for sh in ctx.shapes]
        out = L.VmGrads()
        for i in range(3):
            out.density_plane[i], out.density_line[i] = grads[i].data_ptr(), grads[3 + i].data_ptr()
        gs = model._get_grad_scratch()
        n = ctx.rays.shape[0]
        # an empty queue (M == 0: every weight <= rayMarch_weight_thres, the state of a freshly initialised 128^3 scene — density only
        # learns through acc_map then, tensorBase.py:515 `if app_mask.any()`) hands autograd a [0] gradient whose data_ptr() is NULL
        gw = torch.zeros(max(ctx.M, 1), device=model.device) if (gw is None or ctx.M == 0) else gw.contiguous().float()
        gacc = torch.zeros(n, device=model.device) if gacc is None else gacc.contiguous().float()
        if ctx.z_vals is None:
            L.check(lib.tvr_march_backward(sc, ctx.rays.data_ptr(), n, ctx.S, None if ctx.jitter is None else ctx.jitter.data_ptr(),
                                           float(ctx.eps_T), ctx.scratch.data_ptr(), ctx.scratch.numel(), gw.data_ptr(), gacc.data_ptr(),
                                           gs.data_ptr(), gs.numel(), C.byref(out), _stream_ptr(model.device)), "tvr_march_backward")
        else:
            glam = torch.zeros(n, device=model.device) if glam is None else glam.contiguous().float()
            L.check(lib.tvr_march_backward_z(sc, ctx.rays.data_ptr(), n, ctx.S, ctx.z_vals.data_ptr(), float(ctx.eps_T), ctx.scratch.data_ptr(),
                                             ctx.scratch.numel(), gw.data_ptr(), gacc.data_ptr(), ctx.lam.data_ptr(), glam.data_ptr(),
                                             gs.data_ptr(), gs.numel(), C.byref(out), _stream_ptr(model.device)), "tvr_march_backward_z")
        model._sig = None       # an optimizer step follows; whatever renders next (training or evaluation) re-packs the scene first
        return (None, None, None, None, None, None, *grads)


class _AppHFn(torch.autograd.Function):
    """h [M,144] = bilinear(app_plane)*linear(app_line) at the queue positions, and the scatter-add backward."""

    @staticmethod
    def forward(ctx, model, xyz, *app_params):
        sc = model._ensure_scene()
        h = torch.empty((xyz.shape[0], 144), dtype=torch.float32, device=model.device)       # the kernels' layout: 3 planes x 48 channels (zero behind a plane's own components)
        L.check(L.lib().tvr_app_h_forward(sc, xyz.data_ptr(), xyz.shape[0], h.data_ptr(), L.nbytes(h), _stream_ptr(model.device)), "tvr_app_h_forward")
        ctx.model, ctx.xyz = model, xyz
        ctx.shapes = [p.shape for p in app_params]
        return h

    @staticmethod
    def backward(ctx, dh):
        model = ctx.model
        sc = model._ensure_scene()
        grads = [torch.empty(sh, dtype=torch.float32, device=model.device) for sh in ctx.shapes]
        out = L.VmGrads()
        for i in range(3):
            out.app_plane[i], out.app_line[i] = grads[i].data_ptr(), grads[3 + i].data_ptr()
        gs = model._get_grad_scratch()
        dh = dh.contiguous().float()
        L.check(L.lib().tvr_app_h_backward(sc, ctx.xyz.data_ptr(), ctx.xyz.shape[0], dh.data_ptr(), L.nbytes(dh), gs.data_ptr(), gs.numel(), C.byref(out),
                                           _stream_ptr(model.device)), "tvr_app_h_backward")
        return (None, None, *grads)


class _MlpTrainFn(torch.autograd.Function):
    """rgb [M,3] = sigmoid(MLP([f, d, PE(f), PE(d)])) with f = basis_mat(h), for TensorVMSplit's MLPRender_Fea (27 features, 2/2 frequencies, width 128).
    Forward = the inference shade kernel fed with h (bit-identical to an evaluation render); backward = tvr_mlp_train_backward (dX chain on
    the matrix cores) + seven tall-skinny reductions (tvr_gemm_tn) for the weight and bias gradients.  No library GEMM."""

    @staticmethod
    def forward(ctx, model, h, viewdirs, basis_w, W1, b1, W2, b2, W3, b3):
        lib = L.lib()
        sc = model._ensure_scene()                     # packed by _MarchFn.forward of this step
        dev = h.device
        m = h.shape[0]
        h = h.contiguous()
        vd = viewdirs.detach().contiguous().float()
        rgb = torch.empty((m, 3), dtype=torch.float32, device=dev)
        feats = torch.empty((m, 32), dtype=torch.float32, device=dev)
        h1 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        h2 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        L.check(lib.tvr_mlp_train_forward(sc, h.data_ptr(), vd.data_ptr(), m, rgb.data_ptr(), L.nbytes(rgb), feats.data_ptr(), L.nbytes(feats),
                                          h1.data_ptr(), L.nbytes(h1), h2.data_ptr(), L.nbytes(h2), _stream_ptr(dev)), "tvr_mlp_train_forward")
        ctx.save_for_backward(h, vd, rgb, feats, h1, h2, basis_w, W1, W2, W3)
        ctx.model = model
        return rgb

    @staticmethod
    def backward(ctx, grgb):
        lib = L.lib()
        h, vd, rgb, feats, h1, h2, basis_w, W1, W2, W3 = ctx.saved_tensors
        dev = h.device
        m = h.shape[0]
        if m == 0:
            z = lambda t: torch.zeros_like(t)
            return (None, torch.zeros_like(h), None, z(basis_w), z(W1), torch.zeros(128, device=dev), z(W2), torch.zeros(128, device=dev), z(W3),
                    torch.zeros(3, device=dev))
        grgb = grgb.contiguous().float()
        # power-of-two scale that brings the largest output gradient to ~model.grad_scale_target (2^6; device side: no host sync).  If the chain
        # still reaches fp16's range (large weights) the kernels set model._sat_flag: model.check_gradient_saturation() reads it and lowers the target
        gmax = (grgb.abs().max() * 0.25).clamp_min(1e-30)
        gscale = torch.exp2(torch.floor(torch.log2(float(ctx.model.grad_scale_target) / gmax))).clamp(2.0 ** -60, 2.0 ** 60).reshape(1).float()
        sat = ctx.model._get_sat_flag()
        d_out = torch.empty((m, 4), dtype=torch.float32, device=dev)
        dh2 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        dh1 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        dfe = torch.empty((m, 32), dtype=torch.float32, device=dev)
        dh = torch.empty((m, 144), dtype=torch.float32, device=dev)
        image = ctx.model._get_train_image()
        W1c, W2c, W3c, Bc = (t.detach().contiguous().float() for t in (W1, W2, W3, basis_w))
        L.check(lib.tvr_mlp_train_backward(W1c.data_ptr(), W2c.data_ptr(), W3c.data_ptr(), Bc.data_ptr(), grgb.data_ptr(), rgb.data_ptr(), feats.data_ptr(),
                                           h1.data_ptr(), h2.data_ptr(), m, gscale.data_ptr(), d_out.data_ptr(), L.nbytes(d_out), dh2.data_ptr(), L.nbytes(dh2),
                                           dh1.data_ptr(), L.nbytes(dh1), dfe.data_ptr(), L.nbytes(dfe), dh.data_ptr(), L.nbytes(dh), sat.data_ptr(),
                                           image.data_ptr(), image.numel(), _stream_ptr(dev)), "tvr_mlp_train_backward")
        X = torch.empty((m, 150), dtype=torch.float32, device=dev)                      # the MLP input, re-derived from the saved features (tensorBase.py:77-82)
        f27 = feats[:, :27].contiguous()
        L.check(lib.tvr_pe_concat(f27.data_ptr(), vd.data_ptr(), None, m, X.data_ptr(), L.nbytes(X), _stream_ptr(dev)), "tvr_pe_concat")
        big = m >= 4096
        tn = (lambda a_, lda, ka, b_, ldb, kb: _gemm_tn_call(a_, lda, ka, b_, ldb, kb, m)) if big else None
        if big:
            # (d_out and dfe are handed over at their full row length — 4 and 32 columns, the surplus rows of the product are dropped: rows that
            #  are contiguous and 16-B aligned take tvr_gemm_tn's 16-B staging loads)
            gW3 = tn(d_out, 4, 4, h2, 128, 128)[:3]
            gW2 = tn(dh2, 128, 128, h1, 128, 128)
            gW1 = tn(dh1, 128, 128, X, 150, 150)
            gB = tn(dfe, 32, 32, h, 144, 144)[:27]
            # bias gradients = column sums: torch's column reduction takes 0.1 ms at M ~ 3.5e5 (a gemm_tn pass over the same matrix 0.25 ms)
            gb3, gb2, gb1 = d_out[:, :3].sum(0), dh2.sum(0), dh1.sum(0)
        else:                                          # tiny batches (tests): the reductions as plain torch products
            gW3, gW2, gW1, gB = d_out[:, :3].t() @ h2, dh2.t() @ h1, dh1.t() @ X, dfe[:, :27].t() @ h
            gb3, gb2, gb1 = d_out[:, :3].sum(0), dh2.sum(0), dh1.sum(0)
        return None, dh, None, gB, gW1, gb1, gW2, gb2, gW3, gb3


class _RefMlpTrainFn(torch.autograd.Function):
    """REFTensoRF's appearance network under autograd (models/REFTensoRF.py:125-133, 217-232): h [M,144] -> basis_mat and the four heads -> normalised
    normal, reflection, -dot -> MLPRender_Fea_Ref -> relu(tint) * rgb_s + rgb_d, as ONE forward kernel (tvr_mlp_train_forward_ref = the inference shade
    kernel fed with h) and the register-resident backward chain (tvr_mlp_train_backward_ref); weight and bias gradients by tvr_gemm_tn / column sums.
    No library GEMM.  Outputs (rgb [M,3], in0 [M] = -dot_product): the caller forms the normal penalty sum w relu(in0)^2 (:236-239) from the second."""

    @staticmethod
    def forward(ctx, model, h, viewdirs, basis_w, nW, nb, dW, db, sW, sb, rW, rb_, W1, b1, W2, b2, W3, b3):
        lib = L.lib()
        sc = model._ensure_scene()
        dev, m = h.device, h.shape[0]
        h = h.contiguous()
        vd = viewdirs.detach().contiguous().float()
        rgb = torch.empty((m, 3), dtype=torch.float32, device=dev)
        rgb_s = torch.empty((m, 3), dtype=torch.float32, device=dev)
        feats = torch.empty((m, 32), dtype=torch.float32, device=dev)
        g8 = torch.empty((m, 8), dtype=torch.float32, device=dev)
        h1 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        h2 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        L.check(lib.tvr_mlp_train_forward_ref(sc, h.data_ptr(), vd.data_ptr(), m, rgb.data_ptr(), L.nbytes(rgb), feats.data_ptr(), L.nbytes(feats), h1.data_ptr(),
                                              L.nbytes(h1), h2.data_ptr(), L.nbytes(h2), g8.data_ptr(), L.nbytes(g8), rgb_s.data_ptr(), L.nbytes(rgb_s),
                                              _stream_ptr(dev)), "tvr_mlp_train_forward_ref")
        ctx.save_for_backward(h, vd, rgb_s, feats, h1, h2, g8, basis_w, nW, dW, sW, rW, W1, W2, W3)
        ctx.model = model
        return rgb, feats[:, 30].clone()

    @staticmethod
    def backward(ctx, grgb, gin0):
        lib = L.lib()
        h, vd, rgb_s, feats, h1, h2, g8, basis_w, nW, dW, sW, rW, W1, W2, W3 = ctx.saved_tensors
        dev, m = h.device, h.shape[0]
        z = torch.zeros_like
        if m == 0:
            return (None, z(h), None, z(basis_w), z(nW), torch.zeros(3, device=dev), z(dW), torch.zeros(3, device=dev), z(sW), torch.zeros(1, device=dev), z(rW),
                    torch.zeros(1, device=dev), z(W1), torch.zeros(128, device=dev), z(W2), torch.zeros(128, device=dev), z(W3), torch.zeros(3, device=dev))
        grgb = torch.zeros((m, 3), device=dev) if grgb is None else grgb.contiguous().float()
        gin0 = None if gin0 is None else gin0.contiguous().float()
        tint = g8[:, 3].clamp_min(0)
        gmax = ((grgb * tint[:, None]).abs().max() * 0.25).clamp_min(1e-30)                   # tint * grad enters the network (the heads' product scales itself)
        gscale = torch.exp2(torch.floor(torch.log2(float(ctx.model.grad_scale_target) / gmax))).clamp(2.0 ** -60, 2.0 ** 60).reshape(1).float()
        sat = ctx.model._get_sat_flag()
        d_out = torch.empty((m, 4), dtype=torch.float32, device=dev)
        dh2 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        dh1 = torch.empty((m, 128), dtype=torch.float32, device=dev)
        dfe = torch.empty((m, 32), dtype=torch.float32, device=dev)
        dg8 = torch.empty((m, 8), dtype=torch.float32, device=dev)
        dh = torch.empty((m, 144), dtype=torch.float32, device=dev)
        image = ctx.model._get_train_image()
        W1c, W2c, W3c, Bc, nWc, dWc, sWc, rWc = (t.detach().contiguous().float() for t in (W1, W2, W3, basis_w, nW, dW, sW, rW))
        heads = (C.c_void_p * 4)(nWc.data_ptr(), dWc.data_ptr(), sWc.data_ptr(), rWc.data_ptr())
        L.check(lib.tvr_mlp_train_backward_ref(W1c.data_ptr(), W2c.data_ptr(), W3c.data_ptr(), Bc.data_ptr(), C.byref(heads), grgb.data_ptr(),
                                               None if gin0 is None else gin0.data_ptr(), rgb_s.data_ptr(), feats.data_ptr(), h1.data_ptr(), h2.data_ptr(), g8.data_ptr(),
                                               vd.data_ptr(), m, gscale.data_ptr(), d_out.data_ptr(), L.nbytes(d_out), dh2.data_ptr(), L.nbytes(dh2), dh1.data_ptr(),
                                               L.nbytes(dh1), dfe.data_ptr(), L.nbytes(dfe), dg8.data_ptr(), L.nbytes(dg8), dh.data_ptr(), L.nbytes(dh), sat.data_ptr(),
                                               image.data_ptr(), image.numel(), _stream_ptr(dev)), "tvr_mlp_train_backward_ref")
        # X [m,151] = [-dot, features, reflection, PE(features), PE(reflection)] re-derived from the saved base values (REFTensoRF.py:19-24)
        X = torch.empty((m, 151), dtype=torch.float32, device=dev)
        f27, refl, in0 = feats[:, :27].contiguous(), feats[:, 27:30].contiguous(), feats[:, 30].contiguous()
        L.check(lib.tvr_pe_concat(f27.data_ptr(), refl.data_ptr(), in0.data_ptr(), m, X.data_ptr(), L.nbytes(X), _stream_ptr(dev)), "tvr_pe_concat")
        if m >= 4096:
            tn = lambda a_, lda, ka, b_, ldb, kb: _gemm_tn_call(a_, lda, ka, b_, ldb, kb, m)
            gW3 = tn(d_out, 4, 4, h2, 128, 128)[:3]
            gW2 = tn(dh2, 128, 128, h1, 128, 128)
            gW1 = tn(dh1, 128, 128, X, 151, 151)
            gB = tn(dfe, 32, 32, h, 144, 144)[:27]
            gH = tn(dg8, 8, 8, h, 144, 144)
        else:
            gW3, gW2, gW1, gB, gH = d_out[:, :3].t() @ h2, dh2.t() @ h1, dh1.t() @ X, dfe[:, :27].t() @ h, dg8.t() @ h
        gb3, gb2, gb1, gbh = d_out[:, :3].sum(0), dh2.sum(0), dh1.sum(0), dg8.sum(0)
        # head rows: normal 0..2, specular 3, diffuse 4..6, rho 7 (rho's gradient is zero: `k = 1 / rho` is unused by MLPRender_Fea_Ref, REFTensoRF.py:18)
        return (None, dh, None, gB, gH[0:3], gbh[0:3], gH[4:7], gbh[4:7], gH[3:4], gbh[3:4], gH[7:8], gbh[7:8], gW1, gb1, gW2, gb2, gW3, gb3)


class _StepToken:
    """Held by the autograd node of a fused training forward; the model's workspace keeps a weak reference (see _FusedStepFn.forward)."""
    __slots__ = ("__weakref__",)


class _FusedStepFn(torch.autograd.Function):
    """TensorBase.execute / REFTensoRF.execute under autograd (train.py:225-261) as TWO C-ABI calls with no host read in between: tvr_train_forward
    (march -> appearance gather -> basis / heads / MLP -> compositing) and tvr_train_backward (its gradient, weight gradients, scatter into the VM
    factors, march backward).  Every kernel behind the march takes the number of appearance samples from the device; the buffers live in the model
    (model._train_buffers: sized once for `app_cap` samples, reused every step), so the step is a fixed sequence of launches — hipGraph-capturable —
    and its picture / loss are bit-reproducible (every ray's queue segment is contiguous and sample-ordered, the compositing sums run in a fixed order).  GRADIENTS:
    for batches of at most 65 536 rays (TVR_RAY_ORDER_MAX_RAYS: the march queue is put into ray order behind the march, using the q_out / q_j regions of the scratch
    as temporaries) the network's gradients — basis, W1..W3, biases, REFTensoRF's heads — are BIT-IDENTICAL run to run (torch.equal in tests/test_gpu_fused_step.py);
    the VM factors' gradients are scattered with fp32 atomics and are reproducible to rounding (<= 2e-6 of the largest entry).  Larger batches keep the order the
    march kernel's waves finished in: every gradient then reproducible to rounding only.  Outputs: rgb_map [n,3], depth [n] (no gradient), pen_ray [n] (REFTensoRF: per-ray normal
    penalty terms; zeros otherwise).  Parameter order: density planes 0..2, density lines 0..2, app planes, app lines, basis, W1, b1, W2, b2, W3, b3
    (+ normal W b, diffuse W b, specular W b, rho W b)."""

    @staticmethod
    def forward(ctx, model, rays, jitter, S, eps_T, white_bg, *params):
        lib = L.lib()
        # Whole-step capture: the eager steps before it must have run on a side stream (training.make_graphed_step does that).  An eager step on the legacy
        # default stream leaves autograd state there; capturing after it ends in a segmentation fault inside hipStreamEndCapture on this ROCm — say so in Python.
        on_default = torch.cuda.current_stream(model.device).cuda_stream == 0
        if torch.cuda.is_current_stream_capturing():
            last = getattr(model, "_last_train_stream_default", None)
            if last is None or last:
                raise RuntimeError("hipGraph capture of a training step " + ("with no eager warm-up step before it" if last is None else "whose last eager step ran on the default stream")
                                   + ": warm up on a side stream first — jittor_myc_nerfs_amd.make_graphed_step(step_fn) does it (the default stream's autograd state "
                                   "crashes hipStreamEndCapture on this ROCm)")
        else:
            model._last_train_stream_default = on_default
        sc = model._ensure_scene(force=True)
        n = rays.shape[0]
        B = model._train_buffers(n, S)
        ref = getattr(model, "_variant", 0) == 1
        rgb_map = torch.empty((n, 3), dtype=torch.float32, device=model.device)
        depth = torch.empty((n,), dtype=torch.float32, device=model.device)
        pen = torch.zeros((n,), dtype=torch.float32, device=model.device)
        L.check(lib.tvr_train_forward(sc, rays.data_ptr(), n, S, None if jitter is None else jitter.data_ptr(), float(eps_T), int(bool(white_bg)),
                                      B["scratch"].data_ptr(), B["scratch"].numel(), B["work"].data_ptr(), B["work"].numel(), B["cap"], rgb_map.data_ptr(),
                                      depth.data_ptr(), pen.data_ptr() if ref else None, _stream_ptr(model.device)), "tvr_train_forward")
        ctx.model, ctx.rays, ctx.jitter, ctx.S, ctx.eps_T, ctx.white_bg, ctx.buf, ctx.ref = model, rays, jitter, S, eps_T, white_bg, B, ref
        # The saved state of this step (the march queue, h / h1 / h2 / features / pre-clamp pixels) lives in the model's ONE workspace: a second fused forward
        # before this one's backward would overwrite it.  Each forward takes a generation number; the backward refuses a workspace that has moved on, and
        # the dispatchers (field.render_rays_autograd, variants.REFTensoRF) send a forward that arrives while another is outstanding down the eager chain,
        # whose Functions own their tensors.  `pending` is a weak reference to a token this graph node holds: a graph that was dropped is not outstanding.
        B["gen"] = B.get("gen", 0) + 1
        ctx.gen, ctx.token = B["gen"], _StepToken()
        B["pending"] = weakref.ref(ctx.token)
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(*params[12:])                 # the network parameters: their CURRENT values are packed by the backward call
        ctx.mark_non_differentiable(depth)
        return rgb_map, depth, pen

    @staticmethod
    def backward(ctx, g_map, _gd, g_pen):
        model, lib, B = ctx.model, L.lib(), ctx.buf
        if B.get("gen") != ctx.gen:
            raise RuntimeError("the fused training step's workspace was overwritten by a later forward before this backward ran (gradient accumulation over two "
                               "batches, or two renders in one loss): its saved activations are gone.  Use model.static_training = False for such loops, or call "
                               "backward() before the next forward — render_rays_autograd() does the former by itself when it can see the outstanding forward")
        B["pending"] = None
        sc = model._ensure_scene()
        net = [t.detach().contiguous().float() for t in ctx.saved_tensors]
        dev, n = model.device, ctx.rays.shape[0]
        grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        vm = L.VmGrads()
        for i in range(3):
            vm.density_plane[i], vm.density_line[i] = grads[i].data_ptr(), grads[3 + i].data_ptr()
            vm.app_plane[i], vm.app_line[i] = grads[6 + i].data_ptr(), grads[9 + i].data_ptr()
        wt, mg = L.TrainWeights(), L.TrainMlpGrads()
        wt.basis, wt.W1, wt.W2, wt.W3 = net[0].data_ptr(), net[1].data_ptr(), net[3].data_ptr(), net[5].data_ptr()
        g = grads[12:]
        mg.basis, mg.W1, mg.b1, mg.W2, mg.b2, mg.W3, mg.b3 = (t.data_ptr() for t in g[:7])
        if ctx.ref:                                         # params 19..26: normal W b, diffuse W b, specular W b, rho W b
            for i in range(4):
                wt.heads_W[i] = net[7 + 2 * i].data_ptr()
                mg.heads_W[i], mg.heads_b[i] = g[7 + 2 * i].data_ptr(), g[8 + 2 * i].data_ptr()
        g_map = torch.zeros((n, 3), device=dev) if g_map is None else g_map.contiguous().float()
        g_pen = None if (g_pen is None or not ctx.ref) else g_pen.contiguous().float()
        gs = model._get_grad_scratch()
        L.check(lib.tvr_train_backward(sc, ctx.rays.data_ptr(), n, ctx.S, None if ctx.jitter is None else ctx.jitter.data_ptr(), float(ctx.eps_T), int(bool(ctx.white_bg)),
                                       B["scratch"].data_ptr(), B["scratch"].numel(), B["work"].data_ptr(), B["work"].numel(), B["cap"], C.byref(wt), g_map.data_ptr(),
                                       None if g_pen is None else g_pen.data_ptr(), float(model.grad_scale_target), gs.data_ptr(), gs.numel(), C.byref(vm), C.byref(mg),
                                       model._get_sat_flag().data_ptr(), _stream_ptr(dev)), "tvr_train_backward")
        model._sig = None
        return (None, None, None, None, None, None, *grads)
