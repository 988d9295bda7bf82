"""torch.autograd glue of the training step (tensorf-myc/train.py:225-261): each Function pairs a forward C-ABI call with its backward one.

  * _MarchFn      tvr_march_forward(_z) / tvr_march_backward(_z)   sample_ray .. raw2alpha (tensorBase.py:487-513) w.r.t. the density factors
  * _AppHFn       tvr_app_h_forward / tvr_app_h_backward           the plane*line products of compute_appfeature (tensoRF.py:235-241)
  * _PEConcatFn   tvr_pe_concat (+ backward)                        the MLP input of MLPRender_Fea(_Ref).execute (tensorBase.py:76-82)
  * _LinearFn     library GEMMs forward / dX, tvr_gemm_tn for dW    the Linears of the MLP and basis_mat
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b over the M appearance samples of a batch.  Forward and dX are library GEMMs; the weight gradient dW = dY^T X is the
    tall-skinny reduction tvr_gemm_tn (M ~ 3.5e5 rows, <= 160 columns) that the library runs at ~15 TFLOP/s."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            Ka, Kb, M = weight.shape[0], weight.shape[1], x.shape[0]
            if x.is_cuda and M >= 4096 and ((Ka + 31) // 32) * ((Kb + 31) // 32) <= 20 and x.dtype == torch.float32:
                xc = x.contiguous()
                gw = torch.empty((Ka, Kb), dtype=torch.float32, device=x.device)
                scratch = torch.empty(L.lib().tvr_gemm_tn_scratch_bytes(Ka, Kb, M), dtype=torch.uint8, device=x.device)
                L.check(L.lib().tvr_gemm_tn(gy.data_ptr(), Ka, Ka, xc.data_ptr(), Kb, Kb, M, gw.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                            _stream_ptr(x.device)), "tvr_gemm_tn")
            else:
                gw = gy.t() @ x
        gb = gy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


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
        L.check(L.lib().tvr_pe_concat(f.data_ptr(), v.data_ptr(), None if d is None else d.data_ptr(), m, X.data_ptr(), _stream_ptr(f.device)),
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
        L.check(L.lib().tvr_pe_concat_backward(f.data_ptr(), v.data_ptr(), gX.data_ptr(), m, int(ctx.with_dot), gf.data_ptr(),
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
        scratch = torch.empty(lay.total, dtype=torch.uint8, device=model.device)      # owned by this call: backward needs it intact
        depth = torch.empty(n, dtype=torch.float32, device=model.device)
        lam = torch.ones(n, dtype=torch.float32, device=model.device)
        if z_vals is None:
            L.check(lib.tvr_march_forward(sc, rays.data_ptr(), n, S, None if jitter is None else jitter.data_ptr(), float(eps_T),
                                          depth.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream_ptr(model.device)), "tvr_march_forward")
        else:
            L.check(lib.tvr_march_forward_z(sc, rays.data_ptr(), n, S, z_vals.data_ptr(), float(eps_T), depth.data_ptr(), lam.data_ptr(),
                                            scratch.data_ptr(), scratch.numel(), _stream_ptr(model.device)), "tvr_march_forward_z")
        M = int(scratch[lay.counter:lay.counter + 4].view(torch.int32).item())           # host sync: the queue length sizes what follows
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
        gw = torch.zeros(max(ctx.M, 1), device=model.device) if gw is None else gw.contiguous().float()
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
        h = torch.empty((xyz.shape[0], sum(model.app_n_comp)), dtype=torch.float32, device=model.device)
        L.check(L.lib().tvr_app_h_forward(sc, xyz.data_ptr(), xyz.shape[0], h.data_ptr(), _stream_ptr(model.device)), "tvr_app_h_forward")
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
        L.check(L.lib().tvr_app_h_backward(sc, ctx.xyz.data_ptr(), ctx.xyz.shape[0], dh.data_ptr(), gs.data_ptr(), gs.numel(), C.byref(out),
                                           _stream_ptr(model.device)), "tvr_app_h_backward")
        return (None, None, *grads)
