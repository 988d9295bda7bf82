"""GPU: scene shapes other than the one the kernels are built for.  TensorBase's own defaults are 8 / 24 components
(tensorf-myc/models/tensorBase.py:141) and `opt.py` lets featureC / view_pe / fea_pe vary.  Anything that FITS the kernels' shape — up to 16 density
and 48 appearance components per plane, hidden width up to 128, 0..2 encoding frequencies (3..6: the lockstep layer-1 path, inference) — is packed with zero padding (exact: a zero channel
adds 0, a hidden unit that does not exist outputs relu(0) = 0 into zero columns) and rendered by the same kernels; anything larger is refused."""
import numpy as np
import pytest
import torch

from conftest import TINY, make_model  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [([8, 8, 8], [24, 24, 24], 128, 2, 2),          # TensorBase's default component counts
          ([5, 16, 9], [48, 7, 30], 64, 0, 0),           # ragged components, narrow MLP, no positional encoding at all
          ([16, 16, 16], [48, 48, 48], 96, 1, 2),
          ([8, 8, 8], [24, 24, 24], 128, 2, 1),
          ([1, 2, 3], [1, 2, 3], 1, 1, 1),
          # more than two encoding frequencies (round 4): layer 1 runs in lockstep from a streamed 26-k-step image (tvr_device.h TVR_GEN_*)
          ([8, 8, 8], [24, 24, 24], 128, 6, 6),          # TensorBase.__init__'s own defaults, component counts and frequencies (tensorBase.py:141-145): 390 MLP inputs
          ([16, 16, 16], [48, 48, 48], 128, 4, 3),
          ([16, 16, 16], [48, 48, 48], 64, 0, 6),
          # round 6: the kernels' own component counts with the constructor's six frequencies — what the FUSED training step takes at more than two frequencies
          ([16, 16, 16], [48, 48, 48], 128, 6, 6)]


def _scene(dc, ac, fc, vpe, fpe, hyper_tiny):
    from jittor_myc_nerfs_amd import TensorVMSplit, synthetic
    arrs = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=5, density_n_comp=dc, appearance_n_comp=ac, featureC=fc, view_pe=vpe, fea_pe=fpe)
    hyper = dict(hyper_tiny, view_pe=vpe, fea_pe=fpe)
    m = TensorVMSplit(arrs["aabb"], [int(x) for x in arrs["gridSize"]], "cuda", density_n_comp=dc, appearance_n_comp=ac, app_dim=27,
                      near_far=hyper["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper["density_shift"],
                      distance_scale=hyper["distance_scale"], rayMarch_weight_thres=hyper["rayMarch_weight_thres"], pos_pe=6, view_pe=vpe, fea_pe=fpe,
                      featureC=fc, step_ratio=hyper["step_ratio"], fea2denseAct=hyper["fea2denseAct"])
    m.load_arrays(arrs)
    return arrs, hyper, m


@pytest.mark.parametrize("dc,ac,fc,vpe,fpe", SHAPES)
def test_zero_padded_shapes_render_like_the_oracle(tiny_dump, hyper_tiny, dc, ac, fc, vpe, fpe):
    from oracle import tensorf_oracle as TO
    arrs, hyper, m = _scene(dc, ac, fc, vpe, fpe, hyper_tiny)
    sc = TO.scene_from_arrays(arrs, **hyper)
    rays_np = tiny_dump["rays"]
    S = TINY["N_samples"]
    for wb in (True, False):
        rgb_o, depth_o = TO.execute(sc, torch.tensor(rays_np), white_bg=wb, N_samples=S)
        m.eps_T = 0.0
        rgb, depth = m.render_rays(torch.tensor(rays_np, device="cuda"), white_bg=wb, N_samples=S)
        assert np.abs(rgb.cpu().numpy() - rgb_o.numpy()).max() < 2e-4
        assert np.abs(depth.cpu().numpy() - depth_o.numpy()).max() < 2e-3
    # the lookups on their own (tensoRF.py:209-244), at points inside and outside the box
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand((4001, 3), generator=g) * 2 - 1) * 1.05
    sf = m.compute_densityfeature(xyz.cuda()).cpu()
    assert float((sf - TO.compute_densityfeature(sc, xyz)).abs().max()) < 1e-5 * max(1.0, float(sf.abs().max()))
    af = m.compute_appfeature(xyz.cuda()).cpu()
    assert float((af - TO.compute_appfeature(sc, xyz)).abs().max()) < 2e-5
    vd = torch.nn.functional.normalize(torch.randn((4001, 3), generator=g), dim=1)
    feat = torch.randn((4001, 27), generator=g)
    assert float((m._mlp_render(vd.cuda(), feat.cuda()).cpu() - TO.mlp_render_fea(sc, vd, feat)).abs().max()) < 1e-4


@pytest.mark.parametrize("dc,ac,fc,vpe,fpe,hip_mm", [SHAPES[0] + (False,), SHAPES[1] + (False,), SHAPES[5] + (False,), SHAPES[5] + (True,), SHAPES[6] + (True,),
                                                     SHAPES[6] + (None,), SHAPES[8] + (None,), SHAPES[0] + (None,), SHAPES[5] + (None,),
                                                     SHAPES[1] + (None,), SHAPES[2] + (None,), SHAPES[3] + (None,), SHAPES[4] + (None,), SHAPES[7] + (None,)])
def test_zero_padded_shapes_train(tiny_dump, hyper_tiny, monkeypatch, dc, ac, fc, vpe, fpe, hip_mm):
    """Gradients of every parameter tensor at its own (unpadded) shape against autograd through the oracle.  hip_mm: the eager chain's Linears forced onto the HIP
    kernels whatever the batch size (round 5: what a 4096-ray training batch of a six-frequency scene with fewer than 48 components runs — tvr_linear_dx forward and dX,
    tvr_gemm_tn dW; no library GEMM, scripts/pe6_train_trace.sh).  hip_mm None (round 6): EVERY shape of the list — TensorBase's exact defaults 8 / 24 with 6 / 6, ragged components, widths 1 / 64 / 96, 0 .. 6 frequencies — goes through the FUSED step
    (tvr_train_forward / tvr_train_backward: lockstep layer 1, streamed W1^T backward, dW1 in column blocks) — asserted; True / False: the eager chain."""
    from oracle import tensorf_oracle as TO
    from test_gpu_training import _oracle_with_grads
    if hip_mm:
        from jittor_myc_nerfs_amd import autograd_ops
        monkeypatch.setattr(autograd_ops, "_HIP_MM_MIN_ROWS", 1)
    arrs, hyper, m = _scene(dc, ac, fc, vpe, fpe, hyper_tiny)
    if hip_mm is None:
        assert m._fused_step_ok()
    else:
        m.static_training = False
    rays_np = tiny_dump["rays"]
    S = TINY["N_samples"]
    cw = torch.tensor(np.random.default_rng(12).standard_normal((rays_np.shape[0], 3)).astype(np.float32))
    sc, leaves = _oracle_with_grads(arrs, hyper)
    rgb_o, _ = TO.execute(sc, torch.tensor(rays_np), white_bg=True, N_samples=S)
    (rgb_o * cw).sum().backward()
    m.eps_T = 0.0
    rgb, _ = m.render_rays_autograd(torch.tensor(rays_np, device="cuda"), white_bg=True, N_samples=S)
    assert np.abs(rgb.detach().cpu().numpy() - rgb_o.detach().numpy()).max() < 2e-4
    if hip_mm is None:
        assert type(rgb.grad_fn).__name__.startswith("_FusedStepFn"), type(rgb.grad_fn).__name__
    (rgb * cw.cuda()).sum().backward()
    mlp = m.renderModule.mlp
    got = {"basis_mat": m.basis_mat.weight.grad, "W1": mlp[0].weight.grad, "b1": mlp[0].bias.grad, "W2": mlp[2].weight.grad,
           "b2": mlp[2].bias.grad, "W3": mlp[4].weight.grad, "b3": mlp[4].bias.grad}
    for i in range(3):
        got[f"density_plane.{i}"], got[f"density_line.{i}"] = m.density_plane[i].grad, m.density_line[i].grad
        got[f"app_plane.{i}"], got[f"app_line.{i}"] = m.app_plane[i].grad, m.app_line[i].grad
    for k, ref in leaves.items():
        assert tuple(got[k].shape) == tuple(ref.grad.shape), k
        g, r = got[k].cpu().numpy(), ref.grad.numpy()
        assert np.abs(g - r).max() / max(np.abs(r).max(), 1e-6) < 5e-4, k


@pytest.mark.parametrize("kw", [dict(density_n_comp=[17, 16, 16]), dict(appearance_n_comp=[48, 49, 48]), dict(featureC=129), dict(view_pe=7), dict(fea_pe=8)])
def test_shapes_that_do_not_fit_are_refused(hyper_tiny, kw):
    from jittor_myc_nerfs_amd import TensorVMSplit, _lib as L
    args = dict(density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], featureC=128, view_pe=2, fea_pe=2)
    args.update(kw)
    m = TensorVMSplit(TINY["aabb"], TINY["gridSize"], "cuda", app_dim=27, near_far=TINY["near_far"], shadingMode="MLP_Fea", step_ratio=TINY["step_ratio"], **args)
    with pytest.raises(L.TvrError, match="supports"):
        m.render_rays(torch.zeros((4, 6), device="cuda"))


@pytest.mark.parametrize("n_in,n_out,bias", [(144, 27, False), (390, 128, True), (128, 128, True), (128, 3, True), (151, 128, True), (6, 1, True)])
def test_linear_fn_tall_batches_run_on_hip_kernels(n_in, n_out, bias):
    """autograd_ops._LinearFn on a tall batch (the network of a scene with more than two encoding frequencies trains through it: 390 -> 128 -> 128 -> 3 at
    TensorBase's default 6 / 6, and the 144 -> 27 basis): forward and dX through tvr_linear_dx's fp32-input MFMAs (_hip_mm: blocks of 128 in the reduction,
    ragged blocks zero-padded), dW through tvr_gemm_tn — against float64 torch."""
    from jittor_myc_nerfs_amd.autograd_ops import _LinearFn, _HIP_MM_MIN_ROWS
    g = torch.Generator().manual_seed(n_in * 131 + n_out)
    M = _HIP_MM_MIN_ROWS + 905                                                # not a multiple of any tile
    x = torch.randn((M, n_in), generator=g).cuda().requires_grad_(True)
    w = (torch.randn((n_out, n_in), generator=g) / n_in ** 0.5).cuda().requires_grad_(True)
    b = torch.randn((n_out,), generator=g).cuda().requires_grad_(True) if bias else None
    gy = torch.randn((M, n_out), generator=g).cuda()
    y = _LinearFn.apply(x, w, b)
    y.backward(gy)
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    b64 = b.detach().double().requires_grad_(True) if bias else None
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(gy.double())
    def close(a, ref, tol=2e-6):
        return float((a.double() - ref).abs().max()) <= tol * max(float(ref.abs().max()), 1.0)
    assert y.shape == (M, n_out) and close(y, y64)
    assert close(x.grad, x64.grad) and close(w.grad, w64.grad, 2e-5)
    if bias:
        assert close(b.grad, b64.grad, 2e-5)
