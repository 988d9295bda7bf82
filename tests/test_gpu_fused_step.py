"""GPU (-m gpu): the training step without a host read (tvr_train_forward / tvr_train_backward; autograd_ops._FusedStepFn) —
  * the step is a fixed launch sequence: captured in a hipGraph (forward, backward AND the fused Adam update); a replay equals the eager step bit for bit in
    the loss and to rounding (<= 2e-6 of the largest entry) in every gradient;
  * two static steps on the same batch render the same picture bit for bit (fixed-order compositing sums; torch's index_add in the eager chain is not
    deterministic) and agree to rounding in the gradients — the order of the RAYS in the queue is run-dependent, see _net_and_vm;
  * a batch whose appearance samples exceed the workspace is FLAGGED (check_training_faults() -> 'overflow', capacity doubled), never silently truncated;
  * static step == eager chain to rounding (the eager chain composites with index_add)."""
import numpy as np
import pytest
import torch

from conftest import TINY, make_model

pytestmark = pytest.mark.gpu


def _batch(tiny_dump, n_rep=8, seed=3):
    rays = np.concatenate([tiny_dump["rays"]] * n_rep).copy()
    rays[:, :3] += 0.01 * np.random.default_rng(seed).standard_normal((rays.shape[0], 3)).astype(np.float32)
    return torch.tensor(rays, device="cuda")


def _params(m):
    return [p for g in m.get_optparam_groups(0.02, 0.001) for p in g["params"]]


def _step_fn(m, rays, target, jitter, opt):
    def step():
        opt.zero_grad(set_to_none=False)
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"], jitter=jitter)
        loss = torch.mean((rgb - target) ** 2)
        if hasattr(m, "penalty"):
            loss = loss + 0.5 * m.penalty
        loss.backward()
        opt.step()
        return loss
    return step


def _net_and_vm(m):
    """(network parameters, VM factors).  What is bit-reproducible run to run is everything PER RAY: the picture, the loss (each ray's queue segment is
    contiguous and sample-ordered, compositing sums run in a fixed order).  The ORDER OF THE RAYS in the queue is whatever order the march kernel's waves
    finish in (one atomicAdd per ray), so sums over all appearance samples — the weight gradients (fixed-order reductions over a run-dependent row order) and
    the VM-factor gradients (fp32 atomic scatter) — are reproducible to rounding (measured <= 2 ulp), not to the bit.  scripts/debug/fused_step_repro.py."""
    vm = list(m.density_plane) + list(m.density_line) + list(m.app_plane) + list(m.app_line)
    ids = {id(p) for p in vm}
    return [p for p in _params(m) if id(p) not in ids], vm


def _close(x, y, rel=2e-5):
    return float((x - y).abs().max()) <= rel * max(float(y.abs().max()), 1e-6) + 1e-9


@pytest.mark.parametrize("ref", [False, True])
def test_training_step_is_graph_capturable_and_replay_equals_eager(ref, tiny_dump, tiny_arrays, tiny_ref_arrays, hyper_tiny):
    """Forward + backward of one batch, eager and as a replayed hipGraph: the loss bit for bit, every gradient to rounding (see _net_and_vm for why not
    to the bit).  Then the WHOLE step — forward, backward and the Adam update — captured once and replayed: the loss falls, no fault flag."""
    arrs = tiny_ref_arrays if ref else tiny_arrays
    rays = _batch(tiny_dump)
    target = torch.rand((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    jitter = torch.rand(rays.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    m = make_model(arrs, hyper_tiny)
    for p in _params(m):
        p.grad = torch.zeros_like(p)

    def fwd_bwd():
        for p in _params(m):
            p.grad.zero_()
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"], jitter=jitter)
        loss = torch.mean((rgb - target) ** 2)
        if ref:
            loss = loss + 0.5 * m.penalty
        loss.backward()
        return loss
    net, vm = _net_and_vm(m)
    # every eager step runs on a SIDE stream (PyTorch's rule for whole-step capture: autograd state created on the default stream — the parameters'
    # AccumulateGrad nodes — would drag the legacy default stream into the capture; on this ROCm that ends in a crash inside hipStreamEndCapture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loss_e = fwd_bwd().detach().clone()
        g_net_e, g_vm_e = [p.grad.clone() for p in net], [p.grad.clone() for p in vm]
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert all(float(g.abs().max()) > 0 for g in g_net_e[:3] + g_vm_e)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_g = fwd_bwd()
    for p in _params(m):
        p.grad.fill_(float("nan"))                       # the replay must produce every gradient itself
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_g.detach(), loss_e), (float(loss_g), float(loss_e))
    for i, (p, ge) in enumerate(zip(net + vm, g_net_e + g_vm_e)):
        assert bool(torch.isfinite(p.grad).all()) and _close(p.grad, ge, 2e-6), \
            f"gradient {i} {tuple(p.shape)} of the replayed graph differs from the eager step beyond rounding: {float((p.grad - ge).abs().max()):.3e} of {float(ge.abs().max()):.3e}"
    assert m.check_training_faults() is None

    # the whole step, optimizer included: captured once; replayed BACK TO BACK (no host sync between replays) and, in a twin model, one replay at a time.
    # Both must train alike: every clear inside the step is a kernel node (memset nodes of a graph replayed back to back ran ahead of the previous
    # replay's kernels on this ROCm: the scratch header held garbage and the step diverged, round 3).
    def captured_step():
        mm = make_model(arrs, hyper_tiny)
        opt = torch.optim.Adam(mm.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), capturable=True, foreach=True)
        step = _step_fn(mm, rays, target, jitter, opt)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            l0 = float(step().detach())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            lg = step()
        return mm, gg, lg, l0, opt
    m2, g2, lg2, l0, opt2 = captured_step()
    for _ in range(8):
        g2.replay()
    torch.cuda.synchronize()
    m3, g3, lg3, _, opt3 = captured_step()
    for _ in range(8):
        g3.replay()
        torch.cuda.synchronize()
    a, b = float(lg2.detach()), float(lg3.detach())
    assert m2.check_training_faults() is None and m3.check_training_faults() is None
    assert a < 0.9 * l0 and abs(a - b) <= 1e-3 * abs(b), (l0, a, b)
    # ... and the PARAMETERS moved alike wherever a gradient is actually there.  Adam's m / sqrt(v) turns the last-bit differences of a gradient that is ~0 into an
    # lr-sized step of either sign, and a `weight > 1e-4` flip of one sample changes single texels by a fraction of a step, so the comparison is of the MOVEMENT
    # vectors over the elements whose first moment is above 1 % of their tensor's largest: per tensor, ||d2 - d3|| <= 5 % of ||d2|| (measured 0.2 - 2 %).
    # A real divergence of the VM factors (the round-3 memset-node fault moved them by whole steps) fails this; equal losses alone would not show it.
    p0s = iter([p.detach().clone() for p in _params(make_model(arrs, hyper_tiny))])      # the same start: make_model is deterministic, the groups come in one order
    checked = 0
    for g2_, g3_ in zip(opt2.param_groups, opt3.param_groups):
        for p2, p3 in zip(g2_["params"], g3_["params"]):
            p0 = next(p0s)
            assert p0.shape == p2.shape
            m1a, m1b = opt2.state[p2]["exp_avg"], opt3.state[p3]["exp_avg"]
            big = (m1a.abs() > 1e-2 * m1a.abs().max()) & (m1b.abs() > 1e-2 * m1b.abs().max())
            if int(big.sum()) < 8:
                continue
            d2, d3 = (p2.detach() - p0)[big], (p3.detach() - p0)[big]
            rel = float((d2 - d3).norm() / d2.norm())
            assert rel <= 0.05, f"parameter {tuple(p2.shape)}: back-to-back and one-at-a-time replays moved {rel:.3f} of their own movement apart"
            checked += int(big.sum())
    assert checked > 1000, checked


def test_static_step_is_reproducible_and_agrees_with_the_eager_chain(tiny_dump, tiny_arrays, hyper_tiny):
    rays = _batch(tiny_dump, 16)
    cw = torch.randn((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))

    def grads(static):
        m = make_model(tiny_arrays, hyper_tiny)
        m.static_training = static
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
        (rgb * cw).sum().backward()
        net, vm = _net_and_vm(m)
        return rgb.detach(), [p.grad.clone() for p in net], [p.grad.clone() for p in vm]
    rgb_a, na, va = grads(True)
    rgb_b, nb, vb = grads(True)
    assert torch.equal(rgb_a, rgb_b), "two static steps on the same batch render different pictures"
    # round 4: the training queue is put into RAY ORDER behind the march (tvr_step.hip, queue_*_kernel), so every reduction over the appearance samples — the
    # weight-gradient products, the bias sums — runs over the same rows in the same order whatever order the march kernel's waves finished in: the NETWORK's
    # gradients are bit-identical run to run.  The VM factors' gradients are scattered with fp32 atomics and stay reproducible to rounding.
    for i, (x, y) in enumerate(zip(na, nb)):
        assert torch.equal(x, y), f"network gradient {i} {tuple(x.shape)} differs between two runs on the same batch: {float((x - y).abs().max()):.3e}"
    assert all(_close(x, y, 2e-6) for x, y in zip(va, vb))
    rgb_e, ne, ve = grads(False)
    assert float((rgb_a - rgb_e).abs().max()) < 2e-6
    assert all(_close(x, y) for x, y in zip(na + va, ne + ve))


@pytest.mark.parametrize("vpe,fpe", [(6, 6), (3, 5)])
def test_static_step_with_more_than_two_frequencies(tiny_dump, hyper_tiny, vpe, fpe):
    """Round 6 (tensorBase.py:141-145: the constructor's view_pe = fea_pe = 6): the fused step of such a scene — lockstep layer 1 forward, dX slot by slot over the streamed
    W1^T image, dW1 in column blocks — against the eager chain of autograd Functions on the same batch; bit-reproducible picture and network gradients run to run; and it
    is the SAME forward as the render path's (torch.equal with render_rays)."""
    from jittor_myc_nerfs_amd import TensorVMSplit, synthetic
    arrs = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=5, view_pe=vpe, fea_pe=fpe)
    rays = _batch(tiny_dump, 16)
    cw = torch.randn((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))

    def model():
        m = TensorVMSplit(arrs["aabb"], [int(x) for x in arrs["gridSize"]], "cuda", density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                          near_far=hyper_tiny["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper_tiny["density_shift"],
                          distance_scale=hyper_tiny["distance_scale"], rayMarch_weight_thres=hyper_tiny["rayMarch_weight_thres"], pos_pe=6, view_pe=vpe, fea_pe=fpe,
                          featureC=128, step_ratio=hyper_tiny["step_ratio"], fea2denseAct=hyper_tiny["fea2denseAct"])
        m.load_arrays(arrs)
        return m

    def grads(static):
        m = model()
        m.static_training = static
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
        assert type(rgb.grad_fn).__name__.startswith("_FusedStepFn") == static
        (rgb * cw).sum().backward()
        net, vm = _net_and_vm(m)
        return rgb.detach(), [p.grad.clone() for p in net], [p.grad.clone() for p in vm], m
    rgb_a, na, va, m = grads(True)
    assert m.renderModule.mlp[0].weight.shape == (128, 30 + 54 * fpe + 6 * vpe)
    assert m.check_training_faults() is None
    rgb_b, nb, vb, _ = grads(True)
    assert torch.equal(rgb_a, rgb_b)
    for i, (x, y) in enumerate(zip(na, nb)):
        assert torch.equal(x, y), f"network gradient {i} {tuple(x.shape)} differs between two runs on the same batch: {float((x - y).abs().max()):.3e}"
    assert all(_close(x, y, 2e-6) for x, y in zip(va, vb))
    rgb_e, ne, ve, _ = grads(False)
    # (the eager chain takes sin / cos of 2^f v from libm, the kernels from the hardware units on the once-reduced argument: 2e-6 at two frequencies, more at 2^5 v)
    assert float((rgb_a - rgb_e).abs().max()) < 2e-5
    # Two correct forwards that differ by 1e-6 put a handful of hidden units (pre-activation within rounding of zero) on different sides of the relu; each such unit
    # moves one sample's whole contribution, 1e-3 .. 1e-2 of a gradient entry at this batch size.  So whole steps agree in the L2 sense here; the arithmetic of every
    # stage is checked to 2e-5 on the step's own tensors (test_every_stage_of_the_fused_backward_against_fp64_on_its_own_tensors).
    for i, (x, y) in enumerate(zip(na + va, ne + ve)):
        l2 = float((x - y).norm() / y.norm().clamp_min(1e-12))
        assert l2 < 3e-2 and _close(x, y, 5e-2), f"gradient {i} {tuple(x.shape)}: fused vs eager L2 {l2:.3e}, max {float((x - y).abs().max()):.3e} of {float(y.abs().max()):.3e}"
    with torch.no_grad():
        rgb_r, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert float((rgb_a - rgb_r).abs().max()) < 2e-6


@pytest.mark.parametrize("vpe,fpe,dc,ac,fc", [(2, 2, 16, 48, 128), (6, 6, 16, 48, 128), (3, 5, 16, 48, 128), (6, 0, 16, 48, 128),
                                              (6, 6, 8, 24, 128),                     # TensorBase.__init__'s own defaults, component counts AND frequencies (tensorBase.py:141-145)
                                              (2, 2, (5, 16, 9), (48, 7, 30), 128),   # ragged component counts
                                              (1, 2, 16, 48, 96), (0, 0, (5, 16, 9), (48, 7, 30), 64), (4, 1, 8, 24, 33)])   # narrower networks, fewer than two frequencies
def test_every_stage_of_the_fused_backward_against_fp64_on_its_own_tensors(tiny_dump, hyper_tiny, vpe, fpe, dc, ac, fc):
    """The step's workspace (tvr_train_work_describe) holds what the forward saved and what every backward stage wrote.  Each stage is recomputed in fp64 FROM THE
    TENSORS THE KERNELS THEMSELVES USED — the relu masks are those of the saved activations, so a hidden unit whose pre-activation sits within rounding of zero cannot
    turn a 1e-6 difference of two correct forwards into a 1e-3 difference of two correct gradients (which is what a comparison of whole steps against the oracle's
    autograd measures at six frequencies, scripts/debug/gen_step_check.py).  tensorBase.py:76-86 (MLPRender_Fea.execute), :9-15 (positional_encoding)."""
    import ctypes as C
    from jittor_myc_nerfs_amd import TensorVMSplit, synthetic, _lib as L
    dc, ac = ([dc] * 3 if isinstance(dc, int) else list(dc)), ([ac] * 3 if isinstance(ac, int) else list(ac))
    arrs = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=5, view_pe=vpe, fea_pe=fpe, density_n_comp=dc, appearance_n_comp=ac, featureC=fc)
    m = TensorVMSplit(arrs["aabb"], [int(x) for x in arrs["gridSize"]], "cuda", density_n_comp=dc, appearance_n_comp=ac, app_dim=27,
                      near_far=hyper_tiny["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper_tiny["density_shift"],
                      distance_scale=hyper_tiny["distance_scale"], rayMarch_weight_thres=hyper_tiny["rayMarch_weight_thres"], pos_pe=6, view_pe=vpe, fea_pe=fpe,
                      featureC=fc, step_ratio=hyper_tiny["step_ratio"], fea2denseAct=hyper_tiny["fea2denseAct"])
    m.load_arrays(arrs)
    rays = _batch(tiny_dump, 16)
    n, S = rays.shape[0], TINY["N_samples"]
    cw = torch.randn((n, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=S)
    assert type(rgb.grad_fn).__name__.startswith("_FusedStepFn")
    (rgb * cw).sum().backward()
    assert m.check_training_faults() is None
    B = m._train_buf
    sl, wl = L.ScratchLayout(), L.TrainWorkLayout()
    L.check(L.lib().tvr_scratch_describe(n, S, C.byref(sl)), "tvr_scratch_describe")
    L.check(L.lib().tvr_train_work_describe(m._ensure_scene(), n, S, B["cap"], C.byref(wl)), "tvr_train_work_describe")
    assert wl.total == L.lib().tvr_train_work_bytes(m._ensure_scene(), n, S, B["cap"]) <= B["work"].numel()
    cnt = int(B["scratch"][sl.counter:sl.counter + 4].view(torch.int32).item())
    assert 1000 < cnt <= B["cap"]
    cap = B["cap"]

    def mat(off, cols, rows=cnt, dtype=torch.float32):
        return B["work"][off:off + cap * cols * 4].view(dtype).view(cap, cols)[:rows]
    f64 = lambda t: t.double()
    h, feats, h1, h2, rgb_s = (f64(mat(o, c)) for o, c in ((wl.h, 144), (wl.feats32, 32), (wl.h1, 128), (wl.h2, 128), (wl.rgb, 3)))
    grgb, d_out, dh2, dh1, dfe, dh = (f64(mat(o, c)) for o, c in ((wl.grgb, 3), (wl.d_out4, 4), (wl.dh2, 128), (wl.dh1, 128), (wl.dfeats32, 32), (wl.dh, 144)))
    mlp = m.renderModule.mlp
    W1r, W2r, W3r, Bas_ref = (f64(t.detach()) for t in (mlp[0].weight, mlp[2].weight, mlp[4].weight, m.basis_mat.weight))
    n_in = 30 + 54 * fpe + 6 * vpe
    assert W1r.shape == (fc, n_in) and W2r.shape == (fc, fc) and W3r.shape == (3, fc)
    # the network at the kernels' width: units that do not exist are zero rows / columns (their activations and gradients must come out as exact zeros)
    W1, W2, W3 = (torch.zeros(sh, dtype=torch.float64, device="cuda") for sh in ((128, n_in), (128, 128), (3, 128)))
    W1[:fc], W2[:fc, :fc], W3[:, :fc] = W1r, W2r, W3r
    if fc < 128:
        assert float(h1[:, fc:].abs().max()) == 0.0 and float(h2[:, fc:].abs().max()) == 0.0 and float(dh1[:, fc:].abs().max()) == 0.0 and float(dh2[:, fc:].abs().max()) == 0.0
    # basis_mat in the kernels' channel order: plane p's components at columns 48 p .. (zero columns behind them: the packed scene's zero channels)
    bcols = torch.cat([torch.arange(48 * p, 48 * p + c) for p, c in enumerate(ac)]).cuda()
    assert Bas_ref.shape == (27, sum(ac))
    Bas = torch.zeros((27, 144), dtype=torch.float64, device="cuda")
    Bas[:, bcols] = Bas_ref
    pad = torch.ones(144, dtype=torch.bool, device="cuda")
    pad[bcols] = False
    assert float(h[:, pad].abs().max()) == 0.0 if bool(pad.any()) else True
    q_ray = B["scratch"][sl.q_ray:sl.q_ray + 4 * cnt].view(torch.int32).long()
    dirs = f64(rays[q_ray, 3:6])

    def close(got, want, what, rel=2e-5):
        err, ref = float((got - want).abs().max()), float(want.abs().max())
        assert err <= rel * max(ref, 1e-12), f"{what}: {err:.3e} of {ref:.3e}"
    close(d_out[:, :3], grgb * rgb_s * (1 - rgb_s), "d_out")
    close(dh2, (h2 > 0) * (d_out[:, :3] @ W3), "dH2")
    close(dh1, (h1 > 0) * (dh2 @ W2), "dH1")
    # the MLP input from the saved features and the entries' view directions (reference column order), and the kernels' own X
    F = feats[:, :27]

    def pe(x, freqs):
        if freqs == 0:
            return x[:, :0]
        pts = (x[:, :, None] * (2.0 ** torch.arange(freqs, device="cuda", dtype=torch.float64))).reshape(x.shape[0], -1)
        return torch.cat([torch.sin(pts), torch.cos(pts)], dim=1)
    X = torch.cat([F, dirs, pe(F, fpe), pe(dirs, vpe)], dim=1)
    assert X.shape[1] == n_in
    if wl.x_block_cols == 150:
        assert wl.x_blocks == 1 and (vpe, fpe, fc) == (2, 2, 128)
        Xk = f64(mat(wl.X, 150))
    else:
        assert wl.x_block_cols == 152 and wl.x_blocks == (n_in + 151) // 152
        parts = []
        for b in range(wl.x_blocks):
            cols = min(152, n_in - 152 * b)
            wb = (cols + 3) & ~3
            blk = B["work"][wl.X + b * cap * 152 * 4:wl.X + b * cap * 152 * 4 + cap * wb * 4].view(torch.float32).view(cap, wb)[:cnt]
            assert float(blk[:, cols:].abs().max()) == 0.0 if wb > cols else True
            parts.append(f64(blk[:, :cols]))
        Xk = torch.cat(parts, dim=1)
    assert float((Xk - X).abs().max()) < (2e-6 if max(vpe, fpe) <= 2 else 2e-5)       # hardware sin / cos of the reduced argument times 2^f against fp64
    # dX and the gradient through the positional encoding
    dX = dh1 @ W1
    dF = dX[:, :27].clone()
    for f in range(fpe):                                                                     # (fea_pe = 0: the features enter layer 1 plainly only)
        si, ci = 30 + torch.arange(27, device="cuda") * fpe + f, 30 + 27 * fpe + torch.arange(27, device="cuda") * fpe + f
        dF += (2.0 ** f) * (torch.cos(F * 2.0 ** f) * dX[:, si] - torch.sin(F * 2.0 ** f) * dX[:, ci])
    close(dfe[:, :27], dF, "dF")
    assert float(dfe[:, 27:].abs().max()) == 0.0
    close(dh, dF @ Bas, "dh")
    # the weight gradients the step returned
    g = lambda p: f64(p.grad)
    close(g(mlp[4].weight), (d_out[:, :3].t() @ h2)[:, :fc], "dW3")
    close(g(mlp[4].bias), d_out[:, :3].sum(0), "db3")
    close(g(mlp[2].weight), (dh2.t() @ h1)[:fc, :fc], "dW2")
    close(g(mlp[2].bias), dh2.sum(0)[:fc], "db2")
    close(g(mlp[0].weight), (dh1.t() @ Xk)[:fc], "dW1")
    close(g(mlp[0].bias), dh1.sum(0)[:fc], "db1")
    close(g(m.basis_mat.weight), (dfe[:, :27].t() @ h)[:, bcols], "dBasis")


@pytest.mark.parametrize("shape", ["2/2", "6/6, 8/24 components, width 96"])
def test_workspace_overflow_is_flagged_not_truncated(tiny_dump, tiny_arrays, hyper_tiny, shape):
    rays = _batch(tiny_dump, 64)                                       # 4096 rays
    if shape == "2/2":
        m = make_model(tiny_arrays, hyper_tiny)
    else:                                                              # round 6: the column-block form of dW1, the streamed W1^T backward, cropped gradients — the same void-step contract
        from jittor_myc_nerfs_amd import TensorVMSplit, synthetic
        arrs6 = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=5, view_pe=6, fea_pe=6, density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], featureC=96)
        m = TensorVMSplit(arrs6["aabb"], [int(x) for x in arrs6["gridSize"]], "cuda", density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27,
                          near_far=hyper_tiny["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper_tiny["density_shift"],
                          distance_scale=hyper_tiny["distance_scale"], rayMarch_weight_thres=hyper_tiny["rayMarch_weight_thres"], pos_pe=6, view_pe=6, fea_pe=6,
                          featureC=96, step_ratio=hyper_tiny["step_ratio"], fea2denseAct=hyper_tiny["fea2denseAct"])
        m.load_arrays(arrs6)
        assert m._fused_step_ok()
    m.train_app_samples_per_ray = 1                                    # capacity max(4096, 1 x 4096) = 4096 entries; the batch shades more
    rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    rgb.sum().backward()
    lay_cap = m._train_buf["cap"]
    # the void step is loud ON THE DEVICE, for a loop that never polls the flags (INTEGRATION.md's plain `loss.backward(); optimizer.step()`): every pixel of the
    # batch is NaN — so is any loss made of them — and every gradient it hands back is an exact zero, not a truncated queue's garbage
    assert bool(torch.isnan(rgb).all())
    assert all(p.grad is not None and bool((p.grad == 0).all()) for p in _params(m))
    assert m.check_training_faults() == "overflow" and m.train_app_samples_per_ray == 2 and m._train_buf is None
    m.train_app_samples_per_ray = 192
    rgb2, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    rgb2.sum().backward()
    assert m.check_training_faults() is None and m._train_buf["cap"] > lay_cap
    with torch.no_grad():
        ref, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert float((rgb2.detach() - ref).abs().max()) < 2e-6


def test_saturation_flag_rises_and_the_lowered_scale_gives_right_gradients(tiny_dump, tiny_arrays, hyper_tiny):
    """The fused backward scales gradients by a power of two (max |grad_rgb| -> grad_scale_target, default 64) on their way through fp16 operands, whose
    conversion saturates silently at 65 504; |dH1| <= 128 |W2| 3 |W3| target, so large weights — or, here, a target set far too high — run the chain into
    that limit.  The kernels must SAY so (device flag -> check_training_faults() == 'saturated', target lowered by 2^4 per call), and once the flag stays
    clear the step must agree with the library-GEMM path."""
    rays = _batch(tiny_dump, 8)
    cw = torch.randn((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))

    def grads(m):
        for p in _params(m):
            p.grad = None
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
        (rgb * cw).sum().backward()
        return [p.grad.clone() for p in _params(m)]
    m = make_model(tiny_arrays, hyper_tiny)
    m.grad_scale_target = 2.0 ** 24
    grads(m)
    assert m.check_training_faults() == "saturated" and m.grad_scale_target == 2.0 ** 20
    for _ in range(8):
        g = grads(m)
        if m.check_training_faults() is None:
            break
    else:
        raise AssertionError("the saturation flag did not clear")
    assert 1.0 <= m.grad_scale_target < 2.0 ** 20
    lib = make_model(tiny_arrays, hyper_tiny)
    lib.static_training = lib.fused_mlp_training = False
    for x, y in zip(g, grads(lib)):
        assert bool(torch.isfinite(x).all()) and _close(x, y, 2e-4)


def test_fault_flag_guards_the_fused_optimizer_on_the_device(tiny_dump, tiny_arrays, hyper_tiny):
    """The loop without a host read per step (reconstruct.py): `optimizer.found_inf = model.training_fault_flag()` — torch's fused Adam kernel skips the update
    of a step that raised a fault flag, by itself; a healthy step is applied; and check_training_faults() later reports, from the accumulator, what was
    skipped in between (and adjusts the capacity) — also after a healthy step has overwritten the scratch header."""
    rays = _batch(tiny_dump, 64)
    m = make_model(tiny_arrays, hyper_tiny)
    opt = torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), fused=True)
    net, vm = _net_and_vm(m)

    def one_step():
        opt.zero_grad()
        rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
        ((rgb - 0.3) ** 2).mean().backward()
        opt.found_inf = m.training_fault_flag()
        flag = opt.found_inf.clone()
        before = [p.detach().clone() for p in net + vm]
        opt.step()
        return float(flag), [float((p.detach() - q).abs().max()) for p, q in zip(net + vm, before)]

    flag, moved = one_step()
    assert flag == 0.0 and max(moved) > 0.0                            # a healthy step is applied
    m.train_app_samples_per_ray = 1                                    # the next step overflows its workspace
    m._train_buf = None
    flag, moved = one_step()
    assert flag == 1.0 and max(moved) == 0.0, "the fused optimizer applied a step whose gradients are void"
    steps = {float(st["step"]) for st in opt.state.values()}
    assert steps == {1.0}                                              # ... and did not count it
    m.train_app_samples_per_ray = 192
    m._train_buf = None
    flag, moved = one_step()                                           # healthy again (its march has zeroed the header words)
    assert flag == 0.0 and max(moved) > 0.0
    assert m.check_training_faults() == "overflow"                     # the accumulator remembers the skipped step
    assert m.check_training_faults() is None


def test_a_second_forward_before_the_first_backward_keeps_both_gradients(tiny_dump, tiny_arrays, hyper_tiny):
    """The fused step keeps its saved state in ONE workspace per model.  Two renders in one loss / gradient accumulation over two batches: the second
    forward sees the first one outstanding and goes down the eager chain, so both backwards are right; a fused forward forced over an outstanding one is refused
    in backward instead of differentiating the wrong activations; a dropped graph does not block the workspace."""
    from jittor_myc_nerfs_amd.autograd_ops import _FusedStepFn
    ra, rb = _batch(tiny_dump, 8), _batch(tiny_dump, 8).flip(0).contiguous()
    S = TINY["N_samples"]

    def one(m, r):
        for p in _params(m):
            p.grad = None
        rgb, _ = m.render_rays_autograd(r, white_bg=True, N_samples=S)
        rgb.square().sum().backward()
        return [p.grad.clone() for p in _params(m)]
    m = make_model(tiny_arrays, hyper_tiny)
    ga, gb = one(m, ra), one(m, rb)
    for p in _params(m):
        p.grad = None
    rgb_a, _ = m.render_rays_autograd(ra, white_bg=True, N_samples=S)
    assert m._fused_step_outstanding()
    rgb_b, _ = m.render_rays_autograd(rb, white_bg=True, N_samples=S)           # eager chain: the workspace still belongs to rgb_a's graph
    (rgb_a.square().sum() + rgb_b.square().sum()).backward()
    assert not m._fused_step_outstanding()
    for p, x, y in zip(_params(m), ga, gb):
        assert _close(p.grad, x + y), f"accumulated gradient of {tuple(p.shape)} differs from the sum of the two single-batch gradients"
    # a dropped graph frees the workspace
    tmp, _ = m.render_rays_autograd(ra, white_bg=True, N_samples=S)
    assert m._fused_step_outstanding()
    del tmp
    assert not m._fused_step_outstanding()
    # forcing the fused Function over an outstanding forward: the first backward is refused
    mlp = m.renderModule.mlp
    args = (m, ra, None, S, float(m.rayMarch_weight_thres), True, *m.density_plane, *m.density_line, *m.app_plane, *m.app_line, m.basis_mat.weight,
            mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight, mlp[4].bias)
    first = _FusedStepFn.apply(*args)[0]
    second = _FusedStepFn.apply(*args)[0]
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        first.sum().backward()
    second.sum().backward()


def test_make_graphed_step_and_the_capture_guard(tiny_dump, tiny_arrays, hyper_tiny):
    """training.make_graphed_step captures the whole step after warming up on a side stream; its replays train like eager steps.  And the mistake that
    used to kill the process (round 3: an eager step on the DEFAULT stream, then capture -> segmentation fault inside hipStreamEndCapture) is a Python
    error now, raised by the fused forward before anything is captured."""
    from jittor_myc_nerfs_amd import make_graphed_step
    rays = _batch(tiny_dump, 8)
    target = torch.rand((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    jitter = torch.rand(rays.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))

    def build():
        mm = make_model(tiny_arrays, hyper_tiny)
        opt = torch.optim.Adam(mm.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), capturable=True, foreach=True)
        return mm, opt, _step_fn(mm, rays, target, jitter, opt)
    ma, _, step_a = build()
    replay = make_graphed_step(step_a, warmup=2)
    for _ in range(6):
        la = replay()
    torch.cuda.synchronize()
    mb, _, step_b = build()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2 + 6):                           # the same number of optimizer steps: 2 warm-ups + 6 replays (capturing a step does not run it)
            lb = step_b()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    a, b = float(la.detach()), float(lb.detach())
    assert ma.check_training_faults() is None and mb.check_training_faults() is None
    assert abs(a - b) <= 2e-3 * abs(b), (a, b)
    assert isinstance(replay.graph, torch.cuda.CUDAGraph) and replay.output is la

    mc, _, step_c = build()
    step_c()                                             # eager, on the default stream: the state that crashes a capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="make_graphed_step"):
        with torch.cuda.graph(g):
            step_c()
    torch.cuda.synchronize()
    md, _, step_d = build()
    with pytest.raises(RuntimeError, match="no eager warm-up"):
        with torch.cuda.graph(torch.cuda.CUDAGraph()):
            step_d()
    torch.cuda.synchronize()


def test_graph_replays_do_not_leave_stale_host_caches(tiny_dump, tiny_arrays, hyper_tiny):
    """hipGraph replays of a captured training step re-pack the fp32 images on the device and run no host code.  What the host had cached about the
    parameters must not survive them: (a) the fp16 copies of the appearance factors the "f16" arithmetic gathers — an f16 render BEFORE the replays converted
    them; an f16 render AFTER must see the moved parameters; (b) the fp16-range proof — the in-kernel check stays on for such a model.  Compared against a
    fresh model loaded with the replayed model's parameters."""
    from jittor_myc_nerfs_amd import make_graphed_step
    rays = _batch(tiny_dump, 8)
    target = torch.rand((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    jitter = torch.rand(rays.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    m = make_model(tiny_arrays, hyper_tiny)
    opt = torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), capturable=True, foreach=True)
    replay = make_graphed_step(_step_fn(m, rays, target, jitter, opt), warmup=2)          # captured with mlp_arith == "f32"
    assert m._captured_update
    eval_rays = torch.tensor(tiny_dump["rays"], device="cuda")
    m.mlp_arith = "f16"
    m.mlp_arith_tol = 1e-3                                                               # (this scene's "f16" picture sits at the default tolerance's edge, 2.5e-4: keep the gate open — the copies are the subject)
    rgb0, _ = m.render_rays(eval_rays, white_bg=True, N_samples=TINY["N_samples"])       # converts the fp16 copies from the current images
    rgb0 = rgb0.clone()
    for _ in range(12):
        replay()
    torch.cuda.synchronize()
    rgb1, _ = m.render_rays(eval_rays, white_bg=True, N_samples=TINY["N_samples"])       # (fused Adam bumps no version counter, and a replay packs BEFORE its update: the host re-packs a captured model on every call)
    # no proof was carried over the replays: the forced re-pack voided it and this call proved the range again, on the moved parameters
    assert m._range_proven == bool(m.fp16_range_report()["proven"])
    assert m.arith_in_effect == "f16"                                                    # measured again on the moved parameters
    # round 6: the helper's graphs count their replays — with none in between the next call re-packs nothing, measures nothing, and FrameStream may overlap
    assert not m._captured_raw and not m._replays_pending() and m.scene_settled()
    sig_before = m._sig
    rgb1b, _ = m.render_rays(eval_rays, white_bg=True, N_samples=TINY["N_samples"])
    assert torch.equal(rgb1b, rgb1) and m._sig is sig_before and m.scene_settled()
    rgb1 = rgb1.clone()
    replay()
    torch.cuda.synchronize()
    assert m._replays_pending() and not m.scene_settled()                                # one more replay: the next host-driven call re-packs again
    rgb3, _ = m.render_rays(eval_rays, white_bg=True, N_samples=TINY["N_samples"])
    assert float((rgb3 - rgb1).abs().max()) > 0 and not m._replays_pending()
    rgb1 = rgb3                                                                          # (the fresh model below is loaded with the parameters as they are NOW)
    fresh = make_model(tiny_arrays, hyper_tiny)
    fresh.mlp_arith_tol = 1e-3
    with torch.no_grad():
        for pf, pm in zip(_params(fresh), _params(m)):
            pf.copy_(pm)
    fresh.mlp_arith = "f16"
    rgb2, _ = fresh.render_rays(eval_rays, white_bg=True, N_samples=TINY["N_samples"])
    assert float((rgb1 - rgb0).abs().max()) > 1e-3                                       # twelve Adam steps at lr 0.02 moved the picture
    assert torch.equal(rgb1, rgb2), float((rgb1 - rgb2).abs().max())
