"""GPU (-m gpu): REFTensoRF (models/REFTensoRF.py) — the fused HIP render of the variant configs/Scar.txt trains,
its feature / MLP entry points and its training step, against the oracle's restatement and the committed golden vectors."""
import numpy as np
import pytest
import torch

from conftest import TINY, make_model

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-3      # north_star: RGB L-inf <= 1e-3


def _np(t):
    return t.detach().cpu().numpy()


def test_ref_render_against_golden(tiny_ref, tiny_ref_arrays, hyper_tiny):
    m = make_model(tiny_ref_arrays, hyper_tiny)
    assert type(m).__name__ == "REFTensoRF" and m.renderModule.in_mlpC == 151
    rays = torch.tensor(tiny_ref["rays"], device="cuda")
    rgb_map, depth_map, d = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
    app = _np(d["weight"]) > hyper_tiny["rayMarch_weight_thres"]
    assert np.array_equal(app.astype(np.uint8), tiny_ref["app_mask"])                 # which samples are shaded: exact
    e_s = np.abs(_np(d["rgb"]) - tiny_ref["rgb"]).max()
    e_m = np.abs(_np(rgb_map) - tiny_ref["rgb_map"]).max()
    print(f"REF per-sample rgb Linf {e_s:.2e}, rgb_map Linf {e_m:.2e}")
    assert e_s < RGB_TOL and e_m < RGB_TOL
    assert np.abs(_np(depth_map) - tiny_ref["depth_map"]).max() < 1e-4
    # reference call surface (renderer.py:12-27 -> REFTensoRF.execute), default eps_T
    from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast
    r2, _, d2, _, _ = OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=TINY["N_samples"], white_bg=True, device="cuda")
    assert np.abs(_np(r2) - tiny_ref["rgb_map"]).max() < RGB_TOL
    # masked + jittered + black background variant
    arrs_a = dict(tiny_ref_arrays)
    from jittor_myc_nerfs_amd import synthetic
    al = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=7, alpha_grid=[12, 10, 14])
    arrs_a["alpha_volume"], arrs_a["alpha_aabb"] = al["alpha_volume"], al["alpha_aabb"]
    ma = make_model(arrs_a, hyper_tiny)
    jit = torch.tensor(tiny_ref["wb0_am1_jit.jitter"], device="cuda")
    r3, d3 = ma.render_rays(rays, white_bg=False, N_samples=TINY["N_samples"], jitter=jit, eps_T=0.0)
    assert np.abs(_np(r3) - tiny_ref["wb0_am1_jit.rgb_map"]).max() < RGB_TOL
    assert np.abs(_np(d3) - tiny_ref["wb0_am1_jit.depth_map"]).max() < 1e-4


def test_ref_feature_and_mlp_entry_points(tiny_ref, tiny_ref_arrays, hyper_tiny):
    m = make_model(tiny_ref_arrays, hyper_tiny)
    xyz = torch.tensor(tiny_ref["app_xyz_norm"], device="cuda")
    with torch.no_grad():
        f, rgb_d, tint, normal, rho = m.compute_appfeature(xyz)                         # REFTensoRF.py:107-133
    for got, key, tol in ((f, "app_feature", 2e-5), (rgb_d, "rgb_d", 2e-5), (tint, "specular_tint", 2e-5), (normal, "normal_vector", 2e-5),
                          (rho, "rho", 2e-5)):
        want = tiny_ref[key]
        err = np.abs(_np(got) - want).max() / max(1.0, np.abs(want).max())
        print(f"{key:14s} rel err {err:.2e}")
        assert got.shape == want.shape and err < tol, key
    with torch.no_grad():
        rgb_s = m.renderModule(xyz, torch.tensor(tiny_ref["reflection"], device="cuda"), torch.tensor(tiny_ref["app_feature"], device="cuda"),
                               torch.tensor(-tiny_ref["dot_product"], device="cuda"), None)   # REFTensoRF.py:229
    assert np.abs(_np(rgb_s) - tiny_ref["rgb_s"]).max() < 2e-4
    # the TensorVMSplit-only entry point refuses a REF scene instead of mis-shading it
    from jittor_myc_nerfs_amd._lib import TvrError
    with pytest.raises(TvrError):
        m._mlp_render(torch.zeros(4, 3, device="cuda"), torch.zeros(4, 27, device="cuda"))


def _oracle_with_grads(arrs, hyper):
    from oracle import tensorf_oracle as TO
    sc = TO.scene_from_arrays(arrs, **hyper)
    leaves = {}
    for name in ("density_plane", "density_line", "app_plane", "app_line"):
        for i, t in enumerate(getattr(sc, name)):
            leaves[f"{name}.{i}"] = t.requires_grad_(True)
    leaves["basis_mat"] = sc.basis_mat.requires_grad_(True)
    for k, t in sc.mlp.items():
        leaves[k] = t.requires_grad_(True)
    for k, t in sc.ref.items():
        if not k.startswith("rho"):                     # rho only feeds the unused k argument of MLPRender_Fea_Ref: no gradient
            leaves[k] = t.requires_grad_(True)
    return sc, leaves


@pytest.mark.parametrize("mode", ["static", "eager", "library"])
def test_ref_gradients_match_oracle_autograd(mode, tiny_ref, tiny_ref_arrays, hyper_tiny):
    """train.py:225-257 with model_name = REFTensoRF: loss = sum(rgb_map * c) + 0.5 * penalty (normal_vector_penalty_weight, Scar.txt:7).
    static : the whole step as tvr_train_forward / tvr_train_backward (device-side counts, no host read, fixed-order compositing sums);
    eager  : heads, normalisation, reflection, MLPRender_Fea_Ref and the colour mix as tvr_mlp_train_forward_ref / _backward_ref inside the eager autograd chain;
    library: the same algebra as torch ops over library GEMMs (the round-2 path, kept as a second opinion)."""
    fused = mode != "library"
    from oracle import tensorf_oracle as TO
    rays_np = tiny_ref["rays"]
    S = TINY["N_samples"]
    cw = torch.tensor(np.random.default_rng(12).standard_normal((rays_np.shape[0], 3)).astype(np.float32))
    sc, leaves = _oracle_with_grads(tiny_ref_arrays, hyper_tiny)
    rgb_o, _ = TO.execute(sc, torch.tensor(rays_np), white_bg=True, N_samples=S)
    ((rgb_o * cw).sum() + 0.5 * sc.penalty).backward()
    m = make_model(tiny_ref_arrays, hyper_tiny)
    m.eps_T = 0.0
    m.fused_mlp_training = fused
    m.static_training = mode == "static"
    rgb, depth = m.render_rays_autograd(torch.tensor(rays_np, device="cuda"), white_bg=True, N_samples=S)
    assert np.abs(_np(rgb) - rgb_o.detach().numpy()).max() < 2e-4
    if fused:                                            # the training forward IS the evaluation kernel (per-sample colours bit for bit); the pixels differ
        with torch.no_grad():                            # only by the order in which the compositing sums run (index_add vs the composite kernel)
            rgb_eval, _ = m.render_rays(torch.tensor(rays_np, device="cuda"), white_bg=True, N_samples=S)
        assert float((rgb_eval - rgb.detach()).abs().max()) < 2e-6
    assert abs(float(m.penalty.detach()) - float(sc.penalty.detach())) < 1e-3 * max(1.0, abs(float(sc.penalty.detach())))
    ((rgb * cw.cuda()).sum() + 0.5 * m.penalty).backward()
    mlp = m.renderModule.mlp
    got = {"basis_mat": m.basis_mat.weight.grad, "W1": mlp[0].weight.grad, "b1": mlp[0].bias.grad, "W2": mlp[2].weight.grad,
           "b2": mlp[2].bias.grad, "W3": mlp[4].weight.grad, "b3": mlp[4].bias.grad,
           "normal_W": m.normal_linear.weight.grad, "normal_b": m.normal_linear.bias.grad,
           "diffuse_W": m.diffuse_linear.weight.grad, "diffuse_b": m.diffuse_linear.bias.grad,
           "specular_W": m.specular_linear.weight.grad, "specular_b": m.specular_linear.bias.grad}
    for i in range(3):
        got[f"density_plane.{i}"], got[f"density_line.{i}"] = m.density_plane[i].grad, m.density_line[i].grad
        got[f"app_plane.{i}"], got[f"app_line.{i}"] = m.app_plane[i].grad, m.app_line[i].grad
    for k, ref in leaves.items():
        g, r = got[k].cpu().numpy(), ref.grad.numpy()
        scale = max(np.abs(r).max(), 1e-6)
        err = np.abs(g - r).max() / scale
        print(f"grad {k:18s} rel-max-err {err:.2e}  (max |g| {scale:.2e})")
        assert err < 5e-4, f"{k}: max |grad diff| / max |grad| = {err:.2e}"      # same bound and reasoning as test_gpu_training
        assert np.abs(r).max() > 0, f"{k}: oracle gradient is identically zero"


def test_ref_training_loop_and_checkpoint(tmp_path, tiny_ref, tiny_ref_arrays, hyper_tiny):
    """A short Scar.txt-style optimisation (Adam over get_optparam_groups, MSE + normal penalty) lowers the loss; the checkpoint
    round-trips through save / load (train.py:75-87) to the same pixels."""
    from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast, REFTensoRF
    rays = torch.tensor(tiny_ref["rays"], device="cuda")
    gt = torch.tensor(tiny_ref["rgb_map"], device="cuda").roll(1, dims=1)            # a different target than the model renders
    m = make_model(tiny_ref_arrays, hyper_tiny)
    groups = m.get_optparam_groups(0.02, 0.001)
    assert len(groups) == 10                                                          # REFTensoRF.py:99-106: 6 + 4 heads
    opt = torch.optim.Adam(groups, betas=(0.9, 0.99))
    losses = []
    for it in range(20):
        opt.zero_grad()
        rgb_map, _, _, _, _ = OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=TINY["N_samples"], white_bg=True, is_train=True)
        loss = torch.mean((rgb_map - gt) ** 2)
        total = loss + 0.5 * m.penalty                                                # train.py:253-257
        m.penalty = torch.zeros((), device="cuda")
        total.backward()
        opt.step()
        losses.append(float(loss))
    print("REF training loss", losses[0], "->", losses[-1])
    assert losses[-1] < 0.7 * losses[0]
    with torch.no_grad():
        before, _ = m(rays, is_train=False, white_bg=True, N_samples=TINY["N_samples"])
    path = str(tmp_path / "ref.th")
    m.save(path)
    ckpt = torch.load(path, weights_only=False)
    kwargs = ckpt["kwargs"]; kwargs.update({"device": "cuda"})
    m2 = REFTensoRF(**kwargs)
    m2.load(ckpt)
    with torch.no_grad():
        after, _ = m2(rays, is_train=False, white_bg=True, N_samples=TINY["N_samples"])
    assert torch.equal(before, after)


def test_ref_config1_and_full_size_properties():
    """REFTensoRF at BASELINE configs[0] size (128^3, 64x64 rays, 192 samples) against the oracle on every ray, and at configs[1] size
    (300^3, 800x800 x 512) through size-independent properties plus a random subset against the oracle."""
    from jittor_myc_nerfs_amd import rays as R, synthetic
    from oracle import tensorf_oracle as TO
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"], ref=True)
    hyper = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
    rays = R.frame_rays(R.sphere_poses(8, B["cam_radius"])[0], B["img_wh"][1], B["img_wh"][0], B["camera_angle_x"])
    want, _ = TO.execute(TO.scene_from_arrays(arrs, **hyper), rays, white_bg=True, N_samples=B["N_samples"])
    m = make_model(arrs, hyper)
    got, _ = m.render_rays(rays.cuda(), white_bg=True, N_samples=B["N_samples"])
    e1 = np.abs(_np(got) - want.numpy()).max()
    print(f"REF config1 rgb_map Linf {e1:.2e}")
    assert e1 < 3e-4 < RGB_TOL
    perm = torch.randperm(rays.shape[0], device="cuda")
    gp, _ = m.render_rays(rays.cuda()[perm], white_bg=True, N_samples=B["N_samples"])
    assert torch.equal(gp, got[perm])                                                # batch order / tile position do not matter
    del m

    A = synthetic.SCENE_A
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], ref=True)
    hyper = dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"])
    m = make_model(arrs, hyper)
    rays = R.frame_rays(R.sphere_poses(8, A["cam_radius"])[0], 800, 800, A["camera_angle_x"]).cuda()
    rgb, depth = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"])
    rgb_b, depth_b = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"])
    assert torch.equal(rgb, rgb_b) and torch.equal(depth, depth_b)                  # run-to-run deterministic
    parts = [m.render_rays(rays[c0:c0 + 65536], white_bg=True, N_samples=A["N_samples"]) for c0 in range(0, rays.shape[0], 65536)]
    assert torch.equal(torch.cat([p[0] for p in parts]), rgb)                        # chunk invariance
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    sel = torch.randperm(640000, generator=torch.Generator().manual_seed(2))[:256]
    want, _ = TO.execute(TO.scene_from_arrays(arrs, **hyper), rays[sel.cuda()].cpu(), white_bg=True, N_samples=A["N_samples"])
    e2 = np.abs(_np(rgb[sel.cuda()]) - want.numpy()).max()
    print(f"REF config2 subset rgb_map Linf {e2:.2e}")
    assert e2 < 3e-4 < RGB_TOL


def test_ref_arithmetic_modes(tiny_ref, tiny_ref_arrays, hyper_tiny):
    """`mlp_arith` (include/tvr.h TVR_ARITH_*) on the REFTensoRF render: "f16act" rounds the inputs of layers 1 and 2 (the basis product AND the four heads, whose normal
    feeds the reflection direction and its encoding, keep three products); "f16" takes one product everywhere.  Against the golden picture and per-sample colours;
    what the march decides is bit-identical; at full size each mode against the default mode's frame."""
    m = make_model(tiny_ref_arrays, hyper_tiny)
    m.mlp_arith_tol = 1.0            # the gate (field.py::_settle_arith) forced open: this test MEASURES the modes' raw errors; the gate itself: tests/test_gpu_arith.py
    rays = torch.tensor(tiny_ref["rays"], device="cuda")
    base, errs = None, {}
    for mode in ("f32", "f16act", "f16"):
        m.mlp_arith = mode
        rgb_map, depth_map, d = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
        errs[mode] = (np.abs(_np(d["rgb"]) - tiny_ref["rgb"]).max(), np.abs(_np(rgb_map) - tiny_ref["rgb_map"]).max())
        cur = (depth_map, d["weight"], d["valid"])
        if base is None:
            base = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(base, cur))
    print("REF modes, per-sample rgb / rgb_map L-inf vs oracle: " + ", ".join(f"{k} {v[0]:.2e} / {v[1]:.2e}" for k, v in errs.items()))
    assert errs["f32"][1] < 2e-4 and errs["f16act"][1] < 3e-4 and errs["f16"][1] < 5e-4 < RGB_TOL
    assert errs["f32"][0] < errs["f16act"][0] < errs["f16"][0] < RGB_TOL
    from jittor_myc_nerfs_amd import rays as R, synthetic
    A = synthetic.SCENE_A
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], ref=True)
    big = make_model(arrs, dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"]))
    big.mlp_arith_tol = 1.0
    fr = R.frame_rays(R.sphere_poses(8, A["cam_radius"])[0], 800, 800, A["camera_angle_x"]).cuda()
    pics = {}
    for mode in ("f32", "f16act", "f16"):
        big.mlp_arith = mode
        pics[mode] = big.render_rays(fr, white_bg=True, N_samples=A["N_samples"])
        assert big.arith_in_effect == mode
    e2, e1 = float((pics["f16act"][0] - pics["f32"][0]).abs().max()), float((pics["f16"][0] - pics["f32"][0]).abs().max())
    print(f"REF full size vs the default mode (640 000 rays): f16act {e2:.2e}, f16 {e1:.2e}")
    assert 0 < e2 < 3e-4 and 0 < e1 < 1e-3 and torch.equal(pics["f16"][1], pics["f32"][1])
