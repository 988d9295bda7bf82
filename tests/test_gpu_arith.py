"""GPU (-m gpu): the opt-in arithmetics of the appearance network's matrix products (include/tvr.h, tvr_scene_set_arith; field.py `mlp_arith`).

The default ("f32": three fp16 products per fp32 product) is what every other parity test holds.  The reduced modes trade matrix work for rounding:
  "f16act"  layers 1 and 2 take their activations rounded to fp16 (weights keep hi + lo; the basis product keeps three products), two products;
  "f16"     plain fp16 operands everywhere, one product.
Both must stay inside north_star's bar (RGB L-inf 1e-3 against the fp32 reference path) with the margin written below, must not move anything
the march decides (positions, masks, weights, depth: bit-identical to the default mode), and must leave every other entry point alone."""
import numpy as np
import pytest
import torch

from conftest import TINY, make_model

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-3                                  # north_star
RGB_TIGHT = 2e-4
BAR = {"f32": 2e-4, "f16act": 3e-4, "f16": 5e-4}  # what the tests hold per mode against the oracle's fp32 pictures (measured: 5e-5 / 7e-5 / 1.2e-4, one weight-threshold flip = 1e-4)
MODES = ("f32", "f16act", "f16")


def _np(t):
    return t.detach().cpu().numpy()


def test_modes_on_the_fixtures_against_the_oracle_pictures(tiny_dump, tiny_arrays, hyper_tiny, config1_golden):
    from jittor_myc_nerfs_amd import synthetic
    B = synthetic.SCENE_B
    arrs1 = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    hyper1 = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
    cases = [("tiny_dump", make_model(tiny_arrays, hyper_tiny), tiny_dump["rays"], TINY["N_samples"], tiny_dump["out.rgb_map"], tiny_dump["out.depth_map"]),
             ("config1", make_model(arrs1, hyper1), config1_golden["rays"], B["N_samples"], config1_golden["rgb_map"], None)]
    for name, m, rays, S, want_rgb, want_depth in cases:
        rays = torch.tensor(rays, device="cuda")
        base = None
        for mode in MODES:
            m.mlp_arith = mode
            rgb, depth, d = m.render_rays(rays, white_bg=True, N_samples=S, eps_T=0.0, dense=True)
            err = np.abs(_np(rgb) - want_rgb).max()
            per_sample = np.abs(_np(d["rgb"]) - tiny_dump["out.rgb"]).max() if name == "tiny_dump" else float("nan")
            print(f"{name} {mode}: rgb_map L-inf vs oracle {err:.2e}, per-sample rgb {per_sample:.2e}")
            assert err < BAR[mode] < RGB_TOL, (name, mode, err)
            assert bool(torch.isfinite(rgb).all())
            cur = (depth, d["z"], d["valid"], d["weight"], d["acc"])
            if base is None:
                base = cur
            else:                                                   # the march does not know the mode: what it decides is the same bits
                for a, b in zip(base, cur):
                    assert torch.equal(a, b)
        assert L_get(m) == 2
        m.mlp_arith = "f32"
        rgb_back, _ = m.render_rays(rays, white_bg=True, N_samples=S, eps_T=0.0)
        assert L_get(m) == 0 and np.abs(_np(rgb_back) - want_rgb).max() < BAR["f32"]


def L_get(m):
    from jittor_myc_nerfs_amd import _lib as L
    return L.lib().tvr_scene_get_arith(m._scene)


def test_modes_at_full_size_against_the_default_mode_and_the_oracle():
    """BASELINE configs[1]: the whole 800x800 frame per mode against the default mode's frame (all 640 000 rays), and 4096 random rays against the scalar oracle."""
    from jittor_myc_nerfs_amd import rays as R, synthetic
    from oracle import c_oracle as CO, tensorf_oracle as TO
    A = synthetic.SCENE_A
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"])
    hyper = dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"])
    m = make_model(arrs, hyper)
    rays = R.frame_rays(R.sphere_poses(8, A["cam_radius"])[0], 800, 800, A["camera_angle_x"]).cuda()
    sel = torch.randperm(640000, generator=torch.Generator().manual_seed(3))[:4096].cuda()
    sc = TO.scene_from_arrays(arrs, **hyper)
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)
    ref = co.render(_np(rays[sel]), A["N_samples"], white_bg=True, nthreads=16)
    pics = {}
    for mode in MODES:
        m.mlp_arith = mode
        rgb, depth = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"])
        rgb2, depth2 = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"])
        assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2)              # deterministic in every mode
        pics[mode] = (rgb, depth)
        e_or = np.abs(_np(rgb[sel]) - ref["rgb_map"]).max()
        e_def = float((rgb - pics["f32"][0]).abs().max())
        print(f"full size {mode}: RGB L-inf vs oracle (4096 rays) {e_or:.2e}, vs the default mode (640 000 rays) {e_def:.2e}, mean {float((rgb - pics['f32'][0]).abs().mean()):.2e}")
        assert e_or < BAR[mode] < RGB_TOL
        assert e_def < BAR[mode]
        assert torch.equal(depth, pics["f32"][1])
    # chunk / batch-order invariance holds in the reduced modes too (a sample's arithmetic does not depend on its tile)
    m.mlp_arith = "f16act"
    perm = torch.randperm(640000, generator=torch.Generator().manual_seed(7)).cuda()
    rgb_p, _ = m.render_rays(rays[perm].contiguous(), white_bg=True, N_samples=A["N_samples"])
    assert torch.equal(rgb_p, pics["f16act"][0][perm])
    # ... and so does the equality of a merged call (tiles handed out by tickets, csrc/tvr_shade.hip TVR_TICKET) with 65536-ray calls (static stride: too few tiles per wave)
    for mode in ("f16act", "f16"):
        m.mlp_arith = mode
        parts = [m.render_rays(rays[c0:c0 + 65536], white_bg=True, N_samples=A["N_samples"])[0] for c0 in range(0, rays.shape[0], 65536)]
        assert m.arith_in_effect == mode
        assert torch.equal(torch.cat(parts), pics[mode][0])


def test_mlp_render_in_every_mode_and_what_the_modes_leave_alone(tiny_dump, tiny_arrays, hyper_tiny):
    """tvr_mlp_render (MLPRender_Fea.execute, tensorBase.py:76-86) follows the mode; tvr_app_feature and the training forward never do."""
    m = make_model(tiny_arrays, hyper_tiny)
    xyz = torch.tensor(tiny_dump["app_xyz_norm"], device="cuda")
    dirs = torch.tensor(tiny_dump["app_dirs"], device="cuda")
    feat = torch.tensor(tiny_dump["app_feature"], device="cuda")
    f0 = m.compute_appfeature(xyz)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    errs = {}
    for mode, bar in (("f32", 1e-5), ("f16act", 3e-4), ("f16", 6e-4)):
        m.mlp_arith = mode
        m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])           # the gate: a reduced mode runs only once a render has measured it on this scene's rays
        assert m.arith_in_effect == mode
        with torch.no_grad():
            rgb = m.renderModule(xyz, dirs, feat)
        e = np.abs(_np(rgb) - tiny_dump["app_rgb"]).max()
        print(f"mlp_render {mode}: per-sample rgb L-inf vs oracle {e:.2e}")
        assert e < bar
        errs[mode] = e
        assert torch.equal(m.compute_appfeature(xyz), f0)                         # xyz -> features: three products whatever the mode
    assert errs["f32"] < errs["f16act"] < errs["f16"]                             # (the modes really are different kernels)
    m.mlp_arith = "f16"
    m.train()
    torch.manual_seed(0)
    out_a = m(rays, is_train=True, white_bg=True, N_samples=TINY["N_samples"])[0].detach().clone()
    m.mlp_arith = "f32"
    torch.manual_seed(0)
    out_b = m(rays, is_train=True, white_bg=True, N_samples=TINY["N_samples"])[0].detach().clone()
    assert torch.equal(out_a, out_b)                                              # the training forward computes fp32-class in every mode


def test_mode_is_validated_and_survives_a_repack(tiny_arrays, hyper_tiny, tiny_dump):
    from jittor_myc_nerfs_amd import _lib as L
    m = make_model(tiny_arrays, hyper_tiny)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    m.mlp_arith = "bf16"
    with pytest.raises(ValueError):
        m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    m.mlp_arith = "f16act"
    a, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert L.lib().tvr_scene_set_arith(m._scene, 7) == -1 and b"TVR_ARITH" in L.lib().tvr_last_error()     # TVR_ERR_INVALID
    assert L.lib().tvr_scene_get_arith(m._scene) == 1
    with torch.no_grad():
        m.app_plane[0].mul_(1.0)                                                  # bumps the version counter: the scene is re-packed, the mode stays
    b, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert torch.equal(a, b) and L.lib().tvr_scene_get_arith(m._scene) == 1
    # more than two encoding frequencies: accepted, and computed with three products (tvr.h)
    from jittor_myc_nerfs_amd import TensorVMSplit
    g = TensorVMSplit(tiny_arrays["aabb"], [int(x) for x in tiny_arrays["gridSize"]], "cuda", density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                      near_far=hyper_tiny["near_far"], shadingMode="MLP_Fea", pos_pe=6, view_pe=4, fea_pe=4, featureC=128, step_ratio=hyper_tiny["step_ratio"])
    r0, _ = g.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    g.mlp_arith = "f16"
    r1, _ = g.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert torch.equal(r0, r1)


@pytest.mark.parametrize("scale", [20.0, 200.0])
def test_feature_magnitude_is_where_the_modes_differ(tiny_dump, tiny_arrays, hyper_tiny, scale):
    """fp16 rounding is RELATIVE: a scene whose features and activations are 20x / 200x larger than the synthetic scene's (|F| up to ~220, hidden activations ~50)
    carries absolute errors that much larger into the sigmoid: at 200x "f16" leaves north_star's 1e-3 bar (1.45e-3) and "f16act" has a sample 1.8e-3 off.  Round 5:
    neither can happen SILENTLY — the gate (tvr_scene_validate_arith; field.py::_settle_arith) measures a requested mode on the call's own rays against the fp32-class
    arithmetic and REFUSES it beyond mlp_arith_tol = 2.5e-4: at 200x both reduced modes are refused (the picture is the "f32" one, a RuntimeWarning says so), at 20x
    "f16act" runs and "f16" is refused.  With the gate forced open (mlp_arith_tol = 1) the raw errors are still what round 4 measured."""
    from oracle import tensorf_oracle as TO
    arrs = dict(tiny_arrays)
    arrs["basis_mat"] = tiny_arrays["basis_mat"] * np.float32(scale)
    sc = TO.scene_from_arrays(arrs, **hyper_tiny)
    d = TO.execute(sc, torch.tensor(tiny_dump["rays"]), white_bg=True, N_samples=TINY["N_samples"], dump=True)
    fmax = float(TO.compute_appfeature(sc, d["xyz_norm"][d["app_mask"]]).abs().max())
    m = make_model(arrs, hyper_tiny)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    # (1) the gate as shipped: what runs, and what the caller is told
    import warnings
    f32_pic, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0)
    f32_pic = f32_pic.clone()
    expect = {20.0: {"f16act": "f16act", "f16": "f32"}, 200.0: {"f16act": "f32", "f16": "f32"}}[scale]
    for mode in ("f16act", "f16"):
        m.mlp_arith = mode
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            rgb, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0)
        refused = expect[mode] == "f32"
        assert m.arith_in_effect == expect[mode], (mode, m.arith_in_effect, m.arith_max_diff)
        assert L_get(m) == {"f32": 0, "f16act": 1, "f16": 2}[expect[mode]]
        assert any("REFUSED" in str(w.message) for w in wlist) == refused
        assert (m.arith_max_diff > m.mlp_arith_tol) == refused
        if refused:
            assert torch.equal(rgb, f32_pic)                                    # the picture the caller gets is the fp32-class one
        else:
            assert float((rgb - f32_pic).abs().max()) <= m.mlp_arith_tol
        assert np.abs(_np(rgb) - d["rgb_map"].numpy()).max() < 4e-4 < RGB_TOL   # whatever ran, the picture is well inside the bar
    # (2) the gate forced open: the modes' raw errors on this scene
    m.mlp_arith_tol = 1.0
    err = {}
    for mode in MODES:
        m.mlp_arith = mode
        rgb, _, dd = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
        assert m.arith_in_effect == mode
        err[mode] = (np.abs(_np(dd["rgb"]) - d["rgb"].numpy()).max(), np.abs(_np(rgb) - d["rgb_map"].numpy()).max())
        assert bool(torch.isfinite(rgb).all())
    print(f"basis x{scale:g}: max |feature| {fmax:.0f}; per-sample rgb / rgb_map L-inf vs oracle: " + ", ".join(f"{k} {v[0]:.2e} / {v[1]:.2e}" for k, v in err.items()))
    assert err["f32"][1] < RGB_TIGHT
    assert err["f16"][0] > err["f16act"][0] > err["f32"][0]                    # every reduced mode's error is RELATIVE to the activations


def test_a_probe_that_shades_nothing_validates_nothing(tiny_dump, tiny_arrays, hyper_tiny):
    """the gate accepted on `max difference <= tol` alone — a probe whose rays miss the box (a corner chunk, a sparse rank share) compares two background
    pictures, d = 0, and opened the gate for every later frame on these parameters.  Now such a probe neither validates nor refuses: the call renders in "f32", nothing is
    cached, and the next batch — rays that do hit the scene — is measured."""
    import ctypes as C
    import warnings
    from jittor_myc_nerfs_amd import _lib as L
    m = make_model(tiny_arrays, hyper_tiny)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    miss = rays.clone()
    miss[:, 3:6] = -miss[:, 3:6]                                                # the cameras look away from the box
    m.mlp_arith = "f16act"
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        bg, _ = m.render_rays(miss, white_bg=True, N_samples=TINY["N_samples"])
    assert bool((bg == 1).all())                                                # nothing but background
    assert m.arith_probe_samples == 0 and m.arith_in_effect == "f32" and m.arith_max_diff is None
    assert L_get(m) == 0 and b"NOT MEASURED" in L.lib().tvr_last_error()
    assert getattr(m, "_arith_refused_sig", None) is None and not wl and not m.scene_settled()       # neither verdict cached: the next call probes again
    rgb, _ = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    assert m.arith_in_effect == "f16act" and L_get(m) == 1 and m.arith_probe_samples >= 2 * rays.shape[0] and 0 < m.arith_max_diff <= m.mlp_arith_tol
    # the C-ABI alone: an all-miss probe leaves the mode out of effect and says how many samples it saw
    sc = m._ensure_scene()
    L.check(L.lib().tvr_scene_touch(sc), "touch")
    n, S = miss.shape[0], TINY["N_samples"]
    scratch = m._get_scratch(L.lib().tvr_render_scratch_bytes(sc, n, S))
    work = torch.empty(8 * n + 64, dtype=torch.float32, device="cuda")
    md, shaded = C.c_float(-1.0), C.c_int64(-1)
    from jittor_myc_nerfs_amd.autograd_ops import _stream_ptr
    L.check(L.lib().tvr_scene_validate_arith(sc, miss.data_ptr(), n, S, 1, 1e-4, 2.5e-4, scratch.data_ptr(), scratch.numel(), work.data_ptr(), work.numel() * 4,
                                             C.byref(md), C.byref(shaded), _stream_ptr(miss.device)), "validate")
    assert md.value == 0.0 and shaded.value == 0 and L_get(m) == 0


def test_fp16_factor_copies_follow_the_parameters(tiny_arrays, hyper_tiny, tiny_dump):
    """The "f16" arithmetic gathers fp16 COPIES of the appearance planes / lines (include/tvr.h).  They are converted by the first render in the mode, and again by every
    tvr_scene_update while the mode is set: an in-place edit of a factor must show in the next picture exactly as in a model built with the edited factor."""
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    S = TINY["N_samples"]
    m = make_model(tiny_arrays, hyper_tiny)
    f32_0, _ = m.render_rays(rays, white_bg=True, N_samples=S)          # packs the scene in the default mode: no fp16 copies yet
    m.mlp_arith = "f16"
    a, _ = m.render_rays(rays, white_bg=True, N_samples=S)              # the mode was set after the update: this render converts (behind the gate's measurement)
    assert m.arith_in_effect == "f16"
    assert 0 < float((a - f32_0).abs().max()) < 5e-4
    with torch.no_grad():
        m.app_plane[1].mul_(1.5)
        m.app_line[2].add_(0.05)
    b, _ = m.render_rays(rays, white_bg=True, N_samples=S)              # re-pack in "f16": tvr_scene_update converts
    arrs2 = dict(tiny_arrays)
    arrs2["app_plane.1"] = tiny_arrays["app_plane.1"] * np.float32(1.5)
    arrs2["app_line.2"] = tiny_arrays["app_line.2"] + np.float32(0.05)
    m2 = make_model(arrs2, hyper_tiny)
    m2.mlp_arith = "f16"
    c, _ = m2.render_rays(rays, white_bg=True, N_samples=S)
    assert torch.equal(b, c) and float((b - a).abs().max()) > 1e-3
    m.mlp_arith = "f32"                                                  # and back: the fp32 images were never touched
    m2.mlp_arith = "f32"
    assert torch.equal(m.render_rays(rays, white_bg=True, N_samples=S)[0], m2.render_rays(rays, white_bg=True, N_samples=S)[0])
    rep = m.fp16_range_report()
    assert rep["proven"] and 0 < rep["texels"] < 10
