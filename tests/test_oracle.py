"""CPU: the two oracle restatements against each other, against the committed golden vectors and against
known-answer cases.  No GPU, no product code under test here except the shared scene generator."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import tensorf_oracle as TO

from conftest import TINY


def _scenes(arrs, hyper):
    sc = TO.scene_from_arrays(arrs, **hyper)
    return sc, CO.COracle(arrs, step=float(sc.stepSize), **hyper)


def test_torch_oracle_reproduces_golden(tiny_dump, tiny_arrays, hyper_tiny):
    sc, _ = _scenes(tiny_arrays, hyper_tiny)
    assert np.float32(sc.stepSize.item()) == tiny_dump["step"]
    d = TO.execute(sc, torch.tensor(tiny_dump["rays"]), white_bg=True, N_samples=TINY["N_samples"], dump=True)
    assert np.array_equal(d["z_vals"].numpy(), tiny_dump["out.z_vals"])
    assert np.array_equal(d["valid"].numpy().astype(np.uint8), tiny_dump["out.valid"])
    assert np.array_equal(d["cell"].numpy(), tiny_dump["out.cell"])
    assert np.allclose(d["rgb_map"].numpy(), tiny_dump["out.rgb_map"], atol=1e-6)
    assert np.allclose(d["weight"].numpy(), tiny_dump["out.weight"], atol=1e-6)


def test_c_oracle_matches_golden_bit_exact_indices(tiny_dump, tiny_arrays, hyper_tiny):
    _, co = _scenes(tiny_arrays, hyper_tiny)
    c = co.render(tiny_dump["rays"], TINY["N_samples"], white_bg=True, dump=True)
    v = tiny_dump["out.valid"].astype(bool)
    assert np.array_equal(c["tmin"], tiny_dump["out.t_min"])
    assert np.array_equal(c["z"], tiny_dump["out.z_vals"])                      # bit-exact sample positions
    assert np.array_equal(c["valid"], tiny_dump["out.valid"])                   # bit-exact masks
    assert np.array_equal(c["cell"][v], tiny_dump["out.cell"][v])               # bit-exact cell indices
    assert np.abs(c["sf"] - tiny_dump["out.sigma_feature"]).max() < 2e-5
    assert np.abs(c["weight"] - tiny_dump["out.weight"]).max() < 1e-6
    assert np.abs(c["rgb_map"] - tiny_dump["out.rgb_map"]).max() < 1e-5         # tolerance: fp32 rgb
    assert np.abs(c["depth_map"] - tiny_dump["out.depth_map"]).max() < 1e-4
    assert (c["app"] != tiny_dump["out.app_mask"]).sum() <= 2                   # threshold flips at 1 ulp only


@pytest.mark.parametrize("name,wb,am,jit", [("wb1_am0", True, False, False), ("wb0_am0", False, False, False),
                                            ("wb1_am1", True, True, False), ("wb0_am1_jit", False, True, True)])
def test_c_oracle_edge_cases(tiny_edge, tiny_arrays, hyper_tiny, name, wb, am, jit):
    arrs = dict(tiny_arrays)
    if am:
        arrs["alpha_volume"], arrs["alpha_aabb"] = tiny_edge["alpha_volume"], tiny_edge["alpha_aabb"]
    _, co = _scenes(arrs, hyper_tiny)
    c = co.render(tiny_edge["rays"], TINY["N_samples"], white_bg=wb, jitter=tiny_edge["jitter"] if jit else None, dump=True)
    g = lambda k: tiny_edge[f"{name}.{k}"]
    assert np.array_equal(c["tmin"], g("t_min"))
    assert np.array_equal(c["z"], g("z_vals"))
    assert np.array_equal(c["bbox_valid"], g("bbox_valid"))
    assert np.array_equal(c["valid"], g("valid"))
    v = g("valid").astype(bool)
    assert np.array_equal(c["cell"][v], g("cell")[v])
    assert np.abs(c["rgb_map"] - g("rgb_map")).max() < 1e-5
    assert np.abs(c["depth_map"] - g("depth_map")).max() < 1e-4
    # the ray that misses the box: everything invalid, rgb = background, depth = d_z (tensorBase.py:531)
    assert not c["valid"][3].any()
    assert np.allclose(c["rgb_map"][3], 1.0 if wb else 0.0)
    assert c["depth_map"][3] == tiny_edge["rays"][3, 5]


def test_c_oracle_config1_against_golden(config1_golden):
    """BASELINE.json configs[0]: 128^3 grid, 64x64 rays, 192 samples (scene regenerated from the seed)."""
    import hashlib
    from jittor_myc_nerfs_amd import rays as R, synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    shas = dict(s.split(":") for s in config1_golden["scene_sha"])
    for k in ("app_plane.0", "app_line.2", "W1", "basis_mat"):       # pure-RNG arrays: bit-stable on every host
        assert hashlib.sha256(np.ascontiguousarray(arrs[k]).tobytes()).hexdigest() == shas[k], f"synthetic scene drifted: {k}"
    hyper = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
    sc = TO.scene_from_arrays(arrs, **hyper)
    assert sc.nSamples == int(config1_golden["nSamples"]) == 440
    rays = R.frame_rays(R.sphere_poses(8, B["cam_radius"])[0], 64, 64, B["camera_angle_x"])
    assert np.array_equal(rays.numpy(), config1_golden["rays"])         # ray generation is bit-reproducible
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)
    c = co.render(config1_golden["rays"], B["N_samples"], white_bg=True, dump=True, nthreads=8)
    assert np.array_equal(np.packbits(c["valid"]), config1_golden["valid_bits"])
    assert (np.unpackbits(np.packbits(c["app"])) != np.unpackbits(config1_golden["app_bits"])).sum() <= 8
    # a 1-ulp weight difference may flip `weight > 1e-4` for a sample: RGB then moves by <= weight ~ 1e-4 (SURVEY §7 hard part 4)
    assert np.abs(c["rgb_map"] - config1_golden["rgb_map"]).max() < 2e-4
    assert np.abs(c["acc"] - config1_golden["acc_map"]).max() < 2e-5


def test_jittor_formulations_stay_inside_the_parity_bar(tiny_dump, tiny_edge, tiny_arrays, hyper_tiny, config1_golden):
    """Parity is UNPINNED (no Jittor, no reference fixtures).  What can be bounded: the two Jittor-internal formulations SURVEY Appendix B could not
    verify — jt.cumprod as exp(cumsum(log)) and jt.nn.softplus without log1p — are modelled by OracleScene(jittor_semantics=True); on every fixture
    the modelled variant stays orders of magnitude inside north_star's 1e-3 RGB bar against the torch formulation the oracle (and the golden vectors) use.
    Sample positions, masks and cell indices do not depend on either formulation."""
    from jittor_myc_nerfs_amd import synthetic
    cases = []
    for nm, rays, S, extra in (("tiny_dump", tiny_dump["rays"], TINY["N_samples"], {}), ("tiny_edge", tiny_edge["rays"], TINY["N_samples"], {}),
                               ("tiny_edge+mask", tiny_edge["rays"], TINY["N_samples"], dict(alpha_volume=tiny_edge["alpha_volume"], alpha_aabb=tiny_edge["alpha_aabb"]))):
        cases.append((nm, dict(tiny_arrays, **extra), hyper_tiny, rays, S))
    B = synthetic.SCENE_B
    cases.append(("config1", synthetic.make_scene_arrays(B["gridSize"], B["aabb"]), dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]),
                  config1_golden["rays"], B["N_samples"]))
    for nm, arrs, hyper, rays, S in cases:
        for wb in (True, False):
            a = TO.execute(TO.scene_from_arrays(arrs, **hyper), torch.tensor(rays), white_bg=wb, N_samples=S, dump=True)
            b = TO.execute(TO.scene_from_arrays(arrs, jittor_semantics=True, **hyper), torch.tensor(rays), white_bg=wb, N_samples=S, dump=True)
            assert torch.equal(a["valid"], b["valid"]) and torch.equal(a["z_vals"], b["z_vals"]) and torch.equal(a["cell"], b["cell"])
            dw = float((a["weight"] - b["weight"]).abs().max())
            dsig = float((a["sigma"] - b["sigma"]).abs().max())
            flips = int((a["app_mask"] != b["app_mask"]).sum())
            drgb = float((a["rgb_map"] - b["rgb_map"]).abs().max())
            ddep = float((a["depth_map"] - b["depth_map"]).abs().max())
            n_app = int(a["app_mask"].sum())
            # measured (this container): the cumprod formulation alone moves weights by <= 6e-8 and flips no threshold; softplus without log1p is 1.3e-3 RELATIVE off at
            # sigma ~ 1e-4 (1 + e^x rounds e^x to 24 bits of 1), which moves weights by <= 4.2e-7 and flips `weight > 1e-4` for 51 of config1's 350 889 shaded samples
            # (none in the tiny fixtures); a flipped sample enters or leaves a pixel with its weight ~1e-4: RGB <= 1.02e-4 on config1, <= 8e-7 where nothing flips
            assert dw < 1e-6, (nm, wb, dw)
            assert dsig < 5e-6, (nm, wb, dsig)
            assert flips <= 2 + 3e-4 * n_app, (nm, wb, flips, n_app)
            assert drgb < (2.5e-4 if flips else 2e-6), (nm, wb, drgb, flips)          # north_star's bar: 1e-3
            assert ddep < 1e-4, (nm, wb, ddep)


# ---- known-answer tests (no data needed) ----
def _const_scene(p, l, g=(6, 7, 8)):
    arrs = {"aabb": np.array([[-1, -1, -1], [1, 1, 1]], np.float32), "gridSize": np.array(g, np.int32)}
    from jittor_myc_nerfs_amd.synthetic import MAT_MODE, VEC_MODE
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        arrs[f"density_plane.{i}"] = np.full((1, 16, g[m1], g[m0]), p, np.float32)
        arrs[f"density_line.{i}"] = np.full((1, 16, g[VEC_MODE[i]], 1), l, np.float32)
        arrs[f"app_plane.{i}"] = np.full((1, 48, g[m1], g[m0]), 0.1, np.float32)
        arrs[f"app_line.{i}"] = np.full((1, 48, g[VEC_MODE[i]], 1), 0.1, np.float32)
    arrs["basis_mat"] = np.zeros((27, 144), np.float32)
    for k, s in (("W1", (128, 150)), ("b1", (128,)), ("W2", (128, 128)), ("b2", (128,)), ("W3", (3, 128)), ("b3", (3,))):
        arrs[k] = np.zeros(s, np.float32)
    return arrs


def test_known_answer_constant_factors():
    arrs = _const_scene(0.5, 0.25)
    hyper = dict(near_far=[0.1, 10.0], step_ratio=0.5)
    sc, co = _scenes(arrs, hyper)
    xyz = np.random.default_rng(0).uniform(-0.999, 0.999, (200, 3)).astype(np.float32)
    expect = 3 * 16 * 0.5 * 0.25                                   # sigma_feature = 3*16*p*l exactly
    assert np.allclose(co.density_features(xyz), expect, rtol=1e-6)
    assert np.allclose(TO.compute_densityfeature(sc, torch.tensor(xyz)).numpy(), expect, rtol=1e-6)
    # outside [-1,1]: zeros padding -> feature decays to 0 one cell outside
    far = np.array([[3.0, 0, 0], [0, -3.0, 0]], np.float32)
    assert np.allclose(co.density_features(far), TO.compute_densityfeature(sc, torch.tensor(far)).numpy(), atol=1e-6)


def test_known_answer_empty_and_opaque():
    rays = np.array([[0, 0, 3, 0, 0, -1], [0.2, 0.1, 3, 0, 0, -1]], np.float32)
    # sigma = relu(0) = 0 everywhere -> rgb = white exactly, depth = d_z (tensorBase.py:531), no appearance sample
    _, co = _scenes(_const_scene(0.0, 0.0), dict(near_far=[0.1, 10.0], step_ratio=0.5, fea2denseAct="relu"))
    c = co.render(rays, 32, white_bg=True, dump=True)
    assert c["acc"].max() == 0 and np.array_equal(c["rgb_map"], np.ones((2, 3), np.float32)) and c["app"].sum() == 0
    assert np.array_equal(c["depth_map"], rays[:, 5])
    # softplus(0 - 10) ~ 4.5e-5 is NOT empty at this step size: alpha ~ 1.9e-4 > thres, rgb = sigmoid(0) = 0.5 -> 1 - acc/2
    _, co = _scenes(_const_scene(0.0, 0.0), dict(near_far=[0.1, 10.0], step_ratio=0.5))
    c = co.render(rays, 32, white_bg=True, dump=True)
    assert np.allclose(c["rgb_map"], 1.0 - c["acc"][:, None] / 2, atol=1e-6)
    # opaque slab: sigma_feature = 48 -> alpha ~ 1 at the first in-box sample, acc ~ 1, rgb = sigmoid(0) = 0.5
    _, co = _scenes(_const_scene(1.0, 1.0), dict(near_far=[0.1, 10.0], step_ratio=0.5))
    c = co.render(rays, 32, white_bg=True, dump=True)
    first = c["valid"].argmax(1)
    assert np.all(c["weight"][np.arange(2), first] > 0.999) and np.allclose(c["acc"], 1.0, atol=1e-5)
    assert np.allclose(c["rgb_map"], 0.5, atol=1e-5)


def test_single_texel_impulse_bilinear_weights():
    arrs = _const_scene(0.0, 1.0)
    arrs["density_plane.0"][0, 3, 2, 4] = 1.0                       # plane0 = (x: W=6, y: H=7), channel 3, texel (x=4, y=2)
    _, co = _scenes(arrs, dict(near_far=[0.1, 10.0], step_ratio=0.5))
    fx, fy = 4.25, 1.5                                              # un-normalised -> weights (1-.25)*(.5)
    n = np.array([[fx / 5 * 2 - 1, fy / 6 * 2 - 1, 0.3]], np.float32)
    assert abs(co.density_features(n)[0] - 0.75 * 0.5) < 1e-5


# ---- REFTensoRF (models/REFTensoRF.py) ------------------------------------------------------------------------
def test_ref_oracle_reproduces_golden(tiny_ref, tiny_ref_arrays, hyper_tiny):
    sc = TO.scene_from_arrays(tiny_ref_arrays, **hyper_tiny)
    d = TO.execute(sc, tiny_ref["rays"], white_bg=True, N_samples=TINY["N_samples"], dump=True)
    assert np.array_equal(d["app_mask"].numpy().astype(np.uint8), tiny_ref["app_mask"])
    assert np.abs(d["rgb_map"].numpy() - tiny_ref["rgb_map"]).max() < 1e-5
    assert np.abs(d["rgb"].numpy() - tiny_ref["rgb"]).max() < 1e-5
    assert abs(float(sc.penalty) - float(tiny_ref["penalty"])) < 1e-4
    sca = TO.scene_from_arrays(dict(tiny_ref_arrays, alpha_volume=None), **hyper_tiny)
    assert sca.ref is not None and sca.mlp["W1"].shape == (128, 151)


def test_ref_known_answers(tiny_ref_arrays, hyper_tiny):
    """Hand-checkable REFTensoRF.execute arithmetic (:212-239): a constant normal n = +z seen along -z gives d = +z, dot = 1,
    reflection = 2n - d = +z, MLP input 0 = -1, no penalty; seen along +z gives d = -z, dot = -1, reflection = -2z + z = -z,
    penalty = relu(1)^2 = 1 per unit weight.  With W3 = 0 the specular colour is sigmoid(0) = 0.5, so rgb = 0.5 * tint + rgb_d."""
    a = dict(tiny_ref_arrays)
    a["normal_W"] = np.zeros_like(a["normal_W"]); a["normal_b"] = np.array([0, 0, 2.0], np.float32)       # |n| = 2 -> normalised to z
    a["specular_W"] = np.zeros_like(a["specular_W"]); a["specular_b"] = np.array([0.4], np.float32)
    a["diffuse_W"] = np.zeros_like(a["diffuse_W"]); a["diffuse_b"] = np.array([0.1, 0.2, 0.3], np.float32)
    a["W3"] = np.zeros_like(a["W3"]); a["b3"] = np.zeros_like(a["b3"])
    sc = TO.scene_from_arrays(a, **hyper_tiny)
    xyz = torch.zeros((2, 3))
    views = torch.tensor([[0.0, 0.0, -1.0], [0.0, 0.0, 1.0]])
    w = torch.tensor([0.25, 0.5])
    rgb = TO.shade_ref(sc, xyz, views, w)
    assert torch.allclose(rgb, torch.tensor([[0.3, 0.4, 0.5]] * 2), atol=1e-6)
    assert abs(float(sc.penalty) - 0.5) < 1e-6                    # only the back-facing sample (dot = -1) is penalised: 0.5 * 1
    f, rgb_d, tint, normal, rho = TO.compute_appfeature_ref(sc, xyz)
    nn = TO.jt_normalize(normal)
    dot = ((-views) * nn).sum(1, keepdim=True)
    refl = 2 * dot * nn - (-views)
    assert torch.allclose(dot.view(-1), torch.tensor([1.0, -1.0])) and torch.allclose(refl, torch.tensor([[0, 0, 1.0], [0, 0, -1.0]]))
    _, mlp_in = TO.mlp_render_fea_ref(sc, refl, f, -dot, return_in=True)
    assert mlp_in.shape == (2, 151) and torch.allclose(mlp_in[:, 0], torch.tensor([-1.0, 1.0]))
    assert torch.equal(mlp_in[:, 1:28], f) and torch.equal(mlp_in[:, 28:31], refl)
    # a zero normal stays zero (eps floor of jt.normalize), never NaN
    assert torch.equal(TO.jt_normalize(torch.zeros(1, 3)), torch.zeros(1, 3))


# ---- NerfPlusPlus (second half; models/nerfplusplus.py) ----------------------------------------------------------------
def test_npp_oracle_reproduces_golden_and_known_answers(tiny_npp, tiny_npp_arrays, hyper_tiny):
    sc = TO.scene_from_arrays(tiny_npp_arrays, **hyper_tiny)
    d = TO.execute_npp(sc, tiny_npp["rays"], N_samples=TINY["N_samples"], rand_fg=tiny_npp["rand_fg"], rand_bg=tiny_npp["rand_bg"], dump=True)
    assert np.array_equal(d["z_vals"].numpy(), tiny_npp["out.z_vals"])
    assert np.array_equal(d["app_mask"].numpy().astype(np.uint8), tiny_npp["out.app_mask"])
    assert np.abs(d["rgb_map"].numpy() - tiny_npp["out.rgb_map"]).max() < 1e-5
    # sampling (nerfplusplus.py:239-256): ascending, first sample inside [near, near + half a step], last one on or inside the sphere
    rays = torch.tensor(tiny_npp["rays"])
    z = d["z_vals"]
    assert bool((z[:, 1:] > z[:, :-1]).all()) and bool((z[:, 0] >= hyper_tiny["near_far"][0]).all())
    end = rays[:, :3] + rays[:, 3:6] * z[:, -1:]
    assert bool((end.norm(dim=-1) <= 6.0 + 1e-4).all())
    far = TO.intersect_sphere(rays[:, :3], rays[:, 3:6], 36.0)
    assert torch.allclose((rays[:, :3] + rays[:, 3:6] * far[:, None]).norm(dim=-1), torch.full((rays.shape[0],), 6.0), atol=1e-4)
    # inverted-sphere points (:207-237) lie on the unit... on the radius-`radii` sphere, 4th coordinate = the depth parameter
    pts, _ = TO.depth2pts_outside(rays[:, None, :3].expand(-1, 5, -1), rays[:, None, 3:6].expand(-1, 5, -1), torch.linspace(0.5, 5.5, 5).expand(rays.shape[0], 5), 6.0)
    assert torch.allclose(pts[..., :3].norm(dim=-1), torch.full(pts.shape[:-1], 6.0), atol=1e-3) and torch.equal(pts[0, :, 3], torch.linspace(0.5, 5.5, 5))
    # composition (:311-314): background enters only where more than 10 % of the light passes the foreground
    lam = d["bg_lambda"]
    assert bool(((lam == 0) | (lam > 0.1)).all())
    assert torch.allclose(d["rgb_map"], d["fg_rgb_map"] + d["bg_rgb_map"])
    assert bool((d["bg_rgb_map"][lam == 0] == 0).all())
    # embedder / network shapes (:7-56, :147-163)
    assert TO.embed(torch.zeros(2, 4), 3, 4).shape == (2, 36) and TO.embed(torch.zeros(2, 3), 1, 2).shape == (2, 15)
    assert sc.npp["net"]["base_layers.3.0.weight"].shape == (128, 164) and sc.npp["net"]["rgb_layers.0.weight"].shape == (64, 271)
