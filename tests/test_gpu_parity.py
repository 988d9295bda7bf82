"""GPU (-m gpu): the HIP path, called through the C-ABI (ctypes -> libtvr.so), against the oracle and the
committed golden vectors.  Bars (BASELINE.json north_star): sample positions, masks and cell indices BIT-EXACT;
RGB L-infinity <= 1e-3 (fp32; tolerance written below per quantity).  The oracle is only the checker here."""
import numpy as np
import pytest
import torch

from conftest import TINY, make_model

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-3          # the parity bar of north_star
RGB_TIGHT = 2e-4        # what we actually hold: one threshold flip moves RGB by <= weight ~ 1e-4


def _np(t):
    return t.detach().cpu().numpy()


def _check_dense(d, g, eps_exact=True):
    """d: dict of device tensors from render_rays(dense=True); g(k): golden array accessor."""
    assert np.array_equal(_np(d["t_min"]), g("t_min"))
    assert np.array_equal(_np(d["z"]), g("z_vals"))                                  # bit-exact positions
    assert np.array_equal(_np(d["bbox_valid"]), g("bbox_valid"))
    assert np.array_equal(_np(d["valid"]), g("valid"))                               # bit-exact masks
    v = g("valid").astype(bool)
    assert np.array_equal(_np(d["cell"])[v], g("cell")[v])                           # bit-exact cell indices
    assert np.abs(_np(d["sigma"]) - g("sigma")).max() <= 1e-4 * max(1.0, np.abs(g("sigma")).max())
    assert np.abs(_np(d["weight"]) - g("weight")).max() < 2e-6
    flips = (_np(d["weight"]) > 1e-4) != g("app_mask").astype(bool)
    assert flips.sum() <= 2, f"app-mask Hamming distance {flips.sum()}"


def test_tiny_dense_bit_exact_indices_and_rgb(tiny_dump, tiny_arrays, hyper_tiny):
    m = make_model(tiny_arrays, hyper_tiny)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    rgb, depth, d = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
    _check_dense(d, lambda k: tiny_dump[f"out.{k}"])
    assert np.abs(_np(d["sigma_feature"]) - tiny_dump["out.sigma_feature"]).max() < 5e-5
    assert np.abs(_np(d["alpha"]) - tiny_dump["out.alpha"]).max() < 2e-6
    assert np.abs(_np(d["rgb"]) - tiny_dump["out.rgb"]).max() < 1e-4                 # per-sample MLP output
    assert np.abs(_np(d["acc"]) - tiny_dump["out.acc_map"]).max() < 1e-5
    assert np.abs(_np(rgb) - tiny_dump["out.rgb_map"]).max() < RGB_TIGHT < RGB_TOL
    assert np.abs(_np(depth) - tiny_dump["out.depth_map"]).max() < 1e-4
    # additional_output=True tuple of TensorBase.execute (tensorBase.py:533-534)
    out = m(rays, is_train=False, white_bg=True, N_samples=TINY["N_samples"], additional_output=True)
    assert len(out) == 7 and out[2].shape == (64, 48, 3) and out[6].shape == (64, 1)


@pytest.mark.parametrize("name,wb,am,jit", [("wb1_am0", True, False, False), ("wb0_am0", False, False, False),
                                            ("wb1_am1", True, True, False), ("wb0_am1_jit", False, True, True)])
def test_edge_cases(tiny_edge, tiny_arrays, hyper_tiny, name, wb, am, jit):
    arrs = dict(tiny_arrays)
    if am:
        arrs["alpha_volume"], arrs["alpha_aabb"] = tiny_edge["alpha_volume"], tiny_edge["alpha_aabb"]
    m = make_model(arrs, hyper_tiny)
    rays = torch.tensor(tiny_edge["rays"], device="cuda")
    jitter = torch.tensor(tiny_edge["jitter"], device="cuda") if jit else None
    g = lambda k: tiny_edge[f"{name}.{k}"]
    for eps in (0.0, None):                                     # exact mode and default early termination
        rgb, depth, d = m.render_rays(rays, white_bg=wb, N_samples=TINY["N_samples"], jitter=jitter, eps_T=eps, dense=True)
        if eps == 0.0:
            _check_dense(d, g)
        assert np.abs(_np(rgb) - g("rgb_map")).max() < RGB_TIGHT
        assert np.abs(_np(depth) - g("depth_map")).max() < (1e-4 if eps == 0.0 else 1e-3)
    assert np.allclose(_np(rgb)[3], 1.0 if wb else 0.0) and _np(depth)[3] == tiny_edge["rays"][3, 5]   # ray missing the box


def test_config1_against_golden(config1_golden):
    """BASELINE.json configs[0] at full size: 128^3, 64x64 rays, 192 samples."""
    from jittor_myc_nerfs_amd import rays as R, synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    hyper = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
    m = make_model(arrs, hyper)
    rays = torch.tensor(config1_golden["rays"], device="cuda")      # the fixture carries its inputs
    rgb0, depth0, d = m.render_rays(rays, white_bg=True, N_samples=B["N_samples"], eps_T=0.0, dense=True)
    assert np.array_equal(np.packbits(_np(d["valid"])), config1_golden["valid_bits"])                  # bit-exact mask
    app = (_np(d["weight"]) > 1e-4).astype(np.uint8)
    assert (np.unpackbits(np.packbits(app)) != np.unpackbits(config1_golden["app_bits"])).sum() <= 16
    assert np.abs(_np(rgb0) - config1_golden["rgb_map"]).max() < RGB_TIGHT
    assert np.abs(_np(d["acc"]) - config1_golden["acc_map"]).max() < 2e-5
    # default early termination (eps_T = 1e-4) stays within the parity bar; the renderer's chunk loop is bit-invariant
    from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast
    parts = [m(rays[c0:c0 + 1000], is_train=False, white_bg=True, N_samples=B["N_samples"]) for c0 in range(0, rays.shape[0], 1000)]
    rgb1, depth1 = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])          # 5 separate tvr_render calls, ragged tail
    assert np.abs(_np(rgb1) - config1_golden["rgb_map"]).max() < 3e-4 < RGB_TOL
    rgb1b, _, depth1b, _, _ = OctreeRender_trilinear_fast(rays, m, chunk=1000, N_samples=B["N_samples"], white_bg=True)
    assert torch.equal(rgb1b, rgb1) and torch.equal(depth1b, depth1)                            # the renderer's merged call: same bits
    rgb2, depth2 = m(rays, is_train=False, white_bg=True, N_samples=B["N_samples"])
    assert torch.equal(rgb1, rgb2) and torch.equal(depth1, depth2)                  # chunking does not change a bit
    rgb3, depth3 = m(rays, is_train=False, white_bg=True, N_samples=B["N_samples"])
    assert torch.equal(rgb2, rgb3) and torch.equal(depth2, depth3)                  # run-to-run deterministic
    perm = torch.randperm(rays.shape[0], device="cuda")
    rgbp, depthp = m(rays[perm], is_train=False, white_bg=True, N_samples=B["N_samples"])
    assert torch.equal(rgbp, rgb2[perm]) and torch.equal(depthp, depth2[perm])     # per-ray results ignore batch order


def test_run_to_run_determinism(config1_golden):
    """Bitwise reproducibility of the whole pipeline (caught an exec-masked-load -> MFMA hazard once): 4 repeats, per-sample rgb."""
    from jittor_myc_nerfs_amd import synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    rays = torch.tensor(config1_golden["rays"], device="cuda")
    ref = None
    for rep in range(4):
        junk = torch.full((32 << 20,), 0xFF, dtype=torch.uint8, device="cuda")      # poison what the allocator hands out next
        del junk
        rgb, depth, d = m.render_rays(rays, white_bg=True, N_samples=B["N_samples"], dense=True)
        cur = (rgb.clone(), depth.clone(), d["rgb"].clone(), d["weight"].clone())
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert torch.equal(a, b), f"repeat {rep} differs: max |d| = {float((a - b).abs().max())}"


def test_feature_apis_against_golden_and_oracle(tiny_dump, tiny_arrays, hyper_tiny):
    from oracle import c_oracle as CO, tensorf_oracle as TO
    m = make_model(tiny_arrays, hyper_tiny)
    xyz = torch.tensor(tiny_dump["app_xyz_norm"], device="cuda")
    f = m.compute_appfeature(xyz)
    assert f.shape == (xyz.shape[0], 27)
    assert np.abs(_np(f) - tiny_dump["app_feature"]).max() < 1e-4 * max(1.0, np.abs(tiny_dump["app_feature"]).max())
    with torch.no_grad():                                          # tvr_mlp_render (with gradients enabled the module runs library GEMMs under autograd instead)
        rgb = m.renderModule(xyz, torch.tensor(tiny_dump["app_dirs"], device="cuda"), torch.tensor(tiny_dump["app_feature"], device="cuda"))
    assert np.abs(_np(rgb) - tiny_dump["app_rgb"]).max() < 1e-5
    rgb_ag = m.renderModule(xyz, torch.tensor(tiny_dump["app_dirs"], device="cuda"), torch.tensor(tiny_dump["app_feature"], device="cuda"))
    assert rgb_ag.requires_grad and np.abs(_np(rgb_ag) - tiny_dump["app_rgb"]).max() < 1e-5
    # arbitrary coordinates incl. outside [-1,1] (zeros padding) and exactly on the faces; ragged + empty sizes
    sc = TO.scene_from_arrays(tiny_arrays, **hyper_tiny)
    co = CO.COracle(tiny_arrays, step=float(sc.stepSize), **hyper_tiny)
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1.3, 1.3, (1001, 3)).astype(np.float32)
    pts[:6] = [[1, 1, 1], [-1, -1, -1], [1, -1, 0.5], [0, 0, 0], [1.2, 0, 0], [-1.0, 1.0, -1.0]]
    sf = m.compute_densityfeature(torch.tensor(pts, device="cuda"))
    ref = co.density_features(pts)
    assert np.abs(_np(sf) - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    fa = m.compute_appfeature(torch.tensor(pts, device="cuda"))
    refa = co.app_features(pts)
    assert np.abs(_np(fa) - refa).max() < 1e-4 * max(1.0, np.abs(refa).max())
    assert m.compute_densityfeature(torch.zeros(0, 3, device="cuda")).shape == (0,)
    assert m.compute_appfeature(torch.zeros(0, 3, device="cuda")).shape == (0, 27)
    rgb0, depth0 = m(torch.zeros(0, 6, device="cuda"))
    assert rgb0.shape == (0, 3) and depth0.shape == (0,)


def test_alpha_mask_sample_and_compute_alpha(tiny_edge, tiny_arrays, hyper_tiny):
    from oracle import c_oracle as CO, tensorf_oracle as TO
    from jittor_myc_nerfs_amd import AlphaGridMask
    arrs = dict(tiny_arrays, alpha_volume=tiny_edge["alpha_volume"], alpha_aabb=tiny_edge["alpha_aabb"])
    sc = TO.scene_from_arrays(arrs, **hyper_tiny)
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper_tiny)
    am = AlphaGridMask("cuda", arrs["alpha_aabb"], torch.tensor(arrs["alpha_volume"]))
    pts = np.random.default_rng(9).uniform(-1.7, 1.7, (777, 3)).astype(np.float32)
    got = _np(am.sample_alpha(torch.tensor(pts, device="cuda")))
    ref = co.alpha_samples(pts)
    assert np.abs(got - ref).max() < 1e-6 and np.array_equal(got > 0, ref > 0)
    m = make_model(arrs, hyper_tiny)
    a = m.compute_alpha(torch.tensor(pts, device="cuda"), float(m.stepSize))
    assert a.shape == (777,) and float(a.max()) <= 1.0 and float(a[torch.tensor(ref <= 0, device="cuda")].abs().max()) == 0.0


def test_parameter_update_repacks(tiny_dump, tiny_arrays, hyper_tiny):
    m = make_model(tiny_arrays, hyper_tiny)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    rgb_a, _ = m(rays, N_samples=48)
    with torch.no_grad():
        m.density_plane[0].mul_(0.0)
        m.density_plane[1].mul_(0.0)
        m.density_plane[2].mul_(0.0)
    rgb_b, _ = m(rays, N_samples=48)                                # in-place edit is picked up: only softplus(-10) haze left
    assert float((rgb_b - 1.0).abs().max()) < 1e-2 and float((rgb_a - rgb_b).abs().max()) > 0.1


def test_full_size_properties_config2():
    """BASELINE.json configs[1] at full size (300^3 grid, 800x800 rays, 512 samples): size-independent properties
    + a stratified 16 384-ray set (random / silhouette / box-grazing) against the scalar oracle."""
    from jittor_myc_nerfs_amd import rays as R, synthetic
    from oracle import c_oracle as CO, tensorf_oracle as TO
    A = synthetic.SCENE_A
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"])
    hyper = dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"])
    m = make_model(arrs, hyper)
    assert m.nSamples == 1036
    rays = R.frame_rays(R.sphere_poses(8, A["cam_radius"])[0], 800, 800, A["camera_angle_x"]).cuda()
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    rgb, depth = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"], stats=stats)
    torch.cuda.synchronize()
    st = stats.cpu().numpy()
    assert 0 < st[2] < st[0] <= st[1] <= 640000 * 512
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 and bool(torch.isfinite(depth).all())
    # chunk invariance at full size: 65536-ray chunks == one 640000-ray call, bit for bit
    parts = [m.render_rays(rays[c0:c0 + 65536], white_bg=True, N_samples=A["N_samples"]) for c0 in range(0, rays.shape[0], 65536)]
    rgb_c, depth_c = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    assert torch.equal(rgb_c, rgb) and torch.equal(depth_c, depth)
    # BASELINE configs[3]: the frame as 157 DIRECT tvr_render calls of 4096 rays (train.py:225-226 batch size; renderer.py:16-25 without
    # the chunk merging OctreeRender_trilinear_fast applies at inference), no host sync in between: bit-identical again
    calls = [m.render_rays(rays[c0:c0 + 4096], white_bg=True, N_samples=A["N_samples"]) for c0 in range(0, rays.shape[0], 4096)]
    assert len(calls) == 157 and calls[-1][0].shape[0] == 640000 - 156 * 4096
    assert torch.equal(torch.cat([c[0] for c in calls]), rgb) and torch.equal(torch.cat([c[1] for c in calls]), depth)
    # batch-order invariance at full size: a random permutation of the rays gives the permuted image, bit for bit (which 32-entry tile, lane,
    # wave or workgroup a sample lands in does not matter; neither does which wave holds a SIMD's matrix token when)
    perm = torch.randperm(640000, generator=torch.Generator().manual_seed(7)).cuda()
    rgb_p, depth_p = m.render_rays(rays[perm].contiguous(), white_bg=True, N_samples=A["N_samples"])
    assert torch.equal(rgb_p, rgb[perm]) and torch.equal(depth_p, depth[perm])
    # black background: the same weights, so rgb_white - rgb_black = 1 - acc on every channel wherever neither image is clamped
    rgb_k, depth_k = m.render_rays(rays, white_bg=False, N_samples=A["N_samples"])
    assert torch.equal(depth_k, depth)
    d = rgb - rgb_k
    free = ((rgb < 1.0) & (rgb_k > 0.0)).all(dim=1)
    assert int(free.sum()) > 100000
    assert float((d[free] - d[free][:, :1]).abs().max()) < 1e-6 and float(d[free].min()) > -1e-6
    # oracle on a STRATIFIED set of 16 384 rays at full size: 8192 uniformly random pixels of this frame, 4096 of its silhouette pixels
    # (0.05 < acc < 0.95: weights spread over many samples; early termination and the weight threshold decide what is shaded) and 4096
    # box-grazing / box-missing rays (in-box path < 1.0 of up to 4.3, or none at all: few samples, entry and exit faces close together, the
    # in-box mask decides everything) — this pose sees nothing but box, so those come from poses 1 and 3, whose frames cut the box's silhouette
    g = torch.Generator().manual_seed(1)
    acc_img = (1.0 - d[:, 0]).cpu()
    sil = torch.nonzero((acc_img > 0.05) & (acc_img < 0.95) & free.cpu()).view(-1)
    assert sil.numel() >= 4096, sil.numel()
    sel = torch.cat([torch.randperm(640000, generator=g)[:8192], sil[torch.randperm(sil.numel(), generator=g)[:4096]]]).cuda()
    lo, hi = torch.tensor(A["aabb"][0]), torch.tensor(A["aabb"][1])
    extra = []
    for pose in (1, 3):
        rp = R.frame_rays(R.sphere_poses(8, A["cam_radius"])[pose], 800, 800, A["camera_angle_x"])
        dd = torch.where(rp[:, 3:] == 0, torch.full_like(rp[:, 3:], 1e-6), rp[:, 3:])
        ta, tb = (lo - rp[:, :3]) / dd, (hi - rp[:, :3]) / dd
        path = torch.maximum(ta, tb).min(dim=1).values - torch.minimum(ta, tb).max(dim=1).values
        gz = torch.nonzero(path < 1.0).view(-1)
        assert gz.numel() >= 2048 and int((path[gz] > 0).sum()) > 500 and int((path[gz] <= 0).sum()) > 500
        extra.append(rp[gz[torch.randperm(gz.numel(), generator=g)[:2048]]])
    extra = torch.cat(extra).cuda()
    rgb_x, depth_x = m.render_rays(extra, white_bg=True, N_samples=A["N_samples"])
    test_rays = torch.cat([rays[sel], extra])
    got_rgb, got_depth = torch.cat([rgb[sel], rgb_x]), torch.cat([depth[sel], depth_x])
    assert test_rays.shape[0] == 16384
    sc = TO.scene_from_arrays(arrs, **hyper)
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)
    ref = co.render(_np(test_rays), A["N_samples"], white_bg=True, nthreads=16)
    err_rgb = np.abs(_np(got_rgb) - ref["rgb_map"]).max(axis=1)
    err_dep = np.abs(_np(got_depth) - ref["depth_map"])
    print("full-size oracle subset: RGB L-inf random %.2e, silhouette %.2e, grazing %.2e; depth %.2e" %
          (err_rgb[:8192].max(), err_rgb[8192:12288].max(), err_rgb[12288:].max(), err_dep.max()))
    assert err_rgb.max() < 3e-4 < RGB_TOL
    assert err_dep.max() < 2e-3
    # empty scene (all density factors zero, relu activation): image = white exactly, depth = d_z exactly
    with torch.no_grad():
        for p in m.density_plane:
            p.zero_()
    m.fea2denseAct = "relu"
    m._drop_scene()
    rgb_e, depth_e = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"])
    assert bool((rgb_e == 1.0).all()) and torch.equal(depth_e, rays[:, 5])


def test_filter_rays_kernel_equals_the_reference_formulation(tiny_arrays, hyper_tiny, tiny_dump, tiny_edge):
    """tvr_filter_rays (one pass per call) against the ORACLE's restatement of filtering_rays (tensorBase.py:411-441, oracle/tensorf_oracle.py::filtering_rays_mask;
    the alpha volume it samples is the oracle's own updateAlphaMask, checked equal to the product's first): identical masks for both filters on 40 k perturbed rays —
    grazing rays, rays that miss the box, axis-parallel rays with zero direction components, origins inside the box — with the inputs on the host and on the
    device; and the kept rays are the masked rays."""
    from oracle import tensorf_oracle as TO
    m = make_model(tiny_arrays, hyper_tiny)
    m.updateAlphaMask((9, 8, 7))
    sc = TO.scene_from_arrays(tiny_arrays, **hyper_tiny)
    vol, _ = TO.updateAlphaMask(sc, (9, 8, 7), m.alphaMask_thres)
    assert torch.equal(vol, m.alphaMask.alpha_volume.cpu().view(vol.shape)), "the product's alpha volume differs from the oracle's: the filter test would compare two masks"
    sc.alpha_volume, sc.alpha_aabb = vol, sc.aabb.clone()
    rng = np.random.default_rng(11)
    base = np.concatenate([tiny_dump["rays"], tiny_edge["rays"]]).astype(np.float32)
    rays = np.concatenate([base] * (40000 // base.shape[0] + 1))[:40000].copy()
    rays[:, :3] += 0.15 * rng.standard_normal((rays.shape[0], 3)).astype(np.float32)
    rays[:, 3:] += 0.05 * rng.standard_normal((rays.shape[0], 3)).astype(np.float32)
    rays[::17, 3] = 0.0                                                        # zero direction components: the reference divides by 1e-6 instead
    rays[::29, 4:6] = 0.0
    rays[::31, :3] = 0.05 * rng.standard_normal((rays[::31].shape[0], 3)).astype(np.float32)       # origins inside the box
    lo, hi = np.asarray(tiny_arrays["aabb"], np.float32)
    graze = np.arange(7, rays.shape[0], 37)                                   # rays that graze the box: origin ON a face plane, direction inside that plane
    rays[graze, 0] = hi[0]; rays[graze, 3] = 0.0
    graze2 = np.arange(11, rays.shape[0], 41)                                 # ... and along an edge: two coordinates on faces, one free direction
    rays[graze2, 1] = lo[1]; rays[graze2, 2] = hi[2]; rays[graze2, 4:6] = 0.0; rays[graze2, 3] = 1.0
    rays[-1] = [5, 5, 5, 0, 0, 1]
    host = torch.tensor(rays)
    rgbs = torch.arange(rays.shape[0], dtype=torch.float32).view(-1, 1).expand(-1, 3).contiguous()
    for bbox_only, S in ((True, 256), (False, 48), (False, 256)):
        want = TO.filtering_rays_mask(sc, host, N_samples=S, bbox_only=bbox_only)
        assert 0 < int(want.sum()) < rays.shape[0]
        for src in (host, host.cuda()):
            kept, kept_rgb = m.filtering_rays(src, rgbs.to(src.device), N_samples=S, bbox_only=bbox_only)
            assert kept.device == src.device and kept.shape[0] == int(want.sum())
            assert torch.equal(kept_rgb[:, 0].cpu(), rgbs[want][:, 0]), f"bbox_only={bbox_only}, N_samples={S}: the kernel keeps other rays than the reference's formulas"


def test_scene_maintenance_ops(tiny_arrays, hyper_tiny, tiny_dump):
    """getDenseAlpha / updateAlphaMask / filtering_rays / upsample_volume_grid / shrink feed the render path."""
    from oracle import c_oracle as CO, tensorf_oracle as TO
    m = make_model(tiny_arrays, hyper_tiny)
    sc = TO.scene_from_arrays(tiny_arrays, **hyper_tiny)
    g = [9, 8, 7]
    alpha, dense = m.getDenseAlpha(g)
    ref_alpha, ref_dense = TO.getDenseAlpha(sc, g)
    assert alpha.shape == (9, 8, 7) and np.array_equal(_np(dense), ref_dense.numpy())
    assert np.abs(_np(alpha) - ref_alpha.numpy()).max() < 1e-5
    new_aabb = m.updateAlphaMask(g)
    ref_vol, ref_aabb = TO.updateAlphaMask(sc, g, m.alphaMask_thres)
    assert tuple(m.alphaMask.alpha_volume.shape) == (1, 1, 7, 8, 9)
    assert (_np(m.alphaMask.alpha_volume).reshape(7, 8, 9) != ref_vol.numpy()).sum() <= 1        # threshold at 1e-4: <= 1 borderline cell
    assert np.allclose(_np(new_aabb), ref_aabb.numpy(), atol=1e-6)
    # rendering with the freshly built mask == oracle rendering with the same mask
    arrs = dict(tiny_arrays, alpha_volume=_np(m.alphaMask.alpha_volume).reshape(7, 8, 9), alpha_aabb=tiny_arrays["aabb"])
    sca = TO.scene_from_arrays(arrs, **hyper_tiny)
    ref = CO.COracle(arrs, step=float(sca.stepSize), **hyper_tiny).render(tiny_dump["rays"], TINY["N_samples"], white_bg=True)
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    rgb, depth = m(rays, N_samples=TINY["N_samples"])
    assert np.abs(_np(rgb) - ref["rgb_map"]).max() < RGB_TIGHT
    # filtering_rays: bbox-only and alpha-mask variants keep exactly the rays the host formulas say
    rr = torch.tensor(np.concatenate([tiny_dump["rays"], [[5, 5, 5, 0, 0, 1]]]).astype(np.float32))
    kept, kept_rgb = m.filtering_rays(rr, torch.zeros(rr.shape[0], 3), bbox_only=True)
    o, d = rr[:, :3].numpy(), rr[:, 3:6].numpy()
    v = np.where(d == 0, np.float32(1e-6), d)
    ra, rb = (tiny_arrays["aabb"][1] - o) / v, (tiny_arrays["aabb"][0] - o) / v
    hit = np.maximum(ra, rb).min(1) > np.minimum(ra, rb).max(1)                # tensorBase.py:428-430
    assert kept.shape[0] == int(hit.sum()) < rr.shape[0] and not hit[-1]      # the appended ray misses the box
    kept2, _ = m.filtering_rays(rr, torch.zeros(rr.shape[0], 3), N_samples=48)
    xyz, _, _ = m.sample_ray(rr[:, :3].cuda(), rr[:, 3:6].cuda(), is_train=False, N_samples=48)
    expect = (torch.tensor(CO.COracle(arrs, step=float(sca.stepSize), **hyper_tiny).alpha_samples(_np(xyz).reshape(-1, 3))).view(rr.shape[0], 48) > 0).any(1)
    assert kept2.shape[0] == int(expect.sum())
    # upsample_volume_grid (bilinear, align_corners) then render == oracle on the same upsampled factors
    m2 = make_model(tiny_arrays, hyper_tiny)
    m2.upsample_volume_grid([20, 24, 28])
    assert tuple(m2.density_plane[0].shape) == (1, 16, 24, 20) and tuple(m2.app_line[0].shape) == (1, 48, 28, 1)
    up = dict(tiny_arrays, gridSize=np.array([20, 24, 28]))
    for i in range(3):
        for k in ("density_plane", "density_line", "app_plane", "app_line"):
            up[f"{k}.{i}"] = _np(getattr(m2, k)[i])
    scu = TO.scene_from_arrays(up, **hyper_tiny)
    assert abs(float(scu.stepSize) - float(m2.stepSize)) == 0 and scu.nSamples == m2.nSamples
    refu = CO.COracle(up, step=float(scu.stepSize), **hyper_tiny).render(tiny_dump["rays"], TINY["N_samples"], white_bg=True)
    rgbu, _ = m2(rays, N_samples=TINY["N_samples"])
    assert np.abs(_np(rgbu) - refu["rgb_map"]).max() < RGB_TIGHT
    # shrink to the mask's box: parameter slices + new aabb / step size, still renders against the oracle
    m3 = make_model(tiny_arrays, hyper_tiny)
    na = m3.updateAlphaMask([16, 20, 24])
    m3.shrink(na)
    sh = {"aabb": m3.aabb.numpy(), "gridSize": m3.gridSize.numpy()}
    for i in range(3):
        for k in ("density_plane", "density_line", "app_plane", "app_line"):
            sh[f"{k}.{i}"] = _np(getattr(m3, k)[i])
    for k in ("basis_mat", "W1", "b1", "W2", "b2", "W3", "b3"):
        sh[k] = tiny_arrays[k]
    sh["alpha_volume"] = _np(m3.alphaMask.alpha_volume).reshape(24, 20, 16)
    sh["alpha_aabb"] = m3.alphaMask.aabb.numpy()
    scs = TO.scene_from_arrays(sh, **hyper_tiny)
    refs = CO.COracle(sh, step=float(scs.stepSize), **hyper_tiny).render(tiny_dump["rays"], TINY["N_samples"], white_bg=True)
    rgbs, _ = m3(rays, N_samples=TINY["N_samples"])
    assert np.abs(_np(rgbs) - refs["rgb_map"]).max() < RGB_TIGHT


def test_render_is_hip_graph_capturable(config1_golden):
    """The C-ABI never allocates or synchronises, so a whole tvr_render (memset + 3 kernels) can be captured into a hipGraph and
    replayed; the replay is bit-identical to the eager call."""
    from jittor_myc_nerfs_amd import synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    rays = torch.tensor(config1_golden["rays"], device="cuda")
    eager_rgb, eager_depth = m.render_rays(rays, white_bg=True, N_samples=B["N_samples"])      # also builds scene + scratch
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.render_rays(rays, white_bg=True, N_samples=B["N_samples"])                            # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap_rgb, cap_depth = m.render_rays(rays, white_bg=True, N_samples=B["N_samples"])
    cap_rgb.zero_(); cap_depth.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap_rgb, eager_rgb) and torch.equal(cap_depth, eager_depth)


def test_evaluation_loop_on_gpu(tmp_path, tiny_arrays, hyper_tiny):
    """renderer.py:29-91 driven end to end: transforms json -> rays -> OctreeRender_trilinear_fast (chunk 1024) -> PNG + PSNR."""
    import json
    from PIL import Image
    from jittor_myc_nerfs_amd import BlenderRays, OctreeRender_trilinear_fast, evaluation, rays as R
    from oracle import c_oracle as CO, tensorf_oracle as TO
    m = make_model(tiny_arrays, hyper_tiny)
    poses = R.sphere_poses(2, 4.0)
    meta = {"camera_angle_x": 0.6911, "frames": [{"file_path": f"./test/r_{i}", "transform_matrix": M.tolist()} for i, M in enumerate(poses)]}
    with open(tmp_path / "transforms_test.json", "w") as f:
        json.dump(meta, f)
    ds = BlenderRays(str(tmp_path), split="test", downsample=25.0, near=2.0, far=6.0)     # 32x32
    class A: expname = "tiny"
    evaluation(ds, m, A, OctreeRender_trilinear_fast, savePath=str(tmp_path / "o"), N_vis=-1, N_samples=48, white_bg=True, device="cuda")
    img = np.asarray(Image.open(tmp_path / "o" / "tiny_r_1.png")).astype(np.float32) / 255.0
    sc = TO.scene_from_arrays(tiny_arrays, **hyper_tiny)
    ref = CO.COracle(tiny_arrays, step=float(sc.stepSize), **hyper_tiny).render(ds.all_rays[1].numpy(), 48, white_bg=True)
    assert np.abs(img.reshape(-1, 3) - ref["rgb_map"]).max() < 1.0 / 255 + 1e-3         # 8-bit quantisation + parity bar


@pytest.mark.parametrize("grid,S", [([16, 20, 24], 2000), ([16, 20, 24], 4096), ([2500, 4, 5], 96), ([700, 6, 300], 64)])
def test_march_launch_shapes(hyper_tiny, grid, S):
    """The march launcher's other shapes: long sample lists shrink the workgroup (16 -> 12/8/4 waves so that lines + lists fit the 160 KB
    LDS), and grids whose three density lines do not fit fall back to global line reads.  Same parity as the default shape: bit-exact masks /
    cells against the scalar oracle, RGB within the tight bar."""
    from jittor_myc_nerfs_amd import rays as R, synthetic
    from oracle import c_oracle as CO, tensorf_oracle as TO
    aabb = TINY["aabb"]
    arrs = synthetic.make_scene_arrays(grid, aabb, seed=5)
    hyper = dict(hyper_tiny)
    m = make_model(arrs, hyper)
    rays = R.frame_rays(R.sphere_poses(4, 4.0)[2], 12, 12, 0.6911)
    sc = TO.scene_from_arrays(arrs, **hyper)
    ref = CO.COracle(arrs, step=float(sc.stepSize), **hyper).render(rays.numpy(), S, white_bg=True, dump=True, nthreads=8)
    rgb, depth, d = m.render_rays(rays.cuda(), white_bg=True, N_samples=S, eps_T=0.0, dense=True)
    assert np.array_equal(_np(d["valid"]), ref["valid"]) and np.array_equal(_np(d["cell"])[ref["valid"] > 0], ref["cell"][ref["valid"] > 0])
    assert np.abs(_np(rgb) - ref["rgb_map"]).max() < RGB_TIGHT
    rgb2, depth2 = m.render_rays(rays.cuda(), white_bg=True, N_samples=S)           # the non-dense instantiation, default eps_T
    assert np.abs(_np(rgb2) - ref["rgb_map"]).max() < 3e-4
    # and the training march (forward + backward launch shapes) runs on the same scene
    m.eps_T = 0.0
    out, _ = m.render_rays_autograd(rays.cuda(), white_bg=True, N_samples=S)
    out.sum().backward()
    assert np.abs(_np(out) - ref["rgb_map"]).max() < 3e-4 and bool(torch.isfinite(m.density_line[0].grad).all())


@pytest.mark.parametrize("scale", [20.0, 200.0, 1000.0])
def test_large_feature_magnitudes(tiny_dump, tiny_arrays, hyper_tiny, scale):
    """The positional encoding takes sin / cos of the appearance features; the kernel evaluates them with v_sin_f32 / v_cos_f32 after an
    fp32 reduction to revolutions, whose error grows like |v| * 6e-8 rad.  Features 20x and 200x larger than the synthetic scene's
    (|v| up to ~1100) must still render within the parity bar; beyond |v| = 256 the kernel switches to its Cody-Waite polynomial.  (At |v| in
    the thousands sin(v) is ill-conditioned for ANY fp32 evaluation: the features' own rounding, 1e-7 relative, is already 1e-3 rad.)"""
    from oracle import tensorf_oracle as TO
    arrs = dict(tiny_arrays)
    arrs["basis_mat"] = tiny_arrays["basis_mat"] * np.float32(scale)
    sc = TO.scene_from_arrays(arrs, **hyper_tiny)
    d = TO.execute(sc, torch.tensor(tiny_dump["rays"]), white_bg=True, N_samples=TINY["N_samples"], dump=True)
    fmax = float(TO.compute_appfeature(sc, d["xyz_norm"][d["app_mask"]]).abs().max())
    m = make_model(arrs, hyper_tiny)
    rgb, depth, dd = m.render_rays(torch.tensor(tiny_dump["rays"], device="cuda"), white_bg=True, N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
    e_s = np.abs(_np(dd["rgb"]) - d["rgb"].numpy()).max()
    e_m = np.abs(_np(rgb) - d["rgb_map"].numpy()).max()
    print(f"basis x{scale:g}: max |feature| {fmax:.0f}, per-sample rgb Linf {e_s:.2e}, rgb_map Linf {e_m:.2e}")
    assert e_m < RGB_TOL and e_s < RGB_TOL


def test_fp16_range_is_proven_or_reported_never_clipped(tiny_dump, tiny_arrays, hyper_tiny):
    """include/tvr.h: the appearance network's matrix products take their operands through fp16.  (1) A normal scene: the host's interval bounds prove the
    range, the in-kernel check is switched off, and switching it on changes no bit.  (2) A scene whose appearance factors are scaled until the interpolated
    features h reach ~1e6: the bounds fail, the check stays on, every pixel that shades such a sample comes out NaN (never a finite picture made of clipped
    products), pixels that shade nothing are untouched; tvr_app_feature marks the same samples.  (3) "off" is the caller's vouching: finite, clipped, wrong."""
    rays = torch.tensor(tiny_dump["rays"], device="cuda")
    S = TINY["N_samples"]
    m = make_model(tiny_arrays, hyper_tiny)
    rep = m.fp16_range_report()
    assert rep["proven"] and rep["layer2_inputs"] < 6e4, rep
    rgb_auto, _ = m.render_rays(rays, white_bg=True, N_samples=S)
    assert m._range_proven is True
    m.fp16_range_check, m._range_proven = "on", None
    rgb_on, _, d = m.render_rays(rays, white_bg=True, N_samples=S, dense=True)
    assert m._range_proven is False and torch.equal(rgb_on, rgb_auto) and bool(torch.isfinite(rgb_on).all())

    arrs = dict(tiny_arrays)
    for i in range(3):
        arrs[f"app_plane.{i}"] = tiny_arrays[f"app_plane.{i}"] * 2000.0
        arrs[f"app_line.{i}"] = tiny_arrays[f"app_line.{i}"] * 2000.0
    big = make_model(arrs, hyper_tiny)
    rep = big.fp16_range_report()
    assert not rep["proven"] and rep["h"] > 65504 and rep["weights"] < 6e4, rep
    rgb_b, _, db = big.render_rays(rays, white_bg=True, N_samples=S, dense=True)
    assert big._range_proven is False
    shades = (db["weight"] > big.rayMarch_weight_thres).any(1)                  # the density field is the same: same shaded samples
    assert bool(shades.any()) and not bool(shades.all())
    nan_px = torch.isnan(rgb_b).any(1)
    assert bool(nan_px.any()) and not bool((nan_px & ~shades).any()), "a pixel that shades no sample must not be touched"
    assert torch.equal(rgb_b[~shades], rgb_auto[~shades])
    # which samples: the ones whose interpolated h leaves the range — tvr_app_feature on the shaded positions marks the same set (its inputs are the same h)
    xyz = m.normalize_coord(rays[:, None, :3] + rays[:, None, 3:6] * db["z"][..., None])
    sel = db["weight"] > big.rayMarch_weight_thres
    f_big = big.compute_appfeature(xyz[sel])
    bad = torch.isnan(f_big).any(1)
    assert bool(bad.any()) and bool((torch.isnan(f_big).all(1) == bad).all())
    per_ray_bad = torch.zeros_like(sel, dtype=torch.bool)
    per_ray_bad[sel] = bad
    assert torch.equal(per_ray_bad.any(1), nan_px), "NaN pixels are exactly the pixels with an out-of-range sample"
    big.fp16_range_check, big._range_proven = "off", None
    rgb_off, _ = big.render_rays(rays, white_bg=True, N_samples=S)
    assert bool(torch.isfinite(rgb_off).all())                                  # what "silently clipped" looked like: finite and wrong


def test_frame_stream_two_in_flight(config1_golden):
    """render.FrameStream (round 5): two frames in flight on two streams, own scratch and output slots — every frame handed back equals the plain render of its rays bit for bit,
    whatever overlaps with it, and the slots are reused without a frame reading another's queue."""
    from jittor_myc_nerfs_amd import FrameStream, synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    base = torch.tensor(config1_golden["rays"], device="cuda")
    g = torch.Generator(device="cuda").manual_seed(4)
    sets = []
    for k in range(5):
        r = base.clone()
        r[:, :3] += 0.05 * torch.randn(r[:, :3].shape, device="cuda", generator=g)
        sets.append(r[: base.shape[0] - 37 * k].contiguous())                    # frames of different sizes: the slots' outputs are re-made
    want = [tuple(t.clone() for t in m.render_rays(r, white_bg=True, N_samples=B["N_samples"])) for r in sets]
    fs = FrameStream(m, white_bg=True, N_samples=B["N_samples"])
    got = []
    for r in sets:
        o = fs.submit(r)
        if o is not None:
            got.append((o[0].clone(), o[1].clone()))
    o = fs.flush()
    got.append((o[0].clone(), o[1].clone()))
    assert fs.submit(sets[0]) is None                                            # after a flush the stream starts empty again
    fs.flush()
    assert len(got) == len(want)
    for (a, b), (c, d) in zip(got, want):
        assert torch.equal(a, c) and torch.equal(b, d)


def test_autotune_of_the_pieces_changes_no_pixel(config1_golden):
    """field.autotune_render_pieces decides pieces / one launch set on the card (the gain is -4.6 ... +0.6 % by box): it must leave `render_piece_rays` at one of the two
    forms, report both timings, change no pixel, and leave small calls and a caller-fixed piece size alone."""
    from jittor_myc_nerfs_amd import synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    base = torch.tensor(config1_golden["rays"], device="cuda")                      # 4096 rays
    S = B["N_samples"]
    assert m.autotune_render_pieces(base, N_samples=S) is None and m.render_piece_rays is None          # too small for pieces: nothing measured, nothing changed
    big = base.repeat(46, 1).contiguous()                                           # 188 416 rays: six pieces' worth
    m.render_piece_rays = 0
    want = [t.clone() for t in m.render_rays(big, white_bg=True, N_samples=S)]
    m.render_piece_rays = 4096
    assert m.autotune_render_pieces(big, N_samples=S) is None and m.render_piece_rays == 4096           # a piece size the caller fixed stays
    m.render_piece_rays = None
    res = m.autotune_render_pieces([big, big.flip(0).contiguous()], N_samples=S, blocks=1, frames_per_block=2)
    assert res["chosen"] in ("pieces", "one launch set") and res["frames_each"] == 2
    assert res["ms_per_frame_in_pieces"] > 0 and res["ms_per_frame_one_launch_set"] > 0
    assert m.render_piece_rays == (None if res["chosen"] == "pieces" else 0)
    got = m.render_rays(big, white_bg=True, N_samples=S)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_render_in_pieces_equals_one_launch_set(config1_golden):
    """Round 6 (include/tvr.h, PIECES): a tvr_render call of at least two pieces' worth of rays goes out as pieces of consecutive rays on two library-owned streams, forked from
    and joined into the caller's stream.  A ray's result does not depend on the batch it arrives in, so pixels, depths and counters equal the one-launch-set call BIT FOR BIT
    — with jitter, with a ragged last piece, back to back (the two halves of the scratch are reused), under a profile (one event set per piece), captured into a hipGraph;
    `dense` calls are never cut; the scratch query covers two pieces."""
    import ctypes as C
    from jittor_myc_nerfs_amd import _lib as L, synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    base = torch.tensor(config1_golden["rays"], device="cuda")                      # 4096 rays
    g = torch.Generator(device="cuda").manual_seed(11)
    rays = torch.cat([base + torch.cat([0.03 * torch.randn((base.shape[0], 3), device="cuda", generator=g), torch.zeros((base.shape[0], 3), device="cuda")], 1)
                      for _ in range(3)])[: 3 * 4096 - 333].contiguous()           # 11 955 rays
    jit = torch.rand(rays.shape[0], device="cuda", generator=g)
    S = B["N_samples"]
    m.render_piece_rays = 0
    st0 = torch.zeros(8, dtype=torch.int64, device="cuda")
    want = [t.clone() for t in m.render_rays(rays, white_bg=True, N_samples=S, stats=st0)]
    want_j = [t.clone() for t in m.render_rays(rays, white_bg=True, N_samples=S, jitter=jit)]
    for piece in (512, 1024, 1536):                                                 # 24 / 12 / 8 pieces, the last one ragged
        m.render_piece_rays = piece
        sc = m._ensure_scene()
        assert L.lib().tvr_scene_get_render_pieces(sc) == piece
        st1 = torch.zeros(8, dtype=torch.int64, device="cuda")
        prof = C.c_void_p()
        L.check(L.lib().tvr_profile_create(4, C.byref(prof)), "tvr_profile_create")
        for rep in range(2):
            got = m.render_rays(rays, white_bg=True, N_samples=S, stats=st1 if rep == 0 else None, profile=prof)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (piece, rep)
        got_j = m.render_rays(rays, white_bg=True, N_samples=S, jitter=jit)
        assert torch.equal(got_j[0], want_j[0]) and torch.equal(got_j[1], want_j[1])
        torch.cuda.synchronize()
        assert torch.equal(st1[:4], st0[:4])                                        # the same samples evaluated and shaded, the same rays terminated
        ms = (C.c_float * 3)()
        assert L.lib().tvr_profile_read(prof, C.byref(ms)) == 2 and all(x > 0 for x in ms)
        L.lib().tvr_profile_destroy(prof)
        # a call with per-sample outputs stays ONE launch set (and equals the piecewise picture)
        rgb_d, dep_d, dd = m.render_rays(rays, white_bg=True, N_samples=S, dense=True)
        assert torch.equal(rgb_d, want[0]) and dd["weight"].shape == (rays.shape[0], S)
        # captured: the fork and the join are part of the caller's graph
        out = (torch.empty_like(want[0]), torch.empty_like(want[1]))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.render_rays(rays, white_bg=True, N_samples=S, out=out)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            m.render_rays(rays, white_bg=True, N_samples=S, out=out)
        out[0].fill_(-1)
        out[1].fill_(-1)
        gr.replay()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1]), ("graph", piece)
        del gr
    # the size query: two pieces' scratch fits what the query returns, and the library refuses less
    sc = m._ensure_scene()
    need = L.lib().tvr_render_scratch_bytes(sc, rays.shape[0], S)
    assert need >= 2 * L.lib().tvr_render_scratch_bytes(None, 1536, S) - 4096
    assert L.lib().tvr_scene_set_render_pieces(sc, 7) == -1 and L.lib().tvr_scene_set_render_pieces(sc, -1) == 0
    assert L.lib().tvr_scene_get_render_pieces(sc) == 30720


def test_frame_stream_serialises_around_a_scene_update(config1_golden):
    """Parameters written between two submits: the re-pack (tvr_scene_update on the new frame's stream) must not run beside the frame still in flight on the other stream,
    which reads the same packed images — FrameStream drains before such a frame and before the one after it (field.scene_settled).  Every frame equals the serial render
    made with the parameters as they were at ITS submit.  (The drain is by construction: with it disabled the overlap could not be provoked on this pool —
    scripts/debug/frame_stream_race.py, bench frame included: the small pack kernels did not get onto the CUs beside the other frame's persistent grids.  That is
    scheduling luck, not a guarantee.)"""
    from jittor_myc_nerfs_amd import FrameStream, synthetic
    B = synthetic.SCENE_B
    arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
    rays = torch.tensor(config1_golden["rays"], device="cuda")
    rays = torch.cat([rays] * 8)                                                # long enough for a frame to be in flight when the next is submitted
    w0 = m.app_plane[0].detach().clone()
    S = B["N_samples"]

    def schedule(render):
        out = []
        for k in range(6):
            if k in (2, 3, 5):
                with torch.no_grad():
                    m.app_plane[0].mul_(1.0 + 0.05 * k)
            out.append(render(k))
        return out

    want = schedule(lambda k: tuple(t.clone() for t in m.render_rays(rays, white_bg=True, N_samples=S)))
    assert not torch.equal(want[1][0], want[2][0]) and not torch.equal(want[2][0], want[3][0]) and not torch.equal(want[4][0], want[5][0])     # the updates are visible
    with torch.no_grad():
        m.app_plane[0].copy_(w0)
    fs = FrameStream(m, white_bg=True, N_samples=S)
    got = []

    def submit(k):
        o = fs.submit(rays)
        if o is not None:
            got.append((o[0].clone(), o[1].clone()))

    schedule(submit)
    o = fs.flush()
    got.append((o[0].clone(), o[1].clone()))
    assert len(got) == 6
    for k, ((a, b), (c, d)) in enumerate(zip(got, want)):
        assert torch.equal(a, c) and torch.equal(b, d), f"frame {k}"
    assert m.scene_settled()
