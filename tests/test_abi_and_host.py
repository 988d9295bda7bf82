"""CPU: the C-ABI library loads and exports every symbol include/tvr.h declares; host-side logic (field
constructor, step size / sample count, ray generation, shard maths) — no compute call needs a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, TINY, make_model


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "tvr.h")).read() + open(os.path.join(ROOT, "include", "tvr_ngp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tvr_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from jittor_myc_nerfs_amd import _lib
    names = _header_symbols()
    assert len(names) >= 17
    l = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(l, n), f"libtvr.so does not export {n}"
    assert set(names) == set(_lib.SYMBOLS), "ctypes table and include/*.h disagree"
    assert _lib.lib().tvr_version() == 141


def test_abi_argument_errors_without_gpu():
    from jittor_myc_nerfs_amd import _lib as L
    lib = L.lib()
    d = L.SceneDesc()
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == 0                 # all-zero desc is invalid
    assert b"grid" in lib.tvr_last_error()
    d.grid[:] = [300, 300, 300]
    d.aabb[:] = [-1.5] * 3 + [1.5] * 3
    d.density_n_comp[:] = [17, 8, 8]                                   # more components than the kernels are built for: refused
    d.app_n_comp[:] = [48] * 3
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == 0 and b"density_n_comp" in lib.tvr_last_error()
    d.density_n_comp[:] = [16] * 3
    d.app_dim, d.featureC, d.view_pe, d.fea_pe, d.step_size = 27, 128, 7, 6, 0.005     # more than six encoding frequencies: refused
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == 0 and b"view_pe" in lib.tvr_last_error()
    d.view_pe, d.fea_pe = 6, 6                                         # opt.py's / TensorBase's default frequencies: the lockstep layer-1 path, + 213 KB of packed weights
    n66 = lib.tvr_scene_packed_bytes(C.byref(d))
    d.view_pe, d.fea_pe = 2, 2
    # (a two-frequency TensorVMSplit scene carries the 16x16x32 render kernel's fragment images instead — 163 360 + 2 048 B, 256-B aligned blocks: round 5)
    assert n66 + 163584 + 2048 >= lib.tvr_scene_packed_bytes(C.byref(d)) + 26 * 8192
    nbytes = lib.tvr_scene_packed_bytes(C.byref(d))
    d.density_n_comp[:], d.app_n_comp[:], d.featureC, d.view_pe = [8, 8, 8], [24, 24, 24], 64, 0      # fewer / narrower: zero-padded into the same layout
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == nbytes
    d.density_n_comp[:], d.app_n_comp[:], d.featureC, d.view_pe = [16] * 3, [48] * 3, 128, 2
    # 3 planes x 301x301 x (16+48) ch + lines + MLP, fp32, 256-B aligned blocks, + the appearance factors once more as fp16 (DESIGN.md "data layout")
    assert 69.5e6 + 26.0e6 < nbytes < 70.5e6 + 26.5e6
    h = C.c_void_p()
    assert lib.tvr_scene_create(C.byref(d), None, 0, C.byref(h)) == -3  # TVR_ERR_SCRATCH
    assert lib.tvr_render(None, None, 0, 0, 0, None, 0.0, None, None, None, 0, None, None, None, None) == -1
    assert lib.tvr_render_scratch_bytes(None, 4096, 512) > 4096 * 512 * 20
    # the fused regularisers: 1..8 tensors, non-NULL pointers, 2..48 line components — refused before any launch
    one = (C.c_void_p * 1)(0x1000)
    cnt = (C.c_int64 * 1)(16)
    assert lib.tvr_l1_mean(one, cnt, 0, 0x1000, 0x1000, 64, None) == -1 and lib.tvr_l1_mean(one, cnt, 9, 0x1000, 0x1000, 64, None) == -1
    assert lib.tvr_l1_mean((C.c_void_p * 1)(None), cnt, 1, 0x1000, 0x1000, 64, None) == -1 and b"tensor 0" in lib.tvr_last_error()
    assert lib.tvr_l1_mean(one, cnt, 1, 0x1000, 0x1000, 0, None) == -3 and lib.tvr_l1_mean_scratch_bytes(cnt, 1) == 4
    assert lib.tvr_l1_mean_backward(one, None, cnt, 1, 0x1000, None) == -1
    nc, ns = (C.c_int32 * 1)(49), (C.c_int32 * 1)(300)
    assert lib.tvr_line_ortho(one, nc, ns, 1, 0x1000, 0x1000, 32, None) == -4 and b"components" in lib.tvr_last_error()
    nc[0] = 16
    assert lib.tvr_line_ortho(one, nc, ns, 1, 0x1000, 0x1000, 8, None) == -3
    assert lib.tvr_line_ortho_backward(one, one, nc, ns, 1, None, None) == -1
    d.variant = 2
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == 0 and b"variant" in lib.tvr_last_error()
    d.variant = 1                                                       # REFTensoRF: the same layout + (round 6) the four heads as fragments of the 16x16x32 render kernel: 10 KB + 256 B
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == nbytes + 10 * 1024 + 256
    d.featureC = 64                                                     # ... at the standard shape only
    assert lib.tvr_scene_packed_bytes(C.byref(d)) == 0 and b"REFTensoRF" in lib.tvr_last_error()
    d.featureC = 128
    assert lib.tvr_app_feature_ref(None, None, 0, None, 0, None, 0, None) == -1
    assert lib.tvr_mlp_render_ref(None, None, None, None, 0, None, 0, None) == -1


def test_abi_refuses_undersized_output_buffers_without_gpu():
    """Every output matrix whose row width is the kernels' own carries a byte count that is checked on the host BEFORE any launch (round 2:
    tvr_app_h_forward's h allocated as [m, sum(app_n_comp)] instead of the kernels' [m,144] was overrun on the device and aborted the process).
    The check precedes every other argument check, so it can be exercised here without a GPU: a buffer one float short is TVR_ERR_SCRATCH (-3)."""
    from jittor_myc_nerfs_amd import _lib as L
    lib, m, SCRATCH = L.lib(), 1000, -3
    dummy = C.c_void_p(4096)                                           # never dereferenced: the size check comes first
    big = 1 << 40

    def refused(rc, what):
        assert rc == SCRATCH, (what, rc, lib.tvr_last_error())
        assert what.encode() in lib.tvr_last_error(), lib.tvr_last_error()

    refused(lib.tvr_app_h_forward(None, dummy, m, dummy, m * 144 * 4 - 4, None), "h_out [m,144]")
    refused(lib.tvr_app_h_forward(None, dummy, m, dummy, m * 96 * 4, None), "h_out [m,144]")          # the round-2 caller: 96 = 3 x 32 components
    refused(lib.tvr_app_h_backward(None, dummy, m, dummy, m * 144 * 4 - 4, dummy, big, None, None), "dh [m,144]")
    refused(lib.tvr_density_feature(None, dummy, m, dummy, m * 4 - 4, None), "out [m]")
    refused(lib.tvr_app_feature(None, dummy, m, dummy, m * 27 * 4 - 4, None), "out [m,27]")
    refused(lib.tvr_mlp_render(None, dummy, dummy, m, dummy, m * 3 * 4 - 4, None), "rgb [m,3]")
    refused(lib.tvr_app_feature_ref(None, dummy, m, dummy, m * 27 * 4 - 4, dummy, big, None), "features [m,27]")
    refused(lib.tvr_app_feature_ref(None, dummy, m, dummy, big, dummy, m * 8 * 4 - 4, None), "extra [m,8]")
    refused(lib.tvr_mlp_render_ref(None, dummy, dummy, dummy, m, dummy, m * 3 * 4 - 4, None), "rgb [m,3]")
    ag, ab, inv = (C.c_int32 * 3)(4, 4, 4), (C.c_float * 6)(), (C.c_float * 3)()
    refused(lib.tvr_alpha_sample(dummy, C.byref(ag), C.byref(ab), C.byref(inv), dummy, m, dummy, m * 4 - 4, None), "out [m]")
    refused(lib.tvr_pe_concat(dummy, dummy, None, m, dummy, m * 150 * 4 - 4, None), "X [m,150]")
    refused(lib.tvr_pe_concat(dummy, dummy, dummy, m, dummy, m * 150 * 4, None), "X [m,151]")          # with dot_product the row is 151 wide
    refused(lib.tvr_pe_concat_backward(dummy, dummy, dummy, m, 0, dummy, m * 27 * 4 - 4, None, None, None), "grad_features [m,27]")
    ok = [m * 3 * 4, m * 32 * 4, m * 128 * 4, m * 128 * 4]
    names = ["rgb [m,3]", "feats32 [m,32]", "h1 [m,128]", "h2 [m,128]"]
    for i, n in enumerate(names):
        sizes = list(ok)
        sizes[i] -= 4
        args = [x for sz in sizes for x in (dummy, sz)]
        refused(lib.tvr_mlp_train_forward(None, dummy, dummy, m, *args, None), n)
    assert lib.tvr_mlp_train_forward(None, dummy, dummy, 8_000_000, *[x for sz in ok for x in (dummy, big)], None) == -1      # m * 576 >= 2^32
    assert b"32-bit row offsets" in lib.tvr_last_error()
    okb = [m * 4 * 4, m * 128 * 4, m * 128 * 4, m * 32 * 4, m * 144 * 4]
    namesb = ["d_out4 [m,4]", "dh2 [m,128]", "dh1 [m,128]", "dfeats32 [m,32]", "dh [m,144]"]
    for i, n in enumerate(namesb):
        sizes = list(okb)
        sizes[i] -= 4
        args = [x for sz in sizes for x in (dummy, sz)]
        refused(lib.tvr_mlp_train_backward(*([dummy] * 9), m, dummy, *args, None, dummy, big, None), n)
    # with every size right, the next check (no scene / NULL arguments) answers: nothing was launched
    assert lib.tvr_app_h_forward(None, dummy, m, dummy, m * 144 * 4, None) == -1


def test_ref_field_host_logic():
    """REFTensoRF's host surface (models/REFTensoRF.py:64-106) without a GPU: parameters, optimizer groups, state dict."""
    from jittor_myc_nerfs_amd import REFTensoRF, MLPRender_Fea_Ref
    m = REFTensoRF(np.asarray(TINY["aabb"], np.float32), TINY["gridSize"], "cpu", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3,
                   app_dim=27, near_far=TINY["near_far"], shadingMode="MLP_Fea", view_pe=2, fea_pe=2, featureC=128, step_ratio=0.5)
    assert isinstance(m.renderModule, MLPRender_Fea_Ref) and m.renderModule.in_mlpC == 151
    assert m.normal_linear.weight.shape == (3, 144) and m.rho_linear.weight.shape == (1, 144)
    assert len(m.get_optparam_groups()) == 10 and float(m.penalty) == 0.0
    assert {"normal_linear.weight", "diffuse_linear.bias", "specular_linear.weight", "rho_linear.bias"} <= set(m.state_dict())
    with pytest.raises(NotImplementedError):
        REFTensoRF(np.asarray(TINY["aabb"], np.float32), TINY["gridSize"], "cpu", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3,
                   shadingMode="SH")
    from jittor_myc_nerfs_amd._lib import TvrError
    with pytest.raises(TvrError):                                       # no CPU fallback
        m.render_rays(torch.zeros(4, 6))


def test_field_host_logic_matches_reference_sizes(tiny_arrays, hyper_tiny):
    from jittor_myc_nerfs_amd import synthetic
    A = synthetic.SCENE_A
    arrs = {"aabb": np.asarray(A["aabb"], np.float32), "gridSize": np.asarray(A["gridSize"])}
    from jittor_myc_nerfs_amd import TensorVMSplit
    m = TensorVMSplit(arrs["aabb"], A["gridSize"], "cpu", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27,
                      near_far=A["near_far"], shadingMode="MLP_Fea", view_pe=2, fea_pe=2, featureC=128, step_ratio=0.5)
    assert m.nSamples == 1036 and abs(float(m.stepSize) - 5.0167e-3) < 1e-6
    assert tuple(m.density_plane[1].shape) == (1, 16, 300, 300) and tuple(m.app_line[0].shape) == (1, 48, 300, 1)
    assert m.renderModule.in_mlpC == 150
    kw = m.get_kwargs()
    assert kw["gridSize"] == [300, 300, 300] and kw["shadingMode"] == "MLP_Fea"
    sd = m.state_dict()
    assert "density_plane.0" in sd and "basis_mat.weight" in sd and "renderModule.mlp.4.bias" in sd
    n_par = sum(p.numel() for p in m.parameters())
    assert abs(n_par * 4 / 1e6 - 69.35) < 0.3                                       # 69.35 MB fp32 (Appendix C)


def test_product_path_fails_loudly_without_gpu(tiny_arrays, hyper_tiny, tiny_dump):
    """No CPU fallback: on a CPU device every compute entry point raises."""
    from jittor_myc_nerfs_amd import _lib as L
    m = make_model(tiny_arrays, hyper_tiny, device="cpu")
    rays = torch.tensor(tiny_dump["rays"])
    with pytest.raises(L.TvrError):
        m(rays, is_train=False, white_bg=True, N_samples=48)
    with pytest.raises(L.TvrError):
        m.compute_densityfeature(torch.zeros(4, 3))
    with pytest.raises(NotImplementedError):
        m(rays, ndc_ray=True)


def test_unsupported_shading_mode_raises():
    from jittor_myc_nerfs_amd import TensorVMSplit
    with pytest.raises(NotImplementedError):
        TensorVMSplit([[-1] * 3, [1] * 3], [8, 8, 8], "cpu", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, shadingMode="SH")


def test_ray_generation_blender_convention():
    from jittor_myc_nerfs_amd import rays as R
    H = W = 8
    M = R.sphere_poses(8, 4.0)[3]
    rays = R.frame_rays(M, H, W, 0.6911).numpy()
    assert rays.shape == (64, 6) and rays.dtype == np.float32
    assert np.allclose(np.linalg.norm(rays[:, 3:], axis=1), 1.0, atol=1e-6)       # blender.py:75 normalises
    assert np.allclose(rays[:, :3], M[:3, 3], atol=1e-6)
    # independent restatement of ray_utils.py:91-101 + blender.py:91 for one pixel
    f = 0.5 * 800 / np.tan(0.5 * 0.6911) * (W / 800)
    i, j = 5, 2
    dcam = np.array([-(i + 0.5 - W / 2) / f, (j + 0.5 - H / 2) / f, -1.0])
    dcam /= np.linalg.norm(dcam)
    c2w = M @ R.BLENDER2OPENCV
    assert np.allclose(rays[j * W + i, 3:], c2w[:3, :3] @ dcam, atol=1e-6)
    # the centre of the image looks at the origin
    centre = rays[[27, 28, 35, 36], 3:].mean(0)
    assert np.dot(centre / np.linalg.norm(centre), -M[:3, 3] / np.linalg.norm(M[:3, 3])) > 0.999


def test_shard_indices_partition():
    from jittor_myc_nerfs_amd import shard_indices, shard_capacity
    for R_, w, t in ((10000, 2, 4096), (640000, 8, 4096), (5, 4, 4), (4096, 3, 1024), (0, 2, 16)):
        allidx = torch.cat([shard_indices(R_, r, w, t) for r in range(w)])
        assert sorted(allidx.tolist()) == list(range(R_))
        assert all(shard_indices(R_, r, w, t).numel() <= shard_capacity(R_, w, t) for r in range(w))


def test_checkpoint_save_load_roundtrip(tmp_path, tiny_arrays, hyper_tiny):
    """TensorBase.save / load (tensorBase.py:253-272): kwargs + state_dict + bit-packed alpha mask survive a round trip and
    rebuild an identical field through the reference's `eval(model_name)(**kwargs); load(ckpt)` sequence (train.py:75-87)."""
    from jittor_myc_nerfs_amd import AlphaGridMask, TensorVMSplit
    m = make_model(tiny_arrays, hyper_tiny, device="cpu")
    vol = (np.random.default_rng(0).random((6, 5, 4)) > 0.5).astype(np.float32)
    m.alphaMask = AlphaGridMask("cpu", tiny_arrays["aabb"], torch.tensor(vol))
    path = str(tmp_path / "tiny.th")
    m.save(path, global_kwargs={"global_step": 7})
    ckpt = torch.load(path, weights_only=False)
    assert ckpt["global_step"] == 7 and tuple(ckpt["alphaMask.shape"]) == (1, 1, 6, 5, 4)
    kwargs = ckpt["kwargs"]
    kwargs.update({"device": "cpu"})
    m2 = TensorVMSplit(**kwargs)
    m2.load(ckpt)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert np.array_equal(m2.alphaMask.alpha_volume.numpy().reshape(6, 5, 4), vol)
    assert m2.nSamples == m.nSamples and float(m2.stepSize) == float(m.stepSize)
    assert m2.get_kwargs()["gridSize"] == m.get_kwargs()["gridSize"]


def test_reads_reference_th_checkpoint_layout(tmp_path, tiny_arrays, hyper_tiny):
    """A `.th` file as the reference writes it (tensorBase.py:253-264 via jt.save: a pickled dict whose jt.Vars became numpy arrays; Jittor's
    state_dict also lists the module's non-parameter Vars) loads through load_checkpoint + the reference's rebuild sequence (train.py:75-87)."""
    import pickle
    from jittor_myc_nerfs_amd import TensorVMSplit, load_checkpoint
    m = make_model(tiny_arrays, hyper_tiny, device="cpu")
    sd = {k: v.numpy().copy() for k, v in m.state_dict().items()}
    sd.update({"aabb": tiny_arrays["aabb"], "units": np.ones(3, np.float32), "stepSize": np.float32(0.1), "alphaMask.alpha_volume": np.zeros((1, 1, 2, 2, 2), np.float32)})
    vol = (np.random.default_rng(1).random((1, 1, 5, 4, 3)) > 0.5)
    kwargs = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in m.get_kwargs().items()}
    ckpt = {"kwargs": kwargs, "state_dict": sd, "alphaMask.shape": vol.shape, "alphaMask.mask": np.packbits(vol.reshape(-1)),
            "alphaMask.aabb": np.asarray(tiny_arrays["aabb"], np.float32)}
    path = str(tmp_path / "Scar.th")
    with open(path, "wb") as f:
        pickle.dump(ckpt, f)
    got = load_checkpoint(path)
    kw = got["kwargs"]
    kw.update({"device": "cpu"})
    m2 = TensorVMSplit(**kw)
    m2.load(got)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert np.array_equal(m2.alphaMask.alpha_volume.numpy().reshape(5, 4, 3) > 0.5, vol.reshape(5, 4, 3))
    # our own (torch.save) files go through the same reader
    p2 = str(tmp_path / "own.th")
    m.save(p2)
    assert set(load_checkpoint(p2)["state_dict"]) == set(m.state_dict())
    # a checkpoint of another shape is refused with a clear message, a missing parameter too
    bad = dict(got, state_dict=dict(sd, **{"basis_mat.weight": np.zeros((27, 100), np.float32)}))
    with pytest.raises(ValueError):
        m2.load(bad)
    with pytest.raises(KeyError):
        m2.load(dict(got, state_dict={k: v for k, v in sd.items() if k != "basis_mat.weight"}))


def test_npp_host_logic_on_cpu(tiny_npp, tiny_npp_arrays, hyper_tiny):
    """NerfPlusPlus's host side (models/nerfplusplus.py:143-269, 280-308) is plain torch: sampling and background agree with the golden
    vectors without a GPU; the foreground render needs the HIP library and says so."""
    from jittor_myc_nerfs_amd import NerfPlusPlus
    from jittor_myc_nerfs_amd._lib import TvrError
    m = make_model(tiny_npp_arrays, hyper_tiny, device="cpu")
    assert isinstance(m, NerfPlusPlus) and m.radii == 6.0 and m.bg_embedder_position.out_dim == 36 and m.bg_embedder_viewdir.out_dim == 15
    rays = torch.tensor(tiny_npp["rays"])
    pts, z, mask = m.sample_ray(rays[:, :3], rays[:, 3:6], N_samples=TINY["N_samples"], t_rand=torch.tensor(tiny_npp["rand_fg"]))
    assert np.array_equal(z.numpy(), tiny_npp["out.z_vals"]) and pts.shape == (64, TINY["N_samples"], 3)
    lam = torch.tensor(tiny_npp["out.bg_lambda"])
    with torch.no_grad():
        bg = m._background(rays[:, :3], rays[:, 3:6], torch.tensor(tiny_npp["rand_bg"]))
    assert np.abs((lam[:, None] * bg).numpy() - tiny_npp["out.bg_rgb_map"]).max() < 2e-5
    assert len(m.get_optparam_groups()) == 7 and {"bg_freq", "bg_view_freq", "bg_D", "radii"} <= set(m.get_kwargs())
    assert any(k.startswith("bg_net.base_layers.3.0") for k in m.state_dict())
    with pytest.raises(TvrError):
        m(rays, N_samples=TINY["N_samples"])                          # no CPU fallback for the foreground
    m2 = NerfPlusPlus(tiny_npp_arrays["aabb"], TINY["gridSize"], "cpu", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, shadingMode="MLP_Fea",
                      view_pe=2, fea_pe=2)
    with pytest.raises(TvrError):
        m2(rays)                                                       # set_nerfplusplus() not called


def test_config_parser_reads_the_reference_config_format(tmp_path):
    """opt.py's options over a configs/Scar.txt-style file (written here: `key = value`, `[..]` lists, comments, booleans), command line on top."""
    from jittor_myc_nerfs_amd.reconstruct import SimpleSampler, config_parser
    cfg = tmp_path / "Scene.txt"
    cfg.write_text("""
dataset_name = blender
datadir = ../data/Scene
expname =  Scene
basedir = ./log

normal_vector_penalty_weight = 0.5
bbox = [-5.0, -5.0, -5.0, 5.0, 5.0, 5.0]
near = 5
far = 40
white_bkgd=True

n_iters = 400000
batch_size = 4096
N_voxel_init = 2097156 # 128**3
N_voxel_final = 27000000 # 300**3
upsamp_list = [2000,3000,4000,5500,7000]
update_AlphaMask_list = [2000,4000]
n_lamb_sigma = [16,16,16]
n_lamb_sh = [48,48,48]
model_name = REFTensoRF
shadingMode = MLP_Fea
fea2denseAct = softplus
view_pe = 2
fea_pe = 2
TV_weight_density = 2.
# L1_weight_inital = 1e-5
rm_weight_mask_thre = 1e-6
""")
    a = config_parser(["--config", str(cfg), "--n_iters", "100", "--render_test", "1"])
    assert a.model_name == "REFTensoRF" and a.shadingMode == "MLP_Fea" and a.white_bkgd is True and a.expname == "Scene"
    assert a.bbox == [-5.0, -5.0, -5.0, 5.0, 5.0, 5.0] and a.near == 5.0 and a.far == 40.0 and a.normal_vector_penalty_weight == 0.5
    assert a.n_lamb_sigma == [16, 16, 16] and a.n_lamb_sh == [48, 48, 48] and a.upsamp_list == [2000, 3000, 4000, 5500, 7000]
    assert a.N_voxel_init == 2097156 and a.N_voxel_final == 27000000 and a.TV_weight_density == 2.0 and a.L1_weight_inital == 0.0
    assert a.n_iters == 100 and a.render_test == 1 and a.batch_size == 4096 and a.rm_weight_mask_thre == 1e-6       # command line wins
    d = config_parser([])
    assert d.model_name == "TensorVMSplit" and d.nSamples == 1e6 and d.N_voxel_final == 300 ** 3 and d.lr_init == 0.02 and d.white_bkgd is False
    s = SimpleSampler(10, 4)
    torch.manual_seed(0)
    ids = torch.cat([s.nextids() for _ in range(4)])
    assert ids.numel() == 16 and set(ids[:8].tolist()) <= set(range(10)) and len(set(ids[:8].tolist())) == 8       # two batches of one permutation


def test_grid_sizing_helpers_match_survey_appendix_c():
    from jittor_myc_nerfs_amd import N_to_reso, cal_n_samples
    assert N_to_reso(2097156, ([-5.0] * 3, [5.0] * 3)) == [128, 128, 128]
    assert N_to_reso(27000000, ([-1.5] * 3, [1.5] * 3)) == [300, 300, 300]
    assert cal_n_samples([128, 128, 128], 0.5) == 443 and cal_n_samples([300, 300, 300], 0.5) == 1039


def test_renderer_merges_inference_chunks_but_not_training_batches():
    """OctreeRender_trilinear_fast: evaluation()'s chunk=1024 must not turn into 625 host-bound calls; training keeps its batch."""
    from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast
    calls = []

    class Fake:
        nSamples = 512
        def __call__(self, rays_chunk, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1):
            calls.append(rays_chunk.shape[0])
            return rays_chunk[:, :3], rays_chunk[:, 0]
    rays = torch.zeros(10000, 6)
    with torch.no_grad():
        rgb, _, depth, _, _ = OctreeRender_trilinear_fast(rays, Fake(), chunk=1024, N_samples=512, device="cpu")
    assert calls == [10000] and rgb.shape == (10000, 3) and depth.shape == (10000,)
    calls.clear()
    OctreeRender_trilinear_fast(rays, Fake(), chunk=4096, N_samples=512, is_train=True, device="cpu")      # grad enabled: training batch
    assert calls == [4096, 4096, 1808]
    calls.clear()
    with torch.no_grad():
        OctreeRender_trilinear_fast(torch.zeros(3_000_000, 6), Fake(), chunk=1024, N_samples=1036, device="cpu")
    assert len(calls) > 1 and max(calls) * 1036 * 40 <= 16 << 30 and sum(calls) == 3_000_000


def test_checkpoint_reader_refuses_code_and_reads_arrays(tmp_path):
    """load_checkpoint: a `.th` file is a pickle of numpy arrays and builtin containers (jt.save); anything that would import code is refused."""
    import pickle
    import numpy as np
    from jittor_myc_nerfs_amd.field import load_checkpoint
    good = {"kwargs": {"aabb": np.zeros((2, 3), np.float32), "gridSize": [3, 4, 5]}, "state_dict": {"w": np.arange(6, dtype=np.float32).reshape(2, 3)},
            "global_step": 7, "lr": np.float32(0.02), "alphaMask.mask": np.packbits(np.ones(8, bool))}
    p = tmp_path / "good.th"
    pickle.dump(good, open(p, "wb"))
    d = load_checkpoint(str(p))
    assert d["global_step"] == 7 and np.array_equal(d["state_dict"]["w"], good["state_dict"]["w"]) and d["alphaMask.mask"].dtype == np.uint8

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    q = tmp_path / "evil.th"
    pickle.dump({"state_dict": Evil()}, open(q, "wb"))
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(str(q))
