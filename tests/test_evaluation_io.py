"""CPU: Blender-format dataset reader + evaluation loop plumbing with a stand-in renderer; the GPU variant of the
same loop is in test_gpu_parity.py."""
import json
import os

import numpy as np
import torch

from conftest import ROOT


def _write_scene(tmp_path, n=3, wh=16):
    from jittor_myc_nerfs_amd import rays as R
    from PIL import Image
    poses = R.sphere_poses(n, 4.0)
    meta = {"camera_angle_x": 0.6911, "frames": []}
    os.makedirs(tmp_path / "test", exist_ok=True)
    rng = np.random.default_rng(0)
    for i, M in enumerate(poses):
        meta["frames"].append({"file_path": f"./test/r_{i}", "transform_matrix": M.tolist()})
        rgba = (rng.random((800, 800, 4)) * 255).astype(np.uint8)
        Image.fromarray(rgba).save(tmp_path / "test" / f"r_{i}.png")
    with open(tmp_path / "transforms_test.json", "w") as f:
        json.dump(meta, f)
    return poses


def test_blender_rays_and_evaluation_loop(tmp_path):
    from jittor_myc_nerfs_amd import BlenderRays, evaluation, evaluation_path, rays as R, rgb_ssim
    poses = _write_scene(tmp_path)
    ds = BlenderRays(str(tmp_path), split="test", downsample=50.0)               # 16x16 images
    assert ds.img_wh == (16, 16) and ds.all_rays.shape == (3, 256, 6) and ds.all_rgbs.shape == (3, 16, 16, 3)
    assert torch.equal(ds.all_rays[1], R.frame_rays(poses[1], 16, 16, 0.6911))
    assert float(ds.all_rgbs.min()) >= 0 and float(ds.all_rgbs.max()) <= 1

    def fake_renderer(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True, is_train=False, device="cpu"):
        rgb = (rays[:, 3:6] * 0.5 + 0.5)
        return rgb, None, rays[:, 5].abs() * 10 + 5, None, None

    class A: expname = "exp"
    out = tmp_path / "out"
    psnrs = evaluation(ds, None, A, fake_renderer, savePath=str(out), N_vis=-1, prtx="t_", white_bg=True, device="cpu")
    assert len(psnrs) == 3 and all(np.isfinite(psnrs))
    from PIL import Image
    img = np.asarray(Image.open(out / "exp_r_0.png"))
    assert img.shape == (16, 16, 3) and np.asarray(Image.open(out / "rgbd" / "t_000.png")).shape == (16, 32, 3)
    mean = np.loadtxt(out / "t_mean.txt")
    assert abs(mean[0] - np.mean(psnrs)) < 1e-9 and 0 <= mean[1] <= 1
    frames = evaluation_path(ds, None, [p.numpy() for p in ds.poses], fake_renderer, savePath=str(out / "path"), device="cpu")
    assert len(frames) == 3 and np.array_equal(frames[0], img)                     # same poses -> same images as evaluation()
    a = np.random.default_rng(1).random((24, 24, 3))
    assert abs(rgb_ssim(a, a, 1) - 1.0) < 1e-12 and rgb_ssim(a, 1 - a, 1) < 0.5
    # the device version (what `evaluation` calls) is the same computation
    from jittor_myc_nerfs_amd import rgb_ssim_torch
    rng2 = np.random.default_rng(4)
    x, y = rng2.random((37, 41, 3)).astype(np.float32), rng2.random((37, 41, 3)).astype(np.float32)
    y = 0.7 * x + 0.3 * y
    assert abs(rgb_ssim_torch(torch.from_numpy(x), torch.from_numpy(y), 1) - rgb_ssim(x, y, 1)) < 1e-12


def test_reads_the_reference_repos_refined_poses_if_present():
    """data_refine/Easyship is the only Blender-format fixture in the reference (its test file is misnamed transform_test.json)."""
    d = "/root/reference/data_refine/Easyship"
    if not os.path.isdir(d):
        import pytest
        pytest.skip("reference tree not present on this machine")
    from jittor_myc_nerfs_amd import BlenderRays
    ds = BlenderRays(d, split="test", downsample=100.0)
    assert ds.all_rays.shape[1:] == (64, 6) and ds.all_rays.shape[0] == 10 and abs(ds.camera_angle_x - 1.0472) < 1e-3
    assert len(ds.all_rgbs) == 0                                                   # no images ship with the poses
