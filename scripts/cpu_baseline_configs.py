#!/usr/bin/env python3
"""the restated CPU path (oracle (a): the reference's op sequence on torch-CPU fp32, chunk 1024) on BASELINE config 1
in full (128^3, 64x64 rays x 192 samples) and on a 64x64 centre crop of config 2 (300^3, 512 samples); median of 5; plus the scalar-C oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from jittor_myc_nerfs_amd import rays as R, synthetic
from oracle import c_oracle as CO, tensorf_oracle as TO
cores = bench.usable_cores()
torch.set_num_threads(cores)
def run(tag, S, grid, aabb, near_far, step_ratio, cam_radius, W, crop):
    arrs = synthetic.make_scene_arrays(grid, aabb)
    hyper = dict(synthetic.HYPER, near_far=near_far, step_ratio=step_ratio)
    sc = TO.scene_from_arrays(arrs, **hyper)
    rays = R.frame_rays(R.sphere_poses(8, cam_radius)[0], W, W, synthetic.SCENE_A["camera_angle_x"])
    if crop:
        rays = rays.view(W, W, 6)[W // 2 - 32:W // 2 + 32, W // 2 - 32:W // 2 + 32].reshape(-1, 6).contiguous()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); TO.OctreeRender_trilinear_fast(rays, sc, chunk=1024, N_samples=S, white_bg=True); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)
    t0 = time.perf_counter(); co.render(rays.numpy(), S, white_bg=True, nthreads=cores); tc = time.perf_counter() - t0
    t0 = time.perf_counter(); co.render(rays.numpy(), S, white_bg=True, nthreads=1); t1 = time.perf_counter() - t0
    n = rays.shape[0]
    print(f"{tag}: {n} rays x {S} samples, {cores} cores: restated torch-CPU path {t:.2f} s = {n * S / t:.3e} ray-samples/s, {n / t:.3e} rays/s; "
          f"scalar-C oracle {cores} threads {n * S / tc:.3e} ray-samples/s, 1 thread {n * S / t1:.3e}")
B, A = synthetic.SCENE_B, synthetic.SCENE_A
run("config 1 (128^3, 64x64x192)", B["N_samples"], B["gridSize"], B["aabb"], B["near_far"], B["step_ratio"], B["cam_radius"], 64, False)
run("config 2 crop (300^3, 64x64 centre crop x 512)", A["N_samples"], A["gridSize"], A["aabb"], A["near_far"], A["step_ratio"], A["cam_radius"], 800, True)
