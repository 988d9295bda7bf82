#!/usr/bin/env python3
"""GPU box: sha256 of the bench frame's pixels and depths (8 poses of scene A, 800x800x512) as rendered by the library TVR_LIB_PATH selects.
Two libraries whose digests agree render the same frames bit for bit (used by scripts/phase_rule_test.sh; the default arithmetic unless --arith)."""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jittor_myc_nerfs_amd import rays as R, synthetic                    # noqa: E402
from jittor_myc_nerfs_amd.field import TensorVMSplit                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--arith", default="f32")
    a = ap.parse_args()
    A = synthetic.SCENE_A
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"])
    hyper = dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"])
    m = TensorVMSplit(arrs["aabb"], [int(x) for x in arrs["gridSize"]], "cuda", density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                      near_far=hyper["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper["density_shift"],
                      distance_scale=hyper["distance_scale"], rayMarch_weight_thres=hyper["rayMarch_weight_thres"], pos_pe=6, view_pe=2, fea_pe=2,
                      featureC=128, step_ratio=hyper["step_ratio"], fea2denseAct=hyper["fea2denseAct"])
    m.load_arrays(arrs)
    m.mlp_arith = a.arith
    poses = R.sphere_poses(8, A["cam_radius"])
    for rep in range(a.repeats):
        hsh = hashlib.sha256()
        for p in range(a.poses):
            rays = R.frame_rays(poses[p], 800, 800, A["camera_angle_x"]).cuda()
            rgb, depth = m.render_rays(rays, white_bg=True, N_samples=A["N_samples"])
            hsh.update(rgb.cpu().numpy().tobytes())
            hsh.update(depth.cpu().numpy().tobytes())
        print("lib %s  repeat %d  sha256 %s" % (os.path.basename(os.environ.get("TVR_LIB_PATH", "libtvr.so")), rep, hsh.hexdigest()), flush=True)


if __name__ == "__main__":
    main()
