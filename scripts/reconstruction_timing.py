#!/usr/bin/env python3
"""GPU box: the reference's WHOLE reconstruction run (tensorf-myc/train.py:113-371 through reconstruct.reconstruction — the real loop, not a step in isolation)
at the schedule of configs/Coffee.txt (TensorVMSplit) or configs/Scar.txt (REFTensoRF): 30 000 iterations of 4096 rays, 128^3 -> 300^3 with five upsamplings, two alpha-mask updates (shrink, ray filtering),
TV + L1 regularisers, fused Adam with per-group lr decay.  The training set is synthetic: scene A rendered by a teacher model from `--views`
poses at --img x --img (100 x 800 x 800 = 64 M rays, all resident in HBM).

    python3 scripts/reconstruction_timing.py [--iters 30000] [--views 100] [--img 800] [--model TensorVMSplit]
(Scene A is a random-factor cloud, not a glossy object: a REFTensoRF run on it measures time, its held-out PSNR says nothing about the model.)
Prints the wall time of every phase (teacher render, both ray filters, every 1000 iterations with their it/s) and the PSNR on a held-out pose."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from jittor_myc_nerfs_amd import rays as R
from jittor_myc_nerfs_amd.reconstruct import config_parser, reconstruction

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30000)
ap.add_argument("--views", type=int, default=100)
ap.add_argument("--img", type=int, default=800)
ap.add_argument("--model", default="TensorVMSplit", choices=["TensorVMSplit", "REFTensoRF"])
ap.add_argument("--teacher", default="TensorVMSplit", choices=["TensorVMSplit", "REFTensoRF"], help="the model that renders the synthetic training set")
ap.add_argument("--pe", type=int, default=2, help="view_pe = fea_pe of the model that is trained (6: opt.py's and TensorBase.__init__'s own default, 390 MLP inputs — the fused step since round 6)")
a = ap.parse_args()
dev = torch.device("cuda")
t_all = time.perf_counter()
teacher, arrs, A = bench.build_model(dev, a.teacher)
poses = []                                                   # four rings of elevations; the last pose is held out
per = (a.views + 1 + 3) // 4
for el in (30.0, 55.0, 5.0, -20.0):
    poses += R.sphere_poses(per, A["cam_radius"], elevation_deg=el)
poses = poses[:a.views + 1]
t0 = time.perf_counter()
with torch.no_grad():
    frames = [R.frame_rays(M, a.img, a.img, A["camera_angle_x"]).to(dev) for M in poses]
    rgbs = [teacher.render_rays(f, white_bg=True, N_samples=A["N_samples"])[0] for f in frames]
torch.cuda.synchronize()
print("teacher: %d views of %d x %d rendered in %.2f s" % (len(poses), a.img, a.img, time.perf_counter() - t0), flush=True)
del teacher


class Set:
    pass


train, val = Set(), Set()
train.all_rays, train.all_rgbs = torch.cat(frames[:-1]), torch.cat(rgbs[:-1])
val.all_rays, val.all_rgbs = frames[-1].view(a.img, a.img, 6)[None], rgbs[-1].view(a.img, a.img, 3)[None]
for d in (train, val):
    d.scene_bbox = torch.tensor(np.asarray(A["aabb"], dtype=np.float32)).reshape(2, 3)
    d.white_bg, d.near_far = True, list(A["near_far"])
del frames, rgbs

tmp = tempfile.mkdtemp(prefix="recon_")
cmd = ["--dataset_name", "blender", "--expname", "timing", "--basedir", tmp, "--n_iters", str(a.iters), "--batch_size", "4096",
       "--N_voxel_init", str(128 ** 3), "--N_voxel_final", str(300 ** 3), "--N_vis", "0", "--vis_every", "100000", "--progress_refresh_rate", "10",
       "--model_name", a.model, "--shadingMode", "MLP_Fea", "--fea2denseAct", "softplus", "--view_pe", str(a.pe), "--fea_pe", str(a.pe),
       "--white_bkgd"]
if a.model == "REFTensoRF":        # configs/Scar.txt: no L1 term, TV 2, the normal penalty 0.5, a 400 000-iteration decay (the first --iters of them are run)
    cmd += ["--TV_weight_density", "2.0", "--TV_weight_app", "2.0", "--rm_weight_mask_thre", "1e-6", "--normal_vector_penalty_weight", "0.5", "--lr_decay_iters", "400000"]
else:                              # configs/Coffee.txt
    cmd += ["--L1_weight_inital", "4e-5", "--L1_weight_rest", "2e-5", "--TV_weight_density", "0.3", "--TV_weight_app", "0.3", "--rm_weight_mask_thre", "1e-3"]
for v in (16, 16, 16):
    cmd += ["--n_lamb_sigma", str(v)]
for v in (48, 48, 48):
    cmd += ["--n_lamb_sh", str(v)]
scale = a.iters / 30000.0
for v in (2000, 3000, 4000, 5500, 7000):
    cmd += ["--upsamp_list", str(int(v * scale))]
for v in (2000, 4000):
    cmd += ["--update_AlphaMask_list", str(int(v * scale))]
args = config_parser(cmd)

marks = []


def log(msg):
    if msg.startswith("Iteration"):
        it = int(msg.split()[1].rstrip(":"))
        if it % 1000 == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if marks:
                print("  %s | %.1f it/s over the last 1000" % (msg, 1000.0 / (now - marks[-1])), flush=True)
            else:
                print("  " + msg, flush=True)
            marks.append(now)
    else:
        print("  " + msg, flush=True)


t0 = time.perf_counter()
model, folder, _ = reconstruction(args, device="cuda", log=log, train_dataset=train, val_dataset=val)
torch.cuda.synchronize()
t_train = time.perf_counter() - t0
with torch.no_grad():
    out, _ = model(val.all_rays.view(-1, 6), is_train=False, white_bg=True, N_samples=-1)[:2]
psnr = float(-10 * torch.log10(torch.mean((out - val.all_rgbs.view(-1, 3)) ** 2)))
print("reconstruction(%s): %d iterations in %.1f s (%.1f it/s over the whole run, everything included), held-out PSNR %.2f dB, final grid %s"
      % (a.model, a.iters, t_train, a.iters / t_train, psnr, model.gridSize.tolist()), flush=True)
# the TRAINED scene in the opt-in arithmetics of the appearance network (include/tvr.h TVR_ARITH_*): the held-out view's PSNR per mode, the picture's distance
# from the default mode's, and the magnitudes its features really have (the reduced modes' error is relative to them)
with torch.no_grad():
    rays_v, gt = val.all_rays.view(-1, 6), val.all_rgbs.view(-1, 3)
    rep = model.fp16_range_report() if hasattr(model, "fp16_range_report") else {}
    ref = None
    for mode in ("f32", "f16act", "f16"):
        model.mlp_arith = mode
        pic = model(rays_v, is_train=False, white_bg=True, N_samples=-1)[0]
        if ref is None:
            ref = pic
        print("  mlp_arith %-7s held-out PSNR %.3f dB, RGB L-inf vs the default mode %.2e (mean %.2e)" %
              (mode, float(-10 * torch.log10(torch.mean((pic - gt) ** 2))), float((pic - ref).abs().max()), float((pic - ref).abs().mean())), flush=True)
    model.mlp_arith = "f32"
    print("  interval bounds of the trained scene (fp16_range_report): " + ", ".join("%s %.3g" % (k, v) for k, v in rep.items() if k != "proven"), flush=True)
print("whole script: %.1f s" % (time.perf_counter() - t_all))
