"""The callers of the render path in the reference's own shape (tensorf-myc/train.py, tensorf-myc/opt.py): option parser for the shipped
`configs/*.txt` files, `reconstruction` (train.py:113-371) and `render_test` (train.py:62-110), on the HIP field and renderer.

    python -m jittor_myc_nerfs_amd.reconstruct --config configs/Scar.txt [--datadir ...]

What differs from the reference, on purpose:
  * the optimizer is torch.optim.Adam (fused); the per-group lr decay, the lr reset after upsampling, the regulariser schedule and the
    checkpoint contents (`kwargs`, `state_dict`, packed alpha mask, `lr`, `global_step`) are the reference's;
  * no TensorBoard writer, no `export_mesh` (skimage / plyfile are not available here), no `ndc_ray` datasets (the reference ships only
    the Blender loader); `set_nerfplusplus` is called for NerfPlusPlus only (train.py:172 calls it unconditionally and fails for the others);
  * progress is a plain print every `progress_refresh_rate` iterations.
Host-side plumbing only: every pixel comes from the HIP kernels through `OctreeRender_trilinear_fast`.
"""
from __future__ import annotations

import argparse
import ast
import datetime
import os
import sys
from typing import Dict, List, Optional

import numpy as np
import torch

from .evaluation import BlenderRays, evaluation, evaluation_path
from .field import TensorVMSplit, load_checkpoint
from .losses import TVLoss
from .render import N_to_reso, OctreeRender_trilinear_fast, cal_n_samples
from .variants import NerfPlusPlus, REFTensoRF

MODELS = {"TensorVMSplit": TensorVMSplit, "REFTensoRF": REFTensoRF, "NerfPlusPlus": NerfPlusPlus}


def parse_config_file(path: str) -> Dict[str, object]:
    """The `key = value` files the reference feeds to configargparse (configs/Scar.txt): `#` comments, blank lines, `[a, b, c]` lists."""
    out: Dict[str, object] = {}
    with open(path) as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line or "=" not in line:
                continue
            key, val = [t.strip() for t in line.split("=", 1)]
            if val.startswith("["):
                out[key] = list(ast.literal_eval(val))
            else:
                out[key] = val
    return out


def config_parser(cmd: Optional[List[str]] = None) -> argparse.Namespace:
    """tensorf-myc/opt.py:4-155 — same option names, types and defaults; `--config` values are defaults that the command line overrides."""
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, default=None)
    for name, typ, default in (("bg_freq", int, 4), ("bg_view_freq", int, 2), ("bg_D", int, 4), ("radii", float, 20.0),
                               ("normal_vector_penalty_weight", float, 0.0), ("near", float, None), ("far", float, None), ("gc_every", int, 20),
                               ("expname", str, None), ("basedir", str, "./log"), ("add_timestamp", int, 0), ("datadir", str, "./data/llff/fern"),
                               ("progress_refresh_rate", int, 10), ("downsample_train", float, 1.0), ("downsample_test", float, 1.0),
                               ("batch_size", int, 4096), ("n_iters", int, 30000), ("lr_init", float, 0.02), ("lr_basis", float, 1e-3),
                               ("lr_decay_iters", int, -1), ("lr_decay_target_ratio", float, 0.1), ("lr_upsample_reset", int, 1),
                               ("L1_weight_inital", float, 0.0), ("L1_weight_rest", float, 0.0), ("Ortho_weight", float, 0.0),
                               ("TV_weight_density", float, 0.0), ("TV_weight_app", float, 0.0), ("data_dim_color", int, 27),
                               ("rm_weight_mask_thre", float, 1e-4), ("alpha_mask_thre", float, 1e-4), ("distance_scale", float, 25.0),
                               ("density_shift", float, -10.0), ("shadingMode", str, "MLP_PE"), ("pos_pe", int, 6), ("view_pe", int, 6),
                               ("fea_pe", int, 6), ("featureC", int, 128), ("ckpt", str, None), ("render_only", int, 0), ("render_test", int, 0),
                               ("render_train", int, 0), ("render_path", int, 0), ("export_mesh", int, 0), ("perturb", float, 1.0),
                               ("accumulate_decay", float, 0.998), ("fea2denseAct", str, "softplus"), ("ndc_ray", int, 0), ("nSamples", float, 1e6),
                               ("step_ratio", float, 0.5), ("N_voxel_init", int, 100 ** 3), ("N_voxel_final", int, 300 ** 3), ("idx_view", int, 0),
                               ("N_vis", int, 5), ("vis_every", int, 10000)):
        p.add_argument("--" + name, type=typ, default=default)
    p.add_argument("--model_name", type=str, default="TensorVMSplit", choices=["TensorVMSplit", "TensorCP", "NerfPlusPlus", "REFTensoRF"])
    # (not a reference option) arithmetic of the appearance network in the evaluation renders: field.py `mlp_arith`, include/tvr.h TVR_ARITH_*; training is fp32-class always
    p.add_argument("--mlp_arith", type=str, default="f32", choices=["f32", "f16act", "f16"])
    p.add_argument("--dataset_name", type=str, default="blender", choices=["blender", "llff", "nsvf", "dtu", "tankstemple", "own_data"])
    for name in ("with_depth", "lindisp", "white_bkgd"):
        p.add_argument("--" + name, action="store_true")
    for name, typ in (("bbox", float), ("n_lamb_sigma", int), ("n_lamb_sh", int), ("upsamp_list", int), ("update_AlphaMask_list", int)):
        p.add_argument("--" + name, type=typ, action="append")
    argv = sys.argv[1:] if cmd is None else list(cmd)
    pre, _ = p.parse_known_args(argv)
    if pre.config:
        cfg = parse_config_file(pre.config)
        defaults = {}
        for k, v in cfg.items():
            act = next((a for a in p._actions if a.dest == k), None)
            if act is None:
                raise SystemExit(f"{pre.config}: unknown option {k!r}")
            if isinstance(act, argparse._StoreTrueAction):
                defaults[k] = str(v).lower() in ("1", "true", "yes")
            elif isinstance(v, list):
                defaults[k] = [act.type(x) for x in v]
            else:
                defaults[k] = act.type(v) if act.type is not None else v
        p.set_defaults(**defaults)
    return p.parse_args(argv)


class SimpleSampler:
    """train.py:21-36: a fresh random permutation per epoch, consecutive batches of it.  `device`: where the permutation lives — with the training set on the
    GPU a batch's indices are then a device slice (a host permutation costs one pageable host-to-device copy per step, which waits for the stream)."""

    def __init__(self, total, batch, device=None):
        self.total, self.batch, self.curr, self.ids, self.device = total, batch, total, None, device

    def nextids(self):
        self.curr += self.batch
        if self.curr + self.batch > self.total:
            self.ids = torch.randperm(self.total, device=self.device)
            self.curr = 0
        return self.ids[self.curr:self.curr + self.batch]


def _bbox(args):
    return None if not args.bbox else [args.bbox[:3], args.bbox[3:6]]


def _build_from_ckpt(args, ckpt, device):
    kwargs = dict(ckpt["kwargs"])
    kwargs.update({"device": device})
    bg = {k: kwargs.pop(k) for k in ("bg_D", "bg_freq", "radii", "bg_view_freq") if k in kwargs}       # train.py:45-54
    tensorf = MODELS[args.model_name](**kwargs)
    if bg:
        tensorf.set_nerfplusplus(bg["bg_freq"], bg["bg_view_freq"], bg["bg_D"], bg["radii"])
    tensorf.load(ckpt)
    tensorf.mlp_arith = getattr(args, "mlp_arith", "f32")
    return tensorf, kwargs


@torch.no_grad()
def render_test(args, device="cuda"):
    """train.py:62-110."""
    if args.dataset_name != "blender":
        raise NotImplementedError("only the Blender loader exists in the reference (dataLoader/__init__.py)")
    test_dataset = BlenderRays(args.datadir, split="test", downsample=args.downsample_train, is_stack=True, bbox=_bbox(args), near=args.near,
                               far=args.far, white_bg=args.white_bkgd)
    if not args.ckpt or not os.path.exists(args.ckpt):
        print("the ckpt path does not exists!!")
        return None
    tensorf, _ = _build_from_ckpt(args, load_checkpoint(args.ckpt), device)
    logfolder = os.path.dirname(args.ckpt)
    out = {}
    if args.render_test:
        out["test"] = evaluation(test_dataset, tensorf, args, OctreeRender_trilinear_fast, f"{logfolder}/imgs_test_all/", N_vis=-1, N_samples=-1,
                                 white_bg=test_dataset.white_bg, ndc_ray=args.ndc_ray, device=device)
        print(f"======> {args.expname} test all psnr: {np.mean(out['test'])} <========================")
    if args.render_path:
        out["path"] = evaluation_path(test_dataset, tensorf, [p.numpy() for p in test_dataset.poses], OctreeRender_trilinear_fast,
                                      f"{logfolder}/imgs_path_all/", N_vis=-1, N_samples=-1, white_bg=test_dataset.white_bg, ndc_ray=args.ndc_ray,
                                      device=device)
    return out


FAULT_POLL_EVERY = 16        # iterations between two reads of the training-fault accumulator (field.check_training_faults)


def reconstruction(args, device="cuda", log=print, train_dataset=None, val_dataset=None):
    """train.py:113-371.  Returns (tensorf, logfolder, PSNRs_test of the last visualisation or final test).
    `train_dataset` / `val_dataset`: ready-made datasets (objects with all_rays, all_rgbs, scene_bbox, white_bg, near_far as BlenderRays has them) instead of the
    Blender folder under args.datadir — synthetic training sets (scripts/reconstruction_timing.py)."""
    if args.dataset_name != "blender":
        raise NotImplementedError("only the Blender loader exists in the reference (dataLoader/__init__.py)")
    if args.model_name not in MODELS:
        raise NotImplementedError(f"model_name {args.model_name!r} is outside the accelerated path (TensorVMSplit, REFTensoRF, NerfPlusPlus)")
    if args.ndc_ray:
        raise NotImplementedError("ndc_ray datasets are not part of the reference's loaders")
    if train_dataset is None:
        train_dataset = BlenderRays(args.datadir, split="train", downsample=args.downsample_train, is_stack=False, bbox=_bbox(args), near=args.near,
                                    far=args.far, white_bg=args.white_bkgd)
    if val_dataset is None:
        val_split = "val" if os.path.exists(os.path.join(args.datadir, "transforms_val.json")) else "train"
        val_dataset = BlenderRays(args.datadir, split=val_split, downsample=args.downsample_train, is_stack=True, bbox=_bbox(args), near=args.near,
                                  far=args.far, white_bg=args.white_bkgd)
    white_bg, near_far = train_dataset.white_bg, train_dataset.near_far
    upsamp_list = list(args.upsamp_list or [])
    update_AlphaMask_list = list(args.update_AlphaMask_list or [])
    logfolder = f"{args.basedir}/{args.expname}" + (datetime.datetime.now().strftime("-%Y%m%d-%H%M%S") if args.add_timestamp else "")
    for sub in ("", "/imgs_vis", "/imgs_rgba", "/rgba"):
        os.makedirs(logfolder + sub, exist_ok=True)

    aabb = train_dataset.scene_bbox
    reso_cur = N_to_reso(args.N_voxel_init, aabb)
    nSamples = int(min(args.nSamples, cal_n_samples(reso_cur, args.step_ratio)))
    global_step = 0
    ckpt = None
    if args.ckpt is not None:
        ckpt = load_checkpoint(args.ckpt)
        if "global_step" in ckpt:
            global_step = int(ckpt["global_step"]) + 1
        tensorf, kwargs = _build_from_ckpt(args, ckpt, device)
        nSamples = int(min(args.nSamples, cal_n_samples(kwargs["gridSize"], args.step_ratio)))
        reso_cur = [int(g) for g in kwargs["gridSize"]]
    else:
        tensorf = MODELS[args.model_name](aabb, reso_cur, device, density_n_comp=args.n_lamb_sigma, appearance_n_comp=args.n_lamb_sh,
                                          app_dim=args.data_dim_color, near_far=near_far, shadingMode=args.shadingMode,
                                          alphaMask_thres=args.alpha_mask_thre, density_shift=args.density_shift,
                                          distance_scale=args.distance_scale, pos_pe=args.pos_pe, view_pe=args.view_pe, fea_pe=args.fea_pe,
                                          featureC=args.featureC, step_ratio=args.step_ratio, fea2denseAct=args.fea2denseAct)
        if isinstance(tensorf, NerfPlusPlus):
            tensorf.set_nerfplusplus(bg_freq=args.bg_freq, bg_view_freq=args.bg_view_freq, bg_D=args.bg_D, radii=args.radii)
        tensorf.mlp_arith = getattr(args, "mlp_arith", "f32")            # (the evaluation renders of the run; the training steps compute fp32-class whatever it says)

    if args.lr_decay_iters > 0:
        lr_factor = args.lr_decay_target_ratio ** (1 / args.lr_decay_iters)
    else:
        args.lr_decay_iters = args.n_iters
        lr_factor = args.lr_decay_target_ratio ** (1 / args.n_iters)

    def make_optimizer(lr_xyz, lr_net):
        return torch.optim.Adam(tensorf.get_optparam_groups(lr_xyz, lr_net), betas=(0.9, 0.99), fused=(torch.device(device).type == "cuda"))

    optimizer = make_optimizer(args.lr_init, args.lr_basis)
    if ckpt is not None and "lr" in ckpt:
        for pg, lr in zip(optimizer.param_groups, ckpt["lr"]):
            pg["lr"] = lr
    N_voxel_list = [int(v) for v in torch.round(torch.exp(torch.linspace(np.log(args.N_voxel_init), np.log(args.N_voxel_final),
                                                                          len(upsamp_list) + 1))).long().tolist()][1:]
    PSNRs, PSNRs_test = [], [0]
    allrays, allrgbs = train_dataset.all_rays, train_dataset.all_rgbs
    allrays, allrgbs = allrays.to(device), allrgbs.to(device)              # 288 GB of HBM: the whole training set lives on the device
    allrays, allrgbs = tensorf.filtering_rays(allrays, allrgbs, bbox_only=True)
    sampler = SimpleSampler(allrays.shape[0], args.batch_size, device=allrays.device)
    Ortho_w, L1_w = args.Ortho_weight, args.L1_weight_inital
    TV_d, TV_a = args.TV_weight_density, args.TV_weight_app
    tvreg = TVLoss()
    pen_w = args.normal_vector_penalty_weight
    reso_mask = reso_cur

    fused_adam = torch.device(device).type == "cuda"
    loss_hist = []
    for iteration in range(global_step, args.n_iters):
        optimizer.zero_grad()
        idx = sampler.nextids().to(device)
        rays_train, rgb_train = allrays[idx], allrgbs[idx]
        rgb_map, _, depth_map, _, _ = OctreeRender_trilinear_fast(rays_train, tensorf, chunk=args.batch_size, N_samples=nSamples, white_bg=white_bg,
                                                                  ndc_ray=False, device=device, is_train=True)
        loss = torch.mean((rgb_map - rgb_train) ** 2)
        total_loss = loss
        if Ortho_w > 0:
            total_loss = total_loss + Ortho_w * tensorf.vector_comp_diffs()
        if L1_w > 0:
            total_loss = total_loss + L1_w * tensorf.density_L1()
        if TV_d > 0:
            TV_d *= lr_factor
            total_loss = total_loss + tensorf.TV_loss_density(tvreg) * TV_d
        if TV_a > 0:
            TV_a *= lr_factor
            total_loss = total_loss + tensorf.TV_loss_app(tvreg) * TV_a
        if pen_w > 0 and hasattr(tensorf, "penalty"):                                              # train.py:253-257
            total_loss = total_loss + pen_w * tensorf.penalty
            tensorf.penalty = torch.zeros((), device=device)
        total_loss.backward()
        # No host read per step (train.py:262 takes `loss.item()` here; on this GPU that wait is 1.5 ms of a 3.9 ms step): the fused Adam kernel itself skips an
        # update whose step raised a fault flag (workspace overflow / fp16-range saturation inside the fused step, field.training_fault_flag), and the
        # losses stay on the device until the next log line.
        if fused_adam:
            optimizer.found_inf = tensorf.training_fault_flag()
            optimizer.step()
        else:
            fault = tensorf.check_training_faults()
            if fault is not None:
                log(f"Iteration {iteration:05d}: training step dropped ({fault})")
                optimizer.zero_grad(set_to_none=True)
            else:
                optimizer.step()
        loss_hist.append(loss.detach())
        for pg in optimizer.param_groups:
            pg["lr"] = pg["lr"] * lr_factor
        # the fault accumulator is polled on its own short cadence, not at the log rate: every step between a first fault and the adjustment of capacity / scale
        # is dropped on the device while the learning-rate decay keeps running (one host read per FAULT_POLL_EVERY steps: <= 0.1 ms per step)
        if fused_adam and (iteration % FAULT_POLL_EVERY == 0 or iteration % args.progress_refresh_rate == 0):
            fault = tensorf.check_training_faults()
            if fault is not None:                      # the flagged steps were skipped on the device; capacity / scale are adjusted now
                log(f"Iteration {iteration:05d}: training step(s) dropped since the last poll ({fault}); samples per ray "
                    f"{tensorf.train_app_samples_per_ray}, scale {tensorf.grad_scale_target:g}")
        if iteration % args.progress_refresh_rate == 0:
            hist = torch.stack(loss_hist)
            PSNRs = (-10.0 * torch.log10(hist)).tolist()
            log(f"Iteration {iteration:05d}: train_psnr = {float(np.mean(PSNRs)):.2f} test_psnr = {float(np.mean(PSNRs_test)):.2f} mse = {float(hist[-1]):.6f}")
            PSNRs, loss_hist = [], []
        if iteration % args.vis_every == args.vis_every - 1 and args.N_vis != 0 and len(val_dataset.all_rgbs):
            with torch.no_grad():
                PSNRs_test = evaluation(val_dataset, tensorf, args, OctreeRender_trilinear_fast, f"{logfolder}/imgs_vis/", N_vis=args.N_vis,
                                        prtx=f"{iteration:06d}_", N_samples=nSamples, white_bg=white_bg, ndc_ray=False, compute_extra_metrics=False,
                                        device=device)
        if iteration in update_AlphaMask_list:                                                     # train.py:292-313
            if reso_cur[0] * reso_cur[1] * reso_cur[2] < 256 ** 3:
                reso_mask = reso_cur
            new_aabb = tensorf.updateAlphaMask(tuple(reso_mask))
            if iteration == update_AlphaMask_list[0]:
                tensorf.shrink(new_aabb)
                L1_w = args.L1_weight_rest
            if len(update_AlphaMask_list) > 1 and iteration == update_AlphaMask_list[1]:
                allrays, allrgbs = tensorf.filtering_rays(allrays, allrgbs)
                sampler = SimpleSampler(allrgbs.shape[0], args.batch_size, device=allrays.device)
        if iteration in upsamp_list:                                                               # train.py:316-330
            reso_cur = N_to_reso(N_voxel_list.pop(0), tensorf.aabb)
            nSamples = int(min(args.nSamples, cal_n_samples(reso_cur, args.step_ratio)))
            tensorf.upsample_volume_grid(reso_cur)
            lr_scale = 1 if args.lr_upsample_reset else args.lr_decay_target_ratio ** (iteration / args.n_iters)
            optimizer = make_optimizer(args.lr_init * lr_scale, args.lr_basis * lr_scale)
        if iteration % (5 * args.vis_every) == 0 and iteration > 0:                                # train.py:337-345
            tensorf.save(f"{logfolder}/{args.expname}{iteration}.th",
                         {"lr": [pg["lr"] for pg in optimizer.param_groups], "global_step": iteration})
    tensorf.save(f"{logfolder}/{args.expname}.th", {"lr": [pg["lr"] for pg in optimizer.param_groups], "global_step": args.n_iters})
    if args.render_test:
        test_dataset = BlenderRays(args.datadir, split="test", downsample=args.downsample_train, is_stack=True, bbox=_bbox(args), near=args.near,
                                   far=args.far, white_bg=args.white_bkgd)
        if len(test_dataset.all_rgbs):
            with torch.no_grad():
                PSNRs_test = evaluation(test_dataset, tensorf, args, OctreeRender_trilinear_fast, f"{logfolder}/imgs_test_all/", N_vis=-1,
                                        N_samples=-1, white_bg=white_bg, ndc_ray=False, device=device)
            log(f"======> {args.expname} test all psnr: {np.mean(PSNRs_test)} <========================")
    return tensorf, logfolder, PSNRs_test


def main(cmd: Optional[List[str]] = None):
    torch.manual_seed(20211202)                                                                    # train.py:396-397
    np.random.seed(20211202)
    args = config_parser(cmd)
    if args.export_mesh:
        raise NotImplementedError("export_mesh needs skimage.measure.marching_cubes and plyfile (utils.py:146), neither available here")
    if args.render_only and (args.render_test or args.render_path):
        return render_test(args)
    return reconstruction(args)


if __name__ == "__main__":
    main()
