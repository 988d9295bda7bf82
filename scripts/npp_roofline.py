#!/usr/bin/env python3
"""GPU box: roofline of NerfPlusPlus's background-network kernel (bg_mlp_kernel, csrc/tvr_bg.hip) — VERDICT r4 item 6.

The kernel is Embedder + MLPNet.forward (nerfplusplus.py:7-56, 66-140) for 512 background samples per ray, everything register-resident on the matrix
cores (fp16 hi/lo split, three products per fp32 product).  Bound: MFMA.  Algorithmic work per sample = 2 x the multiply-adds of the network AS THE KERNEL
EVALUATES IT (base_remap folded into the first rgb layer: 128 -> 64 instead of 128 -> 256 -> 64; the reference's own form is printed beside it).
Prints ONE JSON line: {"kernel", "bound", "achieved", "peak", "unit", "frac", ...}.  Under rocprofv3 --kernel-trace --stats the average duration of
bg_mlp_kernel must agree with `ms_per_launch` (profiles/r05_npp_kernel_stats.csv)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jittor_myc_nerfs_amd import NerfPlusPlus, synthetic          # noqa: E402


def main():
    A, H = synthetic.SCENE_A, synthetic.HYPER
    arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], npp=6.0)
    m = NerfPlusPlus(arrs["aabb"], A["gridSize"], "cuda", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27, near_far=A["near_far"],
                     shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=H["density_shift"], distance_scale=H["distance_scale"],
                     rayMarch_weight_thres=H["rayMarch_weight_thres"], pos_pe=6, view_pe=2, fea_pe=2, featureC=128, step_ratio=A["step_ratio"],
                     fea2denseAct=H["fea2denseAct"])
    m.load_arrays(arrs)
    net = m.bg_net
    ic, iv = net.input_ch, net.input_ch_viewdirs
    macs_ref = sum(l[0].in_features * l[0].out_features for l in net.base_layers) + net.sigma_layers[0].in_features + net.base_remap_layers[0].in_features * 256 \
        + (256 + iv) * 64 + 64 * 3
    macs_kernel = sum(l[0].in_features * l[0].out_features for l in net.base_layers) + 128 + (128 + iv) * 64 + 64 * 3      # base_remap folded (tvr_bg.hip)
    n, N = 65536, m.BG_SAMPLES
    g = torch.Generator(device="cuda").manual_seed(1)
    u = torch.randn(n, N, 3, device="cuda", generator=g)
    pts = torch.cat([u / u.norm(dim=-1, keepdim=True), torch.rand(n, N, 1, device="cuda", generator=g)], -1)
    v = torch.randn(n, 3, device="cuda", generator=g)
    v = v / v.norm(dim=-1, keepdim=True)
    out = {}
    with torch.no_grad():
        for mode in ("f32",):
            m.mlp_arith = mode
            m._mlpnet(pts, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                m._mlpnet(pts, v)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            samples = n * N
            ach = 2.0 * macs_kernel * samples / (ms * 1e-3) / 1e12
            out = {"kernel": "bg_mlp_kernel<arith 0>", "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                   "frac_vs_fp32class_ceiling": ach / (2500.0 / 3.0), "traffic": None,
                   "algorithmic_flop_per_sample": 2 * macs_kernel, "reference_form_flop_per_sample": 2 * macs_ref, "samples_per_launch": samples,
                   "ms_per_launch": ms, "G_samples_per_s": samples / (ms * 1e-3) / 1e9,
                   "frame_800x800_samples": 640000 * N, "ms_per_800x800_frame_of_this_kernel": ms * 640000 / n,
                   "network": {"D": len(net.base_layers), "W": 128, "input_ch": ic, "input_ch_viewdirs": iv, "bg_freq": m.bg_freq},
                   "note": "2 x multiply-adds per sample of the network as evaluated (base_remap folded into the first rgb layer) x samples / HIP-event time of the launch; "
                           "peak = dense f16 MFMA 2.5 PFLOP/s; three fp16 products per fp32 product put the ceiling of this arithmetic at peak / 3"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
