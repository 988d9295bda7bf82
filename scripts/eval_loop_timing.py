#!/usr/bin/env python3
"""GPU box: wall time per frame of `evaluation()` (renderer.py:107-141 equivalent: render a test pose, PSNR / SSIM against its picture, write two PNGs) over the bench scene's
8 poses at 800 x 800, with and without the one-frame lookahead of evaluation._FrameFetch (TVR_EVAL_LOOKAHEAD=0 / 1, read at import: run once per setting)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast, evaluation


class DS:
    pass


def main():
    dev = torch.device("cuda")
    m, arrs, A = bench.build_model(dev)
    fr = bench.frames(A)
    W, H = A["img_wh"]
    ds = DS()
    ds.near_far, ds.img_wh = A["near_far"], (W, H)
    ds.all_rays = torch.stack([f for f in fr] * 2)                  # 16 frames
    with torch.no_grad():
        ds.all_rgbs = torch.stack([m.render_rays(f.to(dev), white_bg=True, N_samples=512)[0].cpu().clamp(0, 1) for f in fr] * 2)
    for extra in (False, True):
        with tempfile.TemporaryDirectory() as d:
            evaluation(ds, m, None, OctreeRender_trilinear_fast, savePath=d, N_vis=-1, N_samples=512, white_bg=True, compute_extra_metrics=extra, device="cuda")   # warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ps = evaluation(ds, m, None, OctreeRender_trilinear_fast, savePath=d, N_vis=-1, N_samples=512, white_bg=True, compute_extra_metrics=extra, device="cuda")
            dt = (time.perf_counter() - t0) / len(ps)
        print(f"lookahead {os.environ.get('TVR_EVAL_LOOKAHEAD', '1')}  ssim {int(extra)}  {dt * 1e3:7.1f} ms per frame of evaluation() (16 frames, PNGs written)   PSNR[0] {ps[0]:.2f}", flush=True)


if __name__ == "__main__":
    main()
