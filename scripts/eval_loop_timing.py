"""Per-frame cost of the evaluation loop (renderer.py:29-91 shape: render, PSNR, SSIM, two PNGs per frame) at 800x800 on the synthetic scene."""
import sys, os, time, tempfile, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast, evaluation
m, arrs, A = bench.build_model(torch.device("cuda"))
fr = bench.frames(A)[:4]
with torch.no_grad():
    gts = [m.render_rays(f.cuda(), N_samples=512)[0].cpu().reshape(800, 800, 3) for f in fr]
ds = types.SimpleNamespace(all_rays=torch.stack([f.cpu() for f in fr], 0), all_rgbs=torch.stack(gts, 0), near_far=A["near_far"], img_wh=(800, 800))
args = types.SimpleNamespace(expname="t")
for extra in (True, False):
    with tempfile.TemporaryDirectory() as d:
        evaluation(ds, m, args, OctreeRender_trilinear_fast, d, N_vis=-1, N_samples=512, white_bg=True, compute_extra_metrics=extra)   # warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ps = evaluation(ds, m, args, OctreeRender_trilinear_fast, d, N_vis=-1, N_samples=512, white_bg=True, compute_extra_metrics=extra)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(fr)
    print(f"evaluation loop, compute_extra_metrics={extra}: {dt * 1e3:.0f} ms / frame (PSNR {sum(ps) / len(ps):.1f} dB)")
