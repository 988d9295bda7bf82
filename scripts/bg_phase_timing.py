#!/usr/bin/env python3
"""Phase timing of NerfPlusPlus's background-network kernel on a -DTVR_BG_TIMING=1 build (scripts/build_variant.sh bgtime -DTVR_BG_TIMING=1;
TVR_LIB_PATH=.../libtvr_bgtime.so python scripts/bg_phase_timing.py): per-wave s_memtime sums the kernel adds into its work buffer."""
import os
import sys

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
    n, N = 65536, m.BG_SAMPLES
    g = torch.Generator(device="cuda").manual_seed(1)
    u = torch.randn(n, N, 3, device="cuda", generator=g)
    pts = torch.cat([u / u.norm(dim=-1, keepdim=True), torch.rand(n, N, 1, device="cuda", generator=g)], -1)
    v = torch.randn(n, 3, device="cuda", generator=g)
    v = v / v.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        for _ in range(3):
            m._mlpnet(pts, v)
        torch.cuda.synchronize()
        wk = m._bg_work()
        wk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m._mlpnet(pts, v)
        e1.record()
        torch.cuda.synchronize()
    t = wk.view(torch.int64)[8:16].cpu().numpy().astype(float)
    tiles = n * N / 32
    names = ["tile total", "boundary: wait for the DMA (vmcnt(0))", "boundary: barrier", "boundary: DMA issue", "tile start -> point loaded", "base layers (boundaries incl.)",
             "heads + stores issued (boundary incl.)", "ticket barrier"]
    print(f"kernel {e0.elapsed_time(e1):.3f} ms (timing build); cycles per 32-sample tile and wave:")
    for nm, x in zip(names, t):
        print(f"  {nm:<44} {x / tiles:9.0f}")
    if t[0] == 0:
        print("  (all zero: not a -DTVR_BG_TIMING=1 library)")


if __name__ == "__main__":
    main()
