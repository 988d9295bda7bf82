#!/usr/bin/env python3
"""GPU box: what the scene-maintenance calls of the training loop cost at the reference's sizes (train.py:196-199, 279-296):
filtering_rays over the training set's rays (bbox only, then through the alpha mask), updateAlphaMask at 128^3 / 200^3 / the grid's own size, shrink,
upsample_volume_grid.   python3 scripts/maintenance_timing.py [n_images]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device("cuda")
m, arrs, A = bench.build_model(dev)
fr = bench.frames(A)
rays = torch.cat([fr[i % len(fr)] for i in range(n_img)])                  # host tensor, [n_img * 640000, 6], as train.py holds allrays
rgbs = torch.zeros((rays.shape[0], 3))


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print("%-46s %8.1f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
    return out


timed("updateAlphaMask(128^3)", lambda: m.updateAlphaMask((128, 128, 128)))
timed("updateAlphaMask(200^3)", lambda: m.updateAlphaMask((200, 200, 200)))
timed("updateAlphaMask(gridSize 300^3)", lambda: m.updateAlphaMask(tuple(m.gridSize)))
r1 = timed("filtering_rays(bbox_only) %d M rays" % (rays.shape[0] // 1000000), lambda: m.filtering_rays(rays, rgbs, bbox_only=True))
print("   kept %d of %d" % (r1[0].shape[0], rays.shape[0]))
r2 = timed("filtering_rays(alpha mask) %d M rays" % (rays.shape[0] // 1000000), lambda: m.filtering_rays(rays, rgbs))
print("   kept %d of %d" % (r2[0].shape[0], rays.shape[0]))
timed("upsample_volume_grid(300 -> 300)", lambda: m.upsample_volume_grid(tuple(m.gridSize)))
