"""Experiment: does a 2-D tiled ray order (instead of 800-pixel raster rows) speed the kernels up?  (L2 reuse across image rows)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np, ctypes as C
import bench
from jittor_myc_nerfs_amd import _lib as L
m, arrs, A = bench.build_model(torch.device("cuda"))
rays = bench.frames(A)[0].cuda()
W = H = 800
def tile_perm(T):
    idx = torch.arange(W * H).view(H, W)
    return idx.view(H // T, T, W // T, T).permute(0, 2, 1, 3).reshape(-1).cuda()
prof = C.c_void_p(); L.check(L.lib().tvr_profile_create(64, C.byref(prof)), "p")
for name, perm in (("raster", None), ("tile8", tile_perm(8)), ("tile16", tile_perm(16)), ("tile32", tile_perm(32)), ("tile4", tile_perm(4))):
    r = rays if perm is None else rays[perm].contiguous()
    for _ in range(3): m.render_rays(r, N_samples=512)
    L.lib().tvr_profile_reset(prof)
    for _ in range(10): m.render_rays(r, N_samples=512, profile=prof)
    torch.cuda.synchronize()
    ms = (C.c_float * 3)(); n = L.lib().tvr_profile_read(prof, C.byref(ms))
    print(f"{name:8s} march {ms[0]/n:.2f}  shade {ms[1]/n:.2f}  composite {ms[2]/n:.3f}  total {(ms[0]+ms[1]+ms[2])/n:.2f} ms")
