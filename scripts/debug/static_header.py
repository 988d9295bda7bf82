import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
import bench
from jittor_myc_nerfs_amd import _lib as L, OctreeRender_trilinear_fast
m, arrs, A = bench.build_model(torch.device("cuda"))
rays = bench.frames(A)[0].cuda()
nS = 1039
def hdr():
    b = m._train_buf
    lay = L.ScratchLayout(); L.check(L.lib().tvr_scratch_describe(b["key"][0], b["key"][1], C.byref(lay)), "d")
    torch.cuda.synchronize()
    return b["scratch"][lay.counter:lay.counter + 16].view(torch.int32).tolist(), b["cap"]
for it in range(3):
    idx = torch.randint(0, rays.shape[0], (4096,), device="cuda")
    for p in m.parameters(): p.grad = None
    rgb, _, _, _, _ = OctreeRender_trilinear_fast(rays[idx], m, chunk=4096, N_samples=nS, white_bg=True, is_train=True)
    print("after forward ", hdr())
    rgb.sum().backward()
    print("after backward", hdr())
