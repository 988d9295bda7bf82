import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
import bench
from jittor_myc_nerfs_amd import _lib as L, OctreeRender_trilinear_fast, TVLoss
REG, OPT = int(sys.argv[1]), sys.argv[2]
m, arrs, A = bench.build_model(torch.device("cuda"))
allrays = bench.frames(A)[0].cuda()
target = torch.rand((allrays.shape[0], 3), device="cuda")
nS = 1039
opt = None if OPT == "none" else torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), capturable=True, fused=(OPT == "fused"), foreach=(OPT == "foreach"))
tv = TVLoss()
def hdr():
    b = m._train_buf
    lay = L.ScratchLayout(); L.check(L.lib().tvr_scratch_describe(b["key"][0], b["key"][1], C.byref(lay)), "d")
    torch.cuda.synchronize()
    return b["scratch"][lay.counter:lay.counter + 32].view(torch.int32).tolist()
def step():
    idx = torch.randint(0, allrays.shape[0], (4096,), device="cuda")
    if opt is not None: opt.zero_grad()
    else:
        for p in m.parameters(): p.grad = None
    rgb_map, _, _, _, _ = OctreeRender_trilinear_fast(allrays[idx], m, chunk=4096, N_samples=nS, white_bg=True, is_train=True)
    loss = torch.mean((rgb_map - target[idx]) ** 2)
    if REG: loss = loss + 1e-4 * m.vector_comp_diffs() + 8e-5 * m.density_L1() + 0.1 * m.TV_loss_density(tv) + 0.01 * m.TV_loss_app(tv)
    loss.backward()
    if opt is not None: opt.step()
    return loss
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("REG", REG, "OPT", OPT, "after warm-up", hdr(), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = step()
print("after capture", hdr(), flush=True)
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    g.replay()
    h = hdr()
    if i < 3 or h[2] != 0 or i % 5 == 0: print("after replay", i, h, float(l.detach()), flush=True)
