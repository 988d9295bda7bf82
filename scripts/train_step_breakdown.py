"""Where the training step's time goes (CUDA events around its sections; same setting as train_step_timing.py)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from jittor_myc_nerfs_amd import OctreeRender_trilinear_fast, TVLoss
m, arrs, A = bench.build_model(torch.device("cuda"))
with torch.no_grad():
    for p in m.parameters(): p.mul_(0.9)
fr = bench.frames(A)
allrays = torch.cat(fr[:2]).cuda()
allrgbs = torch.rand((allrays.shape[0], 3), device="cuda")
nS = int(np.linalg.norm(A["gridSize"]) / 0.5)
opt = torch.optim.Adam(m.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), fused=True)
tv = TVLoss()
g = torch.Generator(device="cuda").manual_seed(0)
names = ["render fwd", "mse", "regularisers fwd", "backward", "optimizer"]
acc = {n: 0.0 for n in names}
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(25):
    idx = torch.randint(0, allrays.shape[0], (4096,), device="cuda", generator=g)
    opt.zero_grad()
    e0 = ev()
    rgb_map, _, _, _, _ = OctreeRender_trilinear_fast(allrays[idx], m, chunk=4096, N_samples=nS, white_bg=True, is_train=True)
    e1 = ev()
    loss = torch.mean((rgb_map - allrgbs[idx]) ** 2)
    e2 = ev()
    total = loss + 1e-4 * m.vector_comp_diffs() + 8e-5 * m.density_L1() + 0.1 * m.TV_loss_density(tv) + 0.01 * m.TV_loss_app(tv)
    e3 = ev()
    total.backward()
    e4 = ev()
    opt.step()
    e5 = ev()
    torch.cuda.synchronize()
    if it >= 5:
        for n, (a, b) in zip(names, ((e0, e1), (e1, e2), (e2, e3), (e3, e4), (e4, e5))): acc[n] += a.elapsed_time(b)
for n in names: print(f"{n:18s} {acc[n] / 20:.2f} ms")
print(f"sum {sum(acc.values()) / 20:.2f} ms")
