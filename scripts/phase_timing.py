import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import bench
if os.environ.get('TVR_LIB_PATH') is None: print('note: set TVR_LIB_PATH to a -DTVR_TIMING=1 build')
m, arrs, A = bench.build_model(torch.device("cuda"))
m.mlp_arith = os.environ.get("TVR_ARITH", "f32")             # TVR_ARITH=f16act / f16: the opt-in arithmetics
print("mlp_arith", m.mlp_arith)
rays = bench.frames(A)[0].cuda()
stats = torch.zeros(16, dtype=torch.int64, device="cuda")
for _ in range(2): m.render_rays(rays, N_samples=512)
stats.zero_()
m.render_rays(rays, N_samples=512, stats=stats)
torch.cuda.synchronize()
st = stats.cpu().numpy().astype(float)
tot = st[8:16].sum()
print("entries", st[2], "tiles", st[2] / 32)
names = ("queue-entry fetch for the next tile", "gather", "basis", "L1 (+PE)", "L2", "-", "wait for the matrix token", "token hand-over, layer 3 + store")      # stamps of a -DTVR_TIMING=1 build (scripts/build_variant.sh timing -DTVR_TIMING=1)
for n, v in zip(names, st[8:16]):
    print(f"{n:26s} {v / (st[2] / 32):9.0f} cycles/tile  {100 * v / tot:5.1f} %")
print("sum cycles/tile", tot / (st[2] / 32))
