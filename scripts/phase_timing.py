import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
import bench
m, arrs, A = bench.build_model(torch.device("cuda"))
rays = bench.frames(A)[0].cuda()
stats = torch.zeros(16, dtype=torch.int64, device="cuda")
for _ in range(2): m.render_rays(rays, N_samples=512)
stats.zero_()
m.render_rays(rays, N_samples=512, stats=stats)
torch.cuda.synchronize()
st = stats.cpu().numpy().astype(float)
tot = st[8:15].sum()
print("entries", st[2], "tiles", st[2] / 32)
names = ("basis", "gather begin + PE", "L1 (+gather 0..4)", "L2 (+gather 5..8)", "L3", "epilogue", "-") if os.environ.get("TVR_PIPE_NAMES") else ("gather", "basis", "PE", "token wait", "L1+L2", "L3", "epilogue")
for n, v in zip(names, st[8:15]):
    print(f"{n:12s} {v / (st[2] / 32):9.0f} cycles/tile  {100 * v / tot:5.1f} %")
print("sum cycles/tile", tot / (st[2] / 32))
