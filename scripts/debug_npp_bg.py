import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import TINY, make_model, GOLDEN
from jittor_myc_nerfs_amd import synthetic
g = dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz"))); n = dict(np.load(os.path.join(GOLDEN, "tiny_npp.npz")))
arrs = {k[6:]: v for k, v in g.items() if k.startswith("scene.")}; arrs.update({k[6:]: v for k, v in n.items() if k.startswith("scene.")})
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
cw = torch.tensor(np.random.default_rng(21).standard_normal((64, 3)).astype(np.float32))
res = {}
for dev in ("cpu", "cuda"):
    m = make_model(arrs, hyper, device=dev)
    rays = torch.tensor(n["rays"], device=dev)
    bg = m._background(rays[:, :3], rays[:, 3:6], torch.tensor(n["rand_bg"], device=dev))
    (bg * cw.to(dev)).sum().backward()
    res[dev] = {k: p.grad.cpu().numpy() for k, p in m.bg_net.named_parameters()}, bg.detach().cpu().numpy()
print("bg value diff", np.abs(res["cpu"][1] - res["cuda"][1]).max())
for k in res["cpu"][0]:
    a, b = res["cpu"][0][k], res["cuda"][0][k]
    print(f"{k:32s} rel-max-err {np.abs(a - b).max() / max(np.abs(a).max(), 1e-9):.2e}")
