import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jittor_myc_nerfs_amd import synthetic
from conftest import make_model
B = synthetic.SCENE_B
arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
g = torch.Generator(device="cpu").manual_seed(0)
n = 200000
xyz = (torch.rand(n, 3, generator=g) * 1.2 - 0.6).cuda()
dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).cuda()
f0 = m.compute_appfeature(xyz)
for rep in range(4):
    f1 = m.compute_appfeature(xyz)
    perm = torch.randperm(n, device="cuda")
    f2 = torch.empty_like(f0); f2[perm] = m.compute_appfeature(xyz[perm])
    print("appfeature rep", rep, "same-order equal", torch.equal(f0, f1), int((f0 != f1).any(1).sum()),
          "permuted equal", torch.equal(f0, f2), int((f0 != f2).any(1).sum()), float((f0 - f2).abs().max()))
r0 = m.renderModule(xyz, dirs, f0)
for rep in range(4):
    r1 = m.renderModule(xyz, dirs, f0)
    perm = torch.randperm(n, device="cuda")
    r2 = torch.empty_like(r0); r2[perm] = m.renderModule(xyz[perm], dirs[perm], f0[perm])
    bad = (r0 != r1).any(1).nonzero().flatten()
    print("mlp rep", rep, "same-order equal", torch.equal(r0, r1), bad.numel(), bad[:20].tolist(),
          "permuted equal", torch.equal(r0, r2), int((r0 != r2).any(1).sum()), float((r0 - r2).abs().max()))
