import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jittor_myc_nerfs_amd import synthetic
from conftest import make_model
g = dict(np.load(os.path.join(ROOT, "tests/golden/config1.npz")))
B = synthetic.SCENE_B
arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
m = make_model(arrs, dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
rays = torch.tensor(g["rays"], device="cuda")
S = 192
# reference per-sample rgb through the (position-independent, deterministic) API kernels
_, _, d0 = m.render_rays(rays, white_bg=True, N_samples=S, dense=True, eps_T=0.0)
z = d0["z"]
xyz = rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]
xn = m.normalize_coord(xyz)
app = d0["weight"] > 1e-4
feat = m.compute_appfeature(xn[app])
ref = torch.zeros(rays.shape[0], S, 3, device="cuda")
ref[app] = m.renderModule(None, rays[:, None, 3:6].expand(-1, S, -1)[app], feat)
for rep in range(8):
    _, _, d = m.render_rays(rays, white_bg=True, N_samples=S, dense=True, eps_T=0.0)
    err = (d["rgb"] - ref).abs().amax(2)
    bad = (err > 1e-4).nonzero()
    print("rep", rep, "max err vs API ref", float(err.max()), "n bad", bad.shape[0])
    if bad.shape[0]:
        r = int(bad[0, 0]); js = bad[bad[:, 0] == r][:, 1].tolist()
        al = app[r].nonzero().flatten().tolist()
        print("  ray", r, "bad ranks", [al.index(j) for j in js])
        # does the bad rgb equal the reference of some other sample of this or a neighbouring ray?
        j0 = js[0]; v = d["rgb"][r, j0]
        cand = (ref[max(0, r - 70):r + 70] - v).abs().amax(2)
        mn = cand.min(); loc = (cand == mn).nonzero()[0]
        print("  bad value", v.tolist(), "ref", ref[r, j0].tolist(), "closest ref elsewhere: dist", float(mn), "at ray", int(loc[0]) + max(0, r - 70), "sample", int(loc[1]), "vs j0", j0)
