import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jittor_myc_nerfs_amd import synthetic
from oracle import tensorf_oracle as TO
from conftest import make_model
g = dict(np.load(os.path.join(ROOT, "tests/golden/config1.npz")))
B = synthetic.SCENE_B
arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
hyper = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
m = make_model(arrs, hyper)
sc = TO.scene_from_arrays(arrs, **hyper)
rays = torch.tensor(g["rays"], device="cuda")
S = 192
_, _, d0 = m.render_rays(rays, white_bg=True, N_samples=S, dense=True, eps_T=0.0)
z = d0["z"]; xn = m.normalize_coord(rays[:, None, :3] + rays[:, None, 3:6] * z[..., None])
app = d0["weight"] > 1e-4
feat = m.compute_appfeature(xn[app])
ref = torch.zeros(rays.shape[0], S, 3, device="cuda"); ref[app] = feat[:, :3]
for rep in range(6):
    _, _, d = m.render_rays(rays, white_bg=True, N_samples=S, dense=True, eps_T=0.0)
    err = (d["rgb"] - ref)
    bad = (err.abs().amax(2) > 1e-5).nonzero()
    print("rep", rep, "n bad", bad.shape[0], "max", float(err.abs().max()))
    for (r, j) in bad[:3].tolist():
        x = xn[r, j].cpu()[None]
        f, h = TO.compute_appfeature(sc, x, return_h=True)
        bas = sc.basis_mat
        contrib = torch.stack([(bas[:3, 16 * s:16 * s + 16] * h[0, 16 * s:16 * s + 16]).sum(1) for s in range(9)])   # [9,3]
        # half contributions (lane half h: channels 8h..8h+7 of each k-step)
        half = torch.stack([torch.stack([(bas[:3, 16 * s + 8 * hh:16 * s + 8 * hh + 8] * h[0, 16 * s + 8 * hh:16 * s + 8 * hh + 8]).sum(1) for hh in range(2)]) for s in range(9)])
        e = err[r, j].cpu()
        print("  ray", r, "j", j, "err", e.tolist())
        print("   -contrib per k-step (ch0):", [round(float(-c[0]), 5) for c in contrib], " halves ch0:", [[round(float(-half[s, hh, 0]), 5) for hh in range(2)] for s in range(9)])
