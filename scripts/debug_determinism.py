import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jittor_myc_nerfs_amd import synthetic, OctreeRender_trilinear_fast
from conftest import make_model
g = dict(np.load(os.path.join(ROOT, "tests/golden/config1.npz")))
B = synthetic.SCENE_B
arrs = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
hyper = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
m = make_model(arrs, hyper)
rays = torch.tensor(g["rays"], device="cuda")
outs = []
for rep in range(8):
    rgb, depth, d = m.render_rays(rays, white_bg=True, N_samples=192, dense=True)
    outs.append((rgb.clone(), d["rgb"].clone(), d["weight"].clone()))
for rep in range(1, 8):
    print("run", rep, "rgb_map equal", torch.equal(outs[0][0], outs[rep][0]), "dense rgb equal", torch.equal(outs[0][1], outs[rep][1]),
          "max diff", float((outs[0][1] - outs[rep][1]).abs().max()), "n diff", int((outs[0][1] != outs[rep][1]).sum()))
rgbc, _, depthc, _, _ = OctreeRender_trilinear_fast(rays, m, chunk=1000, N_samples=192, white_bg=True)
print("chunked vs whole rgb_map equal", torch.equal(rgbc, outs[0][0]), float((rgbc - outs[0][0]).abs().max()), int((rgbc != outs[0][0]).any(1).sum()))
bad = (rgbc != outs[0][0]).any(1).nonzero().flatten()[:10]
print("bad rays", bad.tolist())
# per-sample comparison for chunked: render chunk containing a bad ray densely
if bad.numel():
    r = int(bad[0]); c0 = (r // 1000) * 1000
    rgb_k, depth_k, dk = m.render_rays(rays[c0:c0 + 1000], white_bg=True, N_samples=192, dense=True)
    a = dk["rgb"][r - c0]; b = outs[0][1][r]
    idx = (a != b).any(1).nonzero().flatten()
    print("ray", r, "samples differing", idx.tolist()[:20], "of app", int((outs[0][2][r] > 1e-4).sum()))
    for j in idx[:5].tolist():
        print(j, a[j].tolist(), b[j].tolist(), float(dk["weight"][r - c0, j]), float(outs[0][2][r, j]))
# where do the differing samples sit?
for rep in range(1, 8):
    dd = (outs[0][1] != outs[rep][1]).any(2)
    if dd.any():
        idx = dd.nonzero()
        print("rep", rep, "n", idx.shape[0], "rays/samples:", [(int(a), int(b)) for a, b in idx[:40]])
        # rank of each differing sample within its ray's app list
        for r in idx[:, 0].unique()[:3].tolist():
            app = (outs[0][2][r] > 1e-4).nonzero().flatten().tolist()
            js = idx[idx[:, 0] == r][:, 1].tolist()
            print("  ray", r, "napp", len(app), "ranks", [app.index(j) for j in js])
        break
