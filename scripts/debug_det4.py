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
mode = sys.argv[1]
keep = []
ref = None
nbad = 0
for rep in range(int(os.environ.get("REPS", "12"))):
    if mode == "fresh":
        keep.append(m._scratch); m._scratch = None          # never reuse scratch memory
    if mode == "sync":
        torch.cuda.synchronize()
    rgb, depth = m.render_rays(rays, white_bg=True, N_samples=S, eps_T=0.0)
    if mode == "sync":
        torch.cuda.synchronize()
    if ref is None: ref = rgb.clone()
    else:
        bad = int((rgb != ref).any(1).sum()); nbad += bad > 0
        print(mode, "rep", rep, "rays differing", bad)
print(mode, "runs differing from run 0:", nbad, "of", int(os.environ.get("REPS", "12")) - 1)
