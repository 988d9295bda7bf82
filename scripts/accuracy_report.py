"""RGB L-inf of the HIP path against the committed golden vectors (tiny per-sample dump, config1 image) — for A/B library builds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import TINY, make_model
from jittor_myc_nerfs_amd import synthetic
g = dict(np.load(os.path.join(ROOT, "tests/golden/tiny_dump.npz")))
arrs = {k[6:]: v for k, v in g.items() if k.startswith("scene.")}
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
m = make_model(arrs, hyper)
rgb, depth, d = m.render_rays(torch.tensor(g["rays"], device="cuda"), N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
print("tiny   per-sample rgb Linf %.2e   rgb_map Linf %.2e" % (np.abs(d["rgb"].cpu().numpy() - g["out.rgb"]).max(),
                                                               np.abs(rgb.cpu().numpy() - g["out.rgb_map"]).max()))
c = dict(np.load(os.path.join(ROOT, "tests/golden/config1.npz")))
B = synthetic.SCENE_B
m1 = make_model(synthetic.make_scene_arrays(B["gridSize"], B["aabb"]), dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"]))
rgb1, _ = m1.render_rays(torch.tensor(c["rays"], device="cuda"), N_samples=B["N_samples"], eps_T=0.0)
print("config1 rgb_map Linf %.2e (eps_T=0)" % np.abs(rgb1.cpu().numpy() - c["rgb_map"]).max())
rgb1, _ = m1.render_rays(torch.tensor(c["rays"], device="cuda"), N_samples=B["N_samples"])
print("config1 rgb_map Linf %.2e (default eps_T)" % np.abs(rgb1.cpu().numpy() - c["rgb_map"]).max())
r = dict(np.load(os.path.join(ROOT, "tests/golden/tiny_ref.npz")))
a2 = dict(arrs); a2.update({k[6:]: v for k, v in r.items() if k.startswith("scene.")})
m2 = make_model(a2, hyper)
rgb2, _, d2 = m2.render_rays(torch.tensor(r["rays"], device="cuda"), N_samples=TINY["N_samples"], eps_T=0.0, dense=True)
print("REF tiny per-sample rgb Linf %.2e   rgb_map Linf %.2e" % (np.abs(d2["rgb"].cpu().numpy() - r["rgb"]).max(),
                                                                 np.abs(rgb2.cpu().numpy() - r["rgb_map"]).max()))
from jittor_myc_nerfs_amd import _lib
print("library:", _lib.LIB_PATH)
