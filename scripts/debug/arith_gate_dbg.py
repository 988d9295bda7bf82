import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, TINY, make_model
from jittor_myc_nerfs_amd import synthetic, _lib as L
dump = dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz")))
arrs = {k[len("scene."):]: v for k, v in dump.items() if k.startswith("scene.")}
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
rays = torch.tensor(dump["rays"], device="cuda")
S = TINY["N_samples"]
for first in ("appfeature", "render"):
    m = make_model(arrs, hyper)
    if first == "appfeature":
        m.compute_appfeature(torch.tensor(dump["app_xyz_norm"], device="cuda"))
    ref, _ = m.render_rays(rays, white_bg=True, N_samples=S)
    ref = ref.clone()
    for mode in ("f16act", "f16"):
        m.mlp_arith = mode
        a, _ = m.render_rays(rays, white_bg=True, N_samples=S)
        print(first, mode, "in effect", m.arith_in_effect, "gate maxdiff", m.arith_max_diff, "diff vs f32", float((a - ref).abs().max()), "lib mode", L.lib().tvr_scene_get_arith(m._scene), flush=True)
        m.mlp_arith_tol = 1.0
        m._arith_refused_sig = None
        b, _ = m.render_rays(rays, white_bg=True, N_samples=S)
        print("   forced:", m.arith_in_effect, m.arith_max_diff, float((b - ref).abs().max()), flush=True)
        c, _ = m.render_rays(rays, white_bg=True, N_samples=S)
        print("   again :", float((c - ref).abs().max()), flush=True)
        m.mlp_arith_tol = 2.5e-4
