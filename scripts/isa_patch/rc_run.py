"""Root-cause experiment on the round-1 commit 2db8f31 (the build that corrupted): render the config-1 rays repeatedly with a given
libtvr variant and count pixels that differ between runs / chunkings (the per-ray result is order-independent when the shade kernel is clean)."""
import os, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jittor_myc_nerfs_amd import _lib
name = sys.argv[1]
_lib.LIB_PATH = os.path.join(ROOT, "jittor-myc-nerfs_amd", "lib", f"libtvr_{name}.so")
from jittor_myc_nerfs_amd import synthetic
from conftest import make_model
from jittor_myc_nerfs_amd import rays as R
A = synthetic.SCENE_A
arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"])
m = make_model(arrs, dict(synthetic.HYPER, near_far=A["near_far"], step_ratio=A["step_ratio"]))
rays = R.frame_rays(R.sphere_poses(8, A["cam_radius"])[0], 800, 800, A["camera_angle_x"]).cuda()
import time
EPS = None if len(sys.argv) < 3 else float(sys.argv[2])
t0 = time.time()
n = rays.shape[0]
def render(r):
    out = m.render_rays(r, white_bg=True, N_samples=512, eps_T=EPS); torch.cuda.synchronize(); return out[0]
full = render(rays)
tot_rr = tot_ch = tot_pm = 0; mx = 0.0
for rep in range(3):
    a = render(rays); d = (a != full).any(1); tot_rr += int(d.sum()); mx = max(mx, float((a - full).abs().max()))
    ch = 100000 + 4096 * rep
    c = torch.cat([render(rays[i:i + ch]) for i in range(0, n, ch)]); d = (c != full).any(1); tot_ch += int(d.sum()); mx = max(mx, float((c - full).abs().max()))
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(rep))
    p = render(rays[perm]); d = (p != full[perm]).any(1); tot_pm += int(d.sum()); mx = max(mx, float((p - full[perm]).abs().max()))
print(f"variant {name:8s}: differing pixels of {n} rays x 3 reps: re-render {tot_rr}, chunked {tot_ch}, permuted {tot_pm}; max |diff| {mx:.2e}; {time.time()-t0:.1f}s")
