import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import GOLDEN, TINY, make_model
from jittor_myc_nerfs_amd import synthetic
mode = sys.argv[1]
dump = dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz")))
arrs = {k[len("scene."):]: v for k, v in dump.items() if k.startswith("scene.")}
refd = dict(np.load(os.path.join(GOLDEN, "tiny_ref.npz")))
arrs.update({k[len("scene."):]: v for k, v in refd.items() if k.startswith("scene.")})
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
rays = torch.tensor(np.concatenate([dump["rays"]] * 8), device="cuda")
target = torch.rand((rays.shape[0], 3), device="cuda")
m = make_model(arrs, hyper)
print("model", type(m).__name__, flush=True)
def fwd():
    rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    return rgb
def fwd_bwd(pen):
    for p in m.parameters():
        if p.grad is not None: p.grad.zero_()
    rgb = fwd()
    loss = torch.mean((rgb - target) ** 2)
    if pen: loss = loss + 0.5 * m.penalty
    loss.backward()
    return loss
fn = {"fwd": lambda: fwd().sum(), "fb": lambda: fwd_bwd(False), "fbp": lambda: fwd_bwd(True)}[mode]
for p in m.parameters(): p.grad = torch.zeros_like(p)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn(); fn()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("warm ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", float(out), flush=True)
