import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import GOLDEN, TINY, make_model
from jittor_myc_nerfs_amd import synthetic
dump = dict(np.load(os.path.join(GOLDEN, "tiny_dump.npz")))
arrs = {k[len("scene."):]: v for k, v in dump.items() if k.startswith("scene.")}
hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
rays = np.concatenate([dump["rays"]] * 8).copy()
rays[:, :3] += 0.01 * np.random.default_rng(3).standard_normal((rays.shape[0], 3)).astype(np.float32)
rays = torch.tensor(rays, device="cuda")
target = torch.rand((rays.shape[0], 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
m = make_model(arrs, hyper)
names = [n for n, _ in m.named_parameters()]
ps = [p for _, p in m.named_parameters()]
def run():
    for p in ps: p.grad = None
    rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    loss = torch.mean((rgb - target) ** 2)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), [None if p.grad is None else p.grad.clone() for p in ps]
res = [run() for _ in range(4)]
for k in range(1, 4):
    print("call", k, "loss equal", res[k][0] == res[0][0])
    for n, a, b in zip(names, res[0][1], res[k][1]):
        if a is None: continue
        d = float((a - b).abs().max())
        if d != 0.0: print("   ", n, "max abs diff", d, "of", float(a.abs().max()))
import ctypes as C
from jittor_myc_nerfs_amd import _lib as L
b = m._train_buf
lay = L.ScratchLayout(); L.check(L.lib().tvr_scratch_describe(b["key"][0], b["key"][1], C.byref(lay)), "d")
print("header", b["scratch"][lay.counter:lay.counter+16].view(torch.int32).tolist(), "cap", b["cap"])

# ---- which intermediate differs between calls?  (Python replica of tvr_api.hip work_layout)
def work_layout(n, S, cap):
    off = 0; L = {}
    def take(name, floats):
        nonlocal off
        L[name] = (off, floats); off = (off + floats * 4 + 255) // 256 * 256
    for name, fl in [("h", cap*144), ("feats32", cap*32), ("h1", cap*128), ("h2", cap*128), ("rgb", cap*3), ("g8", cap*8), ("rgb_s", cap*3), ("pre", n*3),
                     ("grgb", cap*3), ("gin0", cap), ("grad_w", n*S), ("grad_acc", n), ("d_out4", cap*4), ("dh2", cap*128), ("dh1", cap*128), ("dfeats32", cap*32),
                     ("dg8", cap*8), ("dh", cap*144), ("X", cap*151), ("tmp", 32*144+128)]:
        take(name, fl)
    return L
n, S, cap = b["key"]
WL = work_layout(n, S, cap)
M = 7310
def snap():
    w = m._train_buf["work"]
    out = {}
    for k, (o, fl) in WL.items():
        t = w[o:o + fl * 4].view(torch.float32)
        rows = {"h":144,"feats32":32,"h1":128,"h2":128,"rgb":3,"g8":8,"rgb_s":3,"grgb":3,"gin0":1,"d_out4":4,"dh2":128,"dh1":128,"dfeats32":32,"dg8":8,"dh":144,"X":151}.get(k)
        out[k] = (t[:M * rows] if rows else t).clone()
    return out
run(); s1 = snap(); run(); s2 = snap()
for k in WL:
    d = float((s1[k] - s2[k]).abs().max()) if s1[k].numel() else 0.0
    print(f"{k:10s} max abs diff between two calls {d:.3e}   (max |x| {float(s1[k].abs().max()) if s1[k].numel() else 0:.3e})")
