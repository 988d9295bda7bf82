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
n, S = rays.shape[0], 192
def al(x): return (x + 255) // 256 * 256
def snapshot():
    sc = m._scratch
    off = 0
    counter = sc[0:4].view(torch.int32).item(); off = 256
    ray_off = sc[off:off + n * 4].view(torch.int32).clone(); off = al(off + n * 4)
    ray_cnt = sc[off:off + n * 4].view(torch.int32).clone(); off = al(off + n * 4)
    acc = sc[off:off + n * 4].view(torch.float32).clone(); off = al(off + n * 4)
    cap = n * S
    q_pos = sc[off:off + cap * 16].view(torch.float32).view(cap, 4)[:counter].clone(); off = al(off + cap * 16)
    q_ray = sc[off:off + cap * 4].view(torch.int32)[:counter].clone()
    return counter, ray_off, ray_cnt, acc, q_pos, q_ray
res = []
for rep in range(6):
    rgb, depth = m.render_rays(rays, white_bg=True, N_samples=S)
    torch.cuda.synchronize()
    res.append((rgb.clone(), depth.clone()) + snapshot())
c0, off0, cnt0, acc0, qp0, qr0 = res[0][2:]
for rep in range(1, 6):
    c, off, cnt, acc, qp, qr = res[rep][2:]
    print("rep", rep, "counter", c == c0, "cnt eq", torch.equal(cnt, cnt0), "acc eq", torch.equal(acc, acc0), "depth eq", torch.equal(res[rep][1], res[0][1]),
          "rgb eq", torch.equal(res[rep][0], res[0][0]), "order same", torch.equal(off, off0))
    # compare per-ray entry lists
    bad_w, bad_rgb = 0, []
    for r in range(n):
        k = int(cnt0[r])
        if k == 0: continue
        a = qp0[int(off0[r]):int(off0[r]) + k]; b = qp[int(off[r]):int(off[r]) + k]
        if not torch.equal(a[:, 3], b[:, 3]): bad_w += 1
        d = (a[:, :3] != b[:, :3]).any(1)
        if d.any():
            for i in d.nonzero().flatten().tolist():
                bad_rgb.append((r, i, int(off0[r]) + i, int(off[r]) + i))
    print("   rays with different weights:", bad_w, " entries with different rgb:", len(bad_rgb))
    print("   (ray, rank, qslot_run0, qslot_this):", bad_rgb[:24])
    print("   slot%32 run0:", sorted(set(s % 32 for _, _, s, _ in bad_rgb)), " this:", sorted(set(s % 32 for _, _, _, s in bad_rgb)))
