"""GPU box: how evenly the round-robin shard tiles spread a frame's work over 8 ranks, per tile size.  Work proxies per ray: 1 (ray count), appearance samples
(the shade kernel's work, exact: ray_cnt of the march queue) and samples evaluated by the march (dense `valid` mask, up to the early stop is not visible: in-box
samples).  Prints max-over-ranks / mean for each proxy — the factor by which the slowest rank's share exceeds an equal split."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import torch, bench
from jittor_myc_nerfs_amd import _lib as L, shard_indices
m, arrs, A = bench.build_model(torch.device("cuda"))
fr = bench.frames(A)
S, R, W = A["N_samples"], fr[0].shape[0], 8
lay = L.ScratchLayout()
L.check(L.lib().tvr_scratch_describe(R, S, C.byref(lay)), "tvr_scratch_describe")
print("tile   rays max/mean   app-samples max/mean   in-box samples max/mean      (8 ranks, worst of the 8 bench poses)")
app, box = [], []
for p in range(8):
    rays = fr[p].cuda()
    m.render_rays(rays, white_bg=True, N_samples=S)
    torch.cuda.synchronize()
    app.append(m._scratch[lay.ray_cnt:lay.ray_cnt + 4 * R].view(torch.int32).clone().double())
    # in-box samples per ray from the slab test (sample_ray's positions): count of samples inside the aabb
    o, d = rays[:, :3], rays[:, 3:]
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    a, b = (m.aabb[1].cuda() - o) / vec, (m.aabb[0].cuda() - o) / vec
    tmin = torch.minimum(a, b).amax(1).clamp(m.near_far[0], m.near_far[1])
    tmax = torch.maximum(a, b).amin(1)
    box.append(((tmax - tmin).clamp(min=0) / float(m.stepSize)).clamp(max=S).double())
for tile in (4096, 2048, 1024, 512, 256, 128):
    worst = [0.0, 0.0, 0.0]
    for p in range(8):
        sums = [[], [], []]
        for r in range(W):
            idx = shard_indices(R, r, W, tile).cuda()
            sums[0].append(float(idx.numel())); sums[1].append(float(app[p][idx].sum())); sums[2].append(float(box[p][idx].sum()))
        for k in range(3):
            worst[k] = max(worst[k], max(sums[k]) / (sum(sums[k]) / W))
    print("%5d   %.4f          %.4f                 %.4f" % (tile, *worst))
