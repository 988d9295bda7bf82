"""GPU box: how unevenly the march's work is spread over the rays of a training batch (4096 random rays of the bench scene, 1039 samples): non-empty 64-sample chunks per ray."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
dev = torch.device("cuda:0")
m, arrs, A = bench.build_model(dev, "TensorVMSplit")
fr = bench.frames(A)
allrays = torch.cat(fr[:4]).to(dev)
nS = int(np.linalg.norm(A["gridSize"]) / 0.5)
g = torch.Generator(device="cuda").manual_seed(0)
for trial in range(3):
    idx = torch.randint(0, allrays.shape[0], (4096,), device=dev, generator=g)
    rgb, depth, dd = m.render_rays(allrays[idx], white_bg=True, N_samples=nS, dense=True)
    valid = dd["valid"].bool()                       # [4096, nS]
    w = dd["weight"]
    T_alive = (1.0 - torch.cumsum(w, 1)) > 1e-4     # rough: before early termination
    n = valid.shape[1]
    pad = (64 - n % 64) % 64
    v = torch.nn.functional.pad(valid & T_alive, (0, pad)).view(valid.shape[0], -1, 64)
    chunks = v.any(-1).sum(1).float()                # non-empty chunks per ray
    samples = (valid & T_alive).sum(1).float()
    print(f"trial {trial}: non-empty chunks per ray mean {chunks.mean():.2f} max {chunks.max():.0f}  p90 {chunks.quantile(0.9):.0f}; evaluated samples per ray mean {samples.mean():.0f} max {samples.max():.0f}; "
          f"max / mean = {chunks.max() / chunks.mean():.2f} (chunks), {samples.max() / samples.mean():.2f} (samples)")
