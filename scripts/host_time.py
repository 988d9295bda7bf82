"""GPU box: host-side cost of the entry points (time after each un-synchronised call) — finds calls after which the HIP runtime blocks the host."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
m, arrs, A = bench.build_model(torch.device("cuda"))
rays = bench.frames(A)[0].cuda()
idx = torch.arange(0, 640000, 8, device="cuda")
r = rays[idx].contiguous()                      # 80 000 rays spread over the frame
xyz = (torch.rand(2_000_000, 3, device="cuda") * 2 - 1) * 1.4
vd = torch.nn.functional.normalize(torch.randn(2_000_000, 3, device="cuda"), dim=1)
feat = torch.randn(2_000_000, 27, device="cuda")
def trace(name, fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ts = []
    for _ in range(n):
        fn(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    d = [ts[0]] + [ts[i] - ts[i - 1] for i in range(1, n)]
    print(f"{name:28s} total {tot * 1e3:8.2f} ms for {n} calls; host per call median {sorted(d)[n // 2] * 1e3:.3f} ms, max {max(d) * 1e3:.2f} ms at call {d.index(max(d))}; calls > 1 ms: {[i for i, x in enumerate(d) if x > 1e-3]}")
trace("render_rays 80k", lambda: m.render_rays(r, N_samples=512))
fr = bench.frames(A)
from jittor_myc_nerfs_amd import shard_indices
rr = [f[shard_indices(640000, 0, 8, 4096)].contiguous().cuda() for f in fr]       # rank 0's share of an 8-way split of each of the 8 poses
cnt = [0]
def rot():
    cnt[0] += 1
    return m.render_rays(rr[cnt[0] % 8], white_bg=True, N_samples=512, eps_T=None, stats=None, profile=None)
trace("render_rays 8 poses in turn", rot)
if os.environ.get("TVR_HT_EVENTS"):
    import ctypes as C
    from jittor_myc_nerfs_amd import _lib as L
    prof = C.c_void_p()
    L.check(L.lib().tvr_profile_create(20, C.byref(prof)), "tvr_profile_create")
    trace("after tvr_profile_create", rot)
    m.render_rays(rr[0], N_samples=512, profile=prof)
    trace("after one profiled call", rot)
trace("compute_densityfeature 2M", lambda: m.compute_densityfeature(xyz))
trace("compute_appfeature 2M", lambda: m.compute_appfeature(xyz))
trace("mlp_render 2M", lambda: m._mlp_render(vd, feat))
trace("torch add 2M", lambda: feat.add(1.0))
