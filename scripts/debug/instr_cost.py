import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from jittor_myc_nerfs_amd import _lib as L
dev = torch.device("cuda:0")
m, arrs, A = bench.build_model(dev, "TensorVMSplit")
S = A["N_samples"]
fr = [f.to(dev) for f in bench.frames(A)]
n = fr[0].shape[0]
out = (torch.empty((n, 3), device=dev), torch.empty((n,), device=dev))
def run(pieces, prof, stats, frames=20):
    m.render_piece_rays = pieces
    p = None
    if prof:
        p = C.c_void_p(); L.check(L.lib().tvr_profile_create(frames + 4, C.byref(p)), "c")
    st = torch.zeros(8, dtype=torch.int64, device=dev) if stats else None
    for k in range(3): m.render_rays(fr[k], white_bg=True, N_samples=S, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(frames): m.render_rays(fr[k % 8], white_bg=True, N_samples=S, out=out, profile=p, stats=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / frames * 1e3
    if p: L.lib().tvr_profile_destroy(p)
    return dt
for rnd in range(2):
    for pieces in (None, 0):
        print("round", rnd, "pieces" if pieces is None else "one launch set", " plain %.3f  +profile %.3f  +stats %.3f  +both %.3f" % (run(pieces, 0, 0), run(pieces, 1, 0), run(pieces, 0, 1), run(pieces, 1, 1)), flush=True)
