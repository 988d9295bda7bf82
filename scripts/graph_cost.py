"""GPU box: what a hipGraph replay of one tvr_render (header clear, march, shade, composite) costs against the four plain launches — for ONE graph replayed over and
over, and for K different graphs (different ray sets / output buffers) replayed in turn, which is what a stream of frames needs."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
m, arrs, A = bench.build_model(torch.device("cuda"))
fr = bench.frames(A)
from jittor_myc_nerfs_amd import shard_indices
idx = shard_indices(640000, 0, 8, 4096)
def timeit(fn, k=64):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(k): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for K in (1, 2, 4, 16):
    sets = []
    for j in range(K):
        rays = fr[j % 8][idx].cuda().contiguous()
        sets.append((rays, torch.empty((rays.shape[0], 3), device="cuda"), torch.empty((rays.shape[0],), device="cuda")))
    fs = [(lambda r=r, a=a, b=b: m.render_rays(r, white_bg=True, N_samples=512, out=(a, b))) for r, a, b in sets]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fs: f()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gs = []
    for f in fs:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): f()
        gs.append(g)
    for g in gs: g.replay()
    torch.cuda.synchronize()
    print("%2d ray sets in turn: plain %.3f ms per call, graph replay %.3f ms" % (K, timeit(lambda i=0: fs[i % K]()), timeit(lambda i=0: gs[i % K].replay())))
