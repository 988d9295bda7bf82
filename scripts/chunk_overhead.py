"""Where does the time of the 4096-ray-chunk render loop (BASELINE configs[3]) go: host enqueue vs GPU?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
m, arrs, A = bench.build_model(torch.device("cuda"))
rays = bench.frames(A)[0].cuda()
S = 512
def loop():
    outs = [m.render_rays(rays[c0:c0 + 4096], white_bg=True, N_samples=S) for c0 in range(0, rays.shape[0], 4096)]
    return outs
for _ in range(2): loop()
torch.cuda.synchronize()
t0 = time.perf_counter(); loop(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"157 chunks: host enqueue {1e3 * (t1 - t0):.1f} ms, until GPU idle {1e3 * (t2 - t0):.1f} ms")
# the same through a captured graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    loop()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        outs = loop()
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"hipGraph replay of the 157-chunk frame: {1e3 * (t1 - t0):.1f} ms")
m.render_rays(rays, white_bg=True, N_samples=S); torch.cuda.synchronize(); t0 = time.perf_counter(); m.render_rays(rays, white_bg=True, N_samples=S); torch.cuda.synchronize()
print(f"one merged call: {1e3 * (time.perf_counter() - t0):.1f} ms")
