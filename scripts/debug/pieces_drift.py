import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda:0")
m, arrs, A = bench.build_model(dev, "TensorVMSplit")
S = A["N_samples"]
fr = [f.to(dev) for f in bench.frames(A)]
out = (torch.empty((fr[0].shape[0], 3), device=dev), torch.empty((fr[0].shape[0],), device=dev))
for mode in (None, 0):
    m.render_piece_rays = mode
    m.render_rays(fr[0], white_bg=True, N_samples=S, out=out)
torch.cuda.synchronize()
marks, k = [], 0
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
stats = [torch.zeros(8, dtype=torch.int64, device=dev) for _ in range(2 * nblk)]      # per block: the kernels' own clock probes (s_memtime / s_memrealtime sums, bench.py probe_clock)
for blk in range(nblk):
    for mode in (None, 0):
        m.render_piece_rays = mode
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(8):
            m.render_rays(fr[k % 8], white_bg=True, N_samples=S, out=out, stats=stats[len(marks)]); k += 1
        t1.record()
        marks.append((mode, t0, t1))
torch.cuda.synchronize()
p = [t0.elapsed_time(t1) / 8 for md, t0, t1 in marks if md is None]
o = [t0.elapsed_time(t1) / 8 for md, t0, t1 in marks if md == 0]
def clk(i, a, b):
    st = stats[i].cpu().double()
    return 0.1 * float(st[a]) / float(st[b]) if float(st[b]) > 0 else float("nan")
print("blocks of 8 frames, alternating, one wait at the end; ms per frame, then the march / shade kernels' clocks in GHz (in-kernel probes)")
print("pieces        :", " ".join("%.2f" % x for x in p))
print("one launch set:", " ".join("%.2f" % x for x in o))
print("pieces         march GHz:", " ".join("%.2f" % clk(i, 4, 5) for i, mk in enumerate(marks) if mk[0] is None))
print("one launch set march GHz:", " ".join("%.2f" % clk(i, 4, 5) for i, mk in enumerate(marks) if mk[0] == 0))
print("pieces         shade GHz:", " ".join("%.2f" % clk(i, 6, 7) for i, mk in enumerate(marks) if mk[0] is None))
print("one launch set shade GHz:", " ".join("%.2f" % clk(i, 6, 7) for i, mk in enumerate(marks) if mk[0] == 0))
