"""GPU box, diagnostic build (scripts/build_variant.sh timeline -DTVR_MARCH_TIMELINE; TVR_LIB_PATH=.../libtvr_timeline.so): where the time of ONE
march launch goes, per workgroup, in 100 MHz ticks of s_memrealtime — launch skew, LDS fill, first / last wave end, chunks and rays per group.
usage: march_timeline.py [n_rays ...]   (default 4096 80000 640000)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
sizes = [int(a) for a in sys.argv[1:]] or [4096, 80000, 640000]
m, arrs, A = bench.build_model(torch.device("cuda"))
frame = bench.frames(A)[0].cuda()
S = 512
for n in sizes:
    mid = (frame.shape[0] // 2) // 4096 * 4096          # rows through the middle of the image: the dense part of the scene
    rays = frame[mid:mid + n].contiguous() if n <= 4096 * 8 else frame[torch.arange(0, n) % frame.shape[0]].contiguous()
    if n == 80000:                                        # rank 0's share of the 8-way strong split
        from jittor_myc_nerfs_amd import shard_indices
        rays = frame[shard_indices(frame.shape[0], 0, 8, 4096).cuda()].contiguous()
    for rep in range(3):
        st = torch.zeros(32 + 8 * 256, dtype=torch.int64, device="cuda")
        st[32 + 2::8] = 1 << 62                           # atomicMin slot
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        for _ in range(max(2, 200000 // rays.shape[0])):  # keep the GPU busy (and its clock up) right up to the measured call, as a chunk loop does
            m.render_rays(rays, white_bg=True, N_samples=S)
        ev[0].record()
        m.render_rays(rays, white_bg=True, N_samples=S, stats=st)
        ev[1].record()
        torch.cuda.synchronize()
    t = st[32:].view(256, 8).cpu().numpy().astype(np.float64)
    live = t[:, 0] > 0
    t = t[live]
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    sc = st[:8].cpu().numpy().astype(np.float64)
    print(f"  clock: march {0.1 * sc[4] / max(sc[5], 1):.2f} GHz, shade {0.1 * sc[6] / max(sc[7], 1):.2f} GHz (in-kernel probes)")
    print(f"n_rays {rays.shape[0]}: {live.sum()} groups; whole tvr_render (3 kernels, events) {ev[0].elapsed_time(ev[1]) * 1e3:.0f} us")
    print(f"  group start   : min {us(t[:, 0]).min():7.1f}  median {np.median(us(t[:, 0])):7.1f}  max {us(t[:, 0]).max():7.1f} us")
    print(f"  LDS filled    : min {us(t[:, 1]).min():7.1f}  median {np.median(us(t[:, 1])):7.1f}  max {us(t[:, 1]).max():7.1f} us   (fill itself: median {np.median(t[:, 1] - t[:, 0]) / 100:.1f} us)")
    print(f"  first wave end: min {us(t[:, 2]).min():7.1f}  median {np.median(us(t[:, 2])):7.1f}  max {us(t[:, 2]).max():7.1f} us")
    print(f"  last wave end : min {us(t[:, 3]).min():7.1f}  median {np.median(us(t[:, 3])):7.1f}  max {us(t[:, 3]).max():7.1f} us")
    if t[:, 7].max() > 0:                                 # the shade kernel of the same tvr_render (csrc/tvr_shade16.hip under the same build flag)
        s0 = t[:, 6].min()
        su = lambda x: (x - s0) / 100.0
        e = su(t[:, 7])
        print(f"  shade16: group start min {su(t[:, 6]).min():7.1f} median {np.median(su(t[:, 6])):7.1f} max {su(t[:, 6]).max():7.1f} us; last wave end min {e.min():8.1f} p10 {np.percentile(e, 10):8.1f} "
              f"median {np.median(e):8.1f} p90 {np.percentile(e, 90):8.1f} max {e.max():8.1f} us -> idle CU-time in the tail {100 * (e.max() - e.mean()) / e.max():.2f} % of the kernel")
        x8 = [e[(np.arange(len(e)) % 8) == k].mean() for k in range(8)]
        print("           mean end per XCD (blockIdx % 8): " + " ".join(f"{v:8.1f}" for v in x8))
        if os.environ.get("TL_XCDCLK"):                   # a -DS16_XCDCLK build: slots 4 / 5 hold wave 0's shader-clock / 100 MHz ticks over the shade kernel
            ghz = 0.1 * t[:, 4] / np.maximum(t[:, 5], 1)
            print("           shader clock per XCD, GHz      : " + " ".join(f"{ghz[(np.arange(len(ghz)) % 8) == k].mean():8.3f}" for k in range(8)))
    ch, ry = t[:, 4], t[:, 5]
    busy = (t[:, 3] - t[:, 1]) / 100.0
    print(f"  chunks / group: min {ch.min():.0f} median {np.median(ch):.0f} max {ch.max():.0f}; rays / group median {np.median(ry):.0f}; us per chunk and group (busy / chunks): median {np.median(busy / np.maximum(ch, 1)):.3f}")
    print(f"  per-wave chain: {np.median(busy) :.1f} us busy per group for {np.median(ch) / 16:.1f} chunks per wave -> {np.median(busy) / max(np.median(ch) / 16, 1e-9):.2f} us per chunk in a wave's chain")
