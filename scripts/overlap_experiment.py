#!/usr/bin/env python3
"""can march (VALU / L1-bound, no MFMA) and shade (MFMA + L1) of different ray batches overlap?

Both are persistent kernels that fill a CU's LDS (march 58 KB lines + lists, shade 159 KB weight image), so a march workgroup and a shade
workgroup cannot share a CU; the only way to run them side by side is on disjoint CU sets (grid size = number of CUs each may take).
This script renders one 800x800 frame as K ray batches
  serial   : one stream, full grids (what tvr_render does per batch)
  overlap  : two streams (batch i on stream i % 2, own scratch each), grids limited to G_march + G_shade <= 256 through the
             TVR_EXP_GRID_* experiment hooks, so that march(i+1) and shade(i) can be resident together.  The hooks exist only in a library built
             with them:  scripts/build_variant.sh expgrid -DTVR_EXP_GRID ; TVR_LIB_PATH=.../lib/variants/libtvr_expgrid.so python scripts/overlap_experiment.py
and prints wall times (median of 7 repetitions)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

def run(K, overlap, gm, gs):
    os.environ.pop("TVR_EXP_GRID_MARCH", None); os.environ.pop("TVR_EXP_GRID_SHADE", None)
    if overlap:
        os.environ["TVR_EXP_GRID_MARCH"], os.environ["TVR_EXP_GRID_SHADE"] = str(gm), str(gs)
    dev = torch.device("cuda")
    models = [bench.build_model(dev)[0] for _ in range(2 if overlap else 1)]
    A = bench.build_model(dev)[2]
    rays = bench.frames(A)[0].to(dev)
    n = rays.shape[0]
    parts = [rays[i * n // K:(i + 1) * n // K].contiguous() for i in range(K)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    def frame():
        if not overlap:
            for p in parts:
                models[0].render_rays(p, white_bg=True, N_samples=512)
        else:
            for i, p in enumerate(parts):
                with torch.cuda.stream(streams[i % 2]):
                    models[i % 2].render_rays(p, white_bg=True, N_samples=512)
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); frame(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[3] * 1e3

if __name__ == "__main__":
    print(f"serial, 1 batch, full grids              : {run(1, False, 0, 0):6.2f} ms")
    for K in (4, 8):
        print(f"serial, {K} batches, full grids            : {run(K, False, 0, 0):6.2f} ms")
        for gm, gs in ((128, 128), (96, 160), (64, 192)):
            print(f"2 streams, {K} batches, march {gm:3d} + shade {gs:3d} CUs: {run(K, True, gm, gs):6.2f} ms")
