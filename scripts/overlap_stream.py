#!/usr/bin/env python3
"""GPU box (round 5): a STREAM of frames on one card with the march of one frame beside the shade kernel of another, on disjoint CU sets.

The frame is energy-limited as a whole: the shade kernel runs against the chip's power limit (on half the CUs it clocks 2.34 GHz instead of 1.67 and
takes 1.42 x, not 2 x, as long), the march is bound by the L1 path per CU.  Two streams, two models (own scratch), frames alternating between them, the kernels' grids limited
to G_march + G_shade <= 256 workgroups (a -DTVR_EXP_GRID library: scripts/build_variant.sh expgrid -DTVR_EXP_GRID) so that a march and a shade workgroup never wait for
each other's LDS.  Prints ms per frame over 24 frames against the plain serial loop.  (Round 2 measured -2.4 % for batches of ONE frame, profiles/r02_overlap_experiment.txt.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def run(overlap, gm, gs, n_frames=24):
    os.environ.pop("TVR_EXP_GRID_MARCH", None); os.environ.pop("TVR_EXP_GRID_SHADE", None)
    if gm:
        os.environ["TVR_EXP_GRID_MARCH"], os.environ["TVR_EXP_GRID_SHADE"] = str(gm), str(gs)
    dev = torch.device("cuda")
    models = [bench.build_model(dev)[0] for _ in range(2 if overlap else 1)]
    A = bench.build_model(dev)[2]
    fr = [f.to(dev) for f in bench.frames(A)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [[torch.empty((fr[0].shape[0], 3), device=dev), torch.empty((fr[0].shape[0],), device=dev)] for _ in range(2)]

    def go(n):
        for k in range(n):
            if overlap:
                with torch.cuda.stream(streams[k % 2]):
                    models[k % 2].render_rays(fr[k % len(fr)], white_bg=True, N_samples=512, out=tuple(outs[k % 2]))
            else:
                models[0].render_rays(fr[k % len(fr)], white_bg=True, N_samples=512, out=tuple(outs[0]))
    go(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(n_frames); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_frames * 1e3


if __name__ == "__main__":
    print(f"serial, one stream, full grids                       : {run(False, 0, 0):6.2f} ms per frame")
    print(f"two streams, full grids (kernels queue for the CUs)  : {run(True, 0, 0):6.2f}")
    for gm, gs in ((128, 128), (112, 144), (96, 160), (80, 176), (64, 192)):
        print(f"two streams, march {gm:3d} + shade {gs:3d} workgroups          : {run(True, gm, gs):6.2f}")
    print(f"serial, one stream, full grids (again)               : {run(False, 0, 0):6.2f}")
