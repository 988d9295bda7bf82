#!/usr/bin/env python3
"""GPU box (round 5): is the shade kernel's time a function of the DATA its matrix cores see?

The same library, the same frame (bench.py's scene A, 800 x 800 x 512), the same launches; only the appearance network's weights differ:
  real       the bench scene's W1 / W2
  zeroW2     W2 = 0                     (layer 2's A operands all zero; layer 1 unchanged, so layer 2's B operands are the real activations)
  zeroW1W2   W1 = W2 = 0                (both layers multiply zeros)
  constW2    W2 = 0.01 everywhere       (every fragment of W2 identical: a library that reads every other W2 fragment and reuses it computes the SAME pixels)
  rep16      every 16-row block of W1 / W2 a copy of block 0, every 32-column k-step a copy of k-step 0 (the fragments repeat, values real)
  negb1      b1 = -1e4                  (layer 1's relu outputs all zero: layer 2's B operands zero, its A operands real)
Prints ms and the in-kernel clock of the shade kernel per variant, two interleaved rounds.  A kernel bound by issue slots takes the same time whatever the numbers are;
one held by the chip's power management does not."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from jittor_myc_nerfs_amd import TensorVMSplit, synthetic, _lib as L      # noqa: E402


def make(variant):
    A, H = synthetic.SCENE_A, synthetic.HYPER
    arrs = dict(synthetic.make_scene_arrays(A["gridSize"], A["aabb"]))
    W1, W2, b1 = np.array(arrs["W1"]), np.array(arrs["W2"]), np.array(arrs["b1"])
    if variant in ("zeroW2", "zeroW1W2"):
        W2[:] = 0
    if variant == "zeroW1W2":
        W1[:] = 0
    if variant == "rep16":
        for W in (W1, W2):
            blk = W[:16, :32].copy()
            for r in range(0, W.shape[0], 16):
                for c in range(0, W.shape[1], 32):
                    w = min(32, W.shape[1] - c)
                    W[r:r + 16, c:c + w] = blk[:, :w]
    if variant == "constW2":                                     # every fragment of W2's LDS image is the same 1 KB, whatever the packing's row / column order
        W2[:] = 0.01
    if variant == "negb1":
        b1[:] = -1e4
    arrs["W1"], arrs["W2"], arrs["b1"] = W1, W2, b1
    m = TensorVMSplit(arrs["aabb"], A["gridSize"], "cuda", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27, near_far=A["near_far"],
                      shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=H["density_shift"], distance_scale=H["distance_scale"],
                      rayMarch_weight_thres=H["rayMarch_weight_thres"], pos_pe=6, view_pe=2, fea_pe=2, featureC=128, step_ratio=A["step_ratio"],
                      fea2denseAct=H["fea2denseAct"])
    m.load_arrays(arrs)
    m.fp16_range_check = "off"                                   # the same kernel instance for every variant
    return m


def timeit(m, fr, steps=6):
    prof = C.c_void_p()
    L.check(L.lib().tvr_profile_create(steps, C.byref(prof)), "tvr_profile_create")
    stats = torch.zeros(64, dtype=torch.int64, device="cuda")
    for k in range(2):
        m.render_rays(fr[k % len(fr)], white_bg=True, N_samples=512)
    torch.cuda.synchronize()
    for k in range(steps):
        m.render_rays(fr[k % len(fr)], white_bg=True, N_samples=512, profile=prof, stats=stats)
    torch.cuda.synchronize()
    ms = (C.c_float * 3)()
    n = max(L.lib().tvr_profile_read(prof, C.byref(ms)), 1)
    L.lib().tvr_profile_destroy(prof)
    return ms[0] / n, ms[1] / n


def main():
    A = synthetic.SCENE_A
    fr = [f.to("cuda") for f in bench.frames(A)]
    names = tuple(os.environ.get("WP_VARIANTS", "real,zeroW2,zeroW1W2,rep16,negb1").split(","))
    models = {v: make(v) for v in names}
    if os.environ.get("WP_DIGEST"):                              # sha256 of one frame's pixels: two libraries that print the same digest computed the same numbers
        import hashlib
        for v in names:
            rgb, depth = models[v].render_rays(fr[0], white_bg=True, N_samples=512)
            print(f"{os.environ.get('WP_TAG', ''):10s}{v:10s} sha256(rgb) {hashlib.sha256(rgb.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
    loop_s = float(os.environ.get("WP_LOOP_S", "0"))               # scripts/power_vs_data.sh: keep one variant rendering for this many seconds while rocm-smi samples the board
    if loop_s > 0:
        import time
        m = models[names[0]]
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < loop_s:
            for k in range(8):
                m.render_rays(fr[k % len(fr)], white_bg=True, N_samples=512)
            torch.cuda.synchronize()
            n += 8
        print(f"{names[0]:10s} loop {n} frames  {(time.perf_counter() - t0) / n * 1e3:6.2f} ms per frame", flush=True)
        return
    for rnd in (1, 2):
        for v in names:
            march, shade = timeit(models[v], fr)
            print(f"{os.environ.get('WP_TAG', ''):10s}{v:10s} round {rnd}  march {march:6.2f} ms  shade {shade:6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
