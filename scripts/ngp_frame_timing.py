"""Alt path (BASELINE configs[4]): time one 800x800 Instant-NGP frame on cuda:0 — the reference's 4096-ray slab loop (render_img)
and the one-pass frame path (render_frame) — on the seeded synthetic scene (synthetic.make_ngp_scene_arrays, aabb_scale 4).
Usage: python scripts/ngp_frame_timing.py [--size 800] [--frames 5] [--loop]"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jittor_myc_nerfs_amd import ngp, rays as R, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--loop", action="store_true", help="also time the reference-style slab loop")
    ap.add_argument("--aabb-scale", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = ngp.NGPNetworks(a.aabb_scale).to(dev)
    sampler = ngp.DensityGridSampler(model, a.aabb_scale, rng=ngp.Pcg32(1337)).to(dev)
    ngp.load_scene_arrays(model, sampler, synthetic.make_ngp_scene_arrays(model.pos_encoder.offsets))
    W = H = a.size
    focal = 0.5 * W / math.tan(0.5 * 0.6911)
    poses = R.sphere_poses(8, 4.0)
    frames = [ngp.generate_rays(ngp.matrix_nerf2ngp(p), W, H, (focal, focal), device=dev) for p in poses[:a.frames]]
    stats = {}
    sampler.render_frame(*frames[0], stats=stats)                      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o, d in frames:
        img = sampler.render_frame(o, d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(frames)
    tot, ev = stats["samples"], stats["evaluated"]
    print(f"NGP {W}x{H} frame, fused (march + render kernel): {dt * 1e3:.1f} ms / frame, {tot / 1e6:.1f} M samples marched, {ev / 1e6:.1f} M evaluated "
          f"({tot / dt / 1e9:.2f} G marched samples/s, {W * H / dt / 1e6:.2f} M rays/s); rgb range {float(img.min()):.3f}..{float(img.max()):.3f}")
    hint = int(tot / (W * H) * 1.3) + 8
    sampler.render_frame_rows(*frames[0], samples_per_ray_hint=hint)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o, d in frames:
        sampler.render_frame_rows(o, d, samples_per_ray_hint=hint)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(frames)
    print(f"NGP {W}x{H} frame, row-level one pass (sample + network + composite over all rows): {dt * 1e3:.1f} ms / frame ({tot / dt / 1e9:.2f} G samples/s)")
    if a.loop:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o, d in frames[:2]:
            ngp.render_img(sampler, model, o, d)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / 2
        print(f"NGP {W}x{H} frame, reference-style 4096-ray slab loop (one host read per slab): {dt2 * 1e3:.1f} ms / frame")


if __name__ == "__main__":
    main()
