#!/usr/bin/env python3
"""GPU box: how much of what the shade kernel's gather fetches per 32-entry tile is DISTINCT — the prize of tap reuse between the
consecutive samples of a ray (round-2 review, item 3a).  A statistics pass over the bench frame's own appearance-sample queue; no kernel change.

    python3 scripts/tile_reuse_stats.py > profiles/r03_tile_reuse_stats.json

Per tile (32 consecutive queue entries = the columns of one wave's MFMA tile) and per plane: distinct cells (x0,y0), distinct plane texels (the
union of the four corners), distinct line texels (union of l0, l0+1); rays per tile.  The kernel today loads 4 plane texels + 2 line texels per
entry and plane: 32 x 4 = 128 and 32 x 2 = 64 texel reads per tile and plane, 108 wave-level dwordx4 loads per tile in all."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    import bench
    from jittor_myc_nerfs_amd import _lib as L
    dev = torch.device("cuda", 0)
    model, arrs, A = bench.build_model(dev)
    S = A["N_samples"]
    out = {"what": "distinct cells / texels per 32-entry shade tile on the bench frames (scene A, 800x800, 512 samples, eps_T = 1e-4)", "poses": {}}
    tot = None
    for pose in (0, 1):
        rays = bench.frames(A)[pose].to(dev)
        n = rays.shape[0]
        sc = model._ensure_scene()
        lay = L.ScratchLayout()
        L.check(L.lib().tvr_scratch_describe(n, S, C.byref(lay)), "describe")
        scratch = torch.empty(lay.total, dtype=torch.uint8, device=dev)
        depth = torch.empty(n, device=dev)
        L.check(L.lib().tvr_march_forward(sc, rays.data_ptr(), n, S, None, float(model.rayMarch_weight_thres), depth.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                          torch.cuda.current_stream(dev).cuda_stream), "march")
        M = int(scratch[lay.counter:lay.counter + 4].view(torch.int32).item())
        M32 = M // 32 * 32
        q = scratch[lay.q_pos:lay.q_pos + M32 * 16].view(torch.float32).view(M32, 4)
        ray = scratch[lay.q_ray:lay.q_ray + M32 * 4].view(torch.int32).view(-1, 32)
        g = torch.tensor([float(x - 1) for x in model.gridSize], device=dev)
        f = ((q[:, :3] + 1.0) / 2.0) * g                                   # un-normalised grid coordinates (tvr_device.h unnorm)
        i0 = torch.floor(f).long()                                          # [M,3] cell per axis
        T = M32 // 32
        res = {"entries": M, "tiles": T, "entries_per_ray": M / n}
        rays_per_tile = (ray[:, 1:] != ray[:, :-1]).sum(1) + 1
        res["tiles_with_one_ray_frac"] = float((rays_per_tile == 1).float().mean())
        res["rays_per_tile_mean"] = float(rays_per_tile.float().mean())
        mats, vecs = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)
        W = 4096

        def distinct_per_tile(keys):                                       # keys [T, k] int64 -> mean / max number of distinct values per row
            srt, _ = torch.sort(keys, dim=1)
            d = (srt[:, 1:] != srt[:, :-1]).sum(1) + 1
            return d

        cells, ptex, ltex = [], [], []
        for p in range(3):
            a, b = mats[p]
            key = (i0[:, b] * W + i0[:, a]).view(T, 32)
            cells.append(distinct_per_tile(key))
            corners = torch.cat([key, key + 1, key + W, key + W + 1], dim=1)
            ptex.append(distinct_per_tile(corners))
            l0 = i0[:, vecs[p]].view(T, 32)
            ltex.append(distinct_per_tile(torch.cat([l0, l0 + 1], dim=1)))
        cells, ptex, ltex = torch.stack(cells).float(), torch.stack(ptex).float(), torch.stack(ltex).float()
        res["distinct_cells_per_tile_and_plane"] = {"mean": float(cells.mean()), "p90": float(cells.flatten().kthvalue(int(0.9 * cells.numel())).values), "max": float(cells.max())}
        res["distinct_plane_texels_per_tile_and_plane"] = {"mean": float(ptex.mean()), "max": float(ptex.max()), "fetched_today": 128}
        res["distinct_line_texels_per_tile_and_plane"] = {"mean": float(ltex.mean()), "max": float(ltex.max()), "fetched_today": 64}
        # bytes: a texel is 192 B (48 channels); per tile the kernel moves 3 planes x (128 + 64) texel reads x 192 B = 110.6 KB through the vector L1
        distinct_kb = float((ptex.sum(0) + ltex.sum(0)).mean()) * 192 / 1024
        res["distinct_tap_KB_per_tile"] = distinct_kb
        res["fetched_tap_KB_per_tile"] = 3 * 192 * 192 / 1024
        res["dedupe_floor_loads_per_tile"] = distinct_kb                    # 1 KB per wave-level dwordx4 load
        out["poses"][str(pose)] = res
        del scratch
    out["reading"] = ("a tile's 108 KB of tap reads hold only ~distinct_tap_KB_per_tile of distinct bytes; reaching them needs a cross-lane hand-over of whole texels "
                      "(LDS staging: 8 KB per wave and k-step, the 158 KB weight image leaves 5.5 KB; ds_bpermute: 4 B per lane and instruction) — DESIGN.md 4.2")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
