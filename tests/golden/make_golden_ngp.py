"""Golden fixture of the alt path, generated from oracle/ngp_oracle.c (scalar C): tests/golden/ngp.npz.

Like the TensoRF fixtures these are outputs of the RESTATED path (parity unpinned at the Jittor/CUDA boundary): they pin the
oracle against later edits and give data-only expectations that travel without /root/reference.  The scene (52 MB of hash grid)
is regenerated from its seed; the fixture holds sha256 digests of its arrays, the rays, and per-stage outputs:
step counts / bases, a digest of all sample rows plus the first 512 rows, network outputs of those rows, the picture.

    python tests/golden/make_golden_ngp.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import NGP_AABB_SCALE, ngp_camera_rays, ngp_edge_rays  # noqa: E402
from jittor_myc_nerfs_amd import synthetic  # noqa: E402
from oracle import ngp_oracle as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    levels = N.grid_levels(NGP_AABB_SCALE)
    arrs = synthetic.make_ngp_scene_arrays(levels["offsets"])
    bits, mean = N.update_bitfield(arrs["density_grid"])
    arrs["density_grid_bitfield"] = bits
    o, d = ngp_camera_rays(40, 40, pose_index=2)
    eo, ed = ngp_edge_rays()
    o, d = np.concatenate([o, eo]), np.concatenate([d, ed])
    rng = N.Pcg32(1337)
    rng.advance()                                                      # "second call" state
    coords, index, numsteps, counter, startt = N.sample(o, d, bits, NGP_AABB_SCALE, rng.state)
    head = coords[:512]
    enc, cells = N.hash_encode_c(levels, arrs["grid"], head[:, :3], want_cells=True)
    net = N.network_c(levels, arrs, coords)
    rgb, T = N.composite_c(net, coords, numsteps)
    out = {
        "levels.offsets": levels["offsets"], "levels.scale": levels["scale"],
        "sha.grid": sha(arrs["grid"]), "sha.density_grid": sha(arrs["density_grid"]), "sha.bitfield": sha(bits),
        **{f"sha.{k}": sha(arrs[k]) for k in arrs if k.endswith(".weight")},
        "bitfield_popcount": np.array([int(np.unpackbits(bits[i * 262144:(i + 1) * 262144]).sum()) for i in range(5)]),
        "density_grid_mean": np.float32(mean),
        "rays_o": o, "rays_d": d, "rng_state": np.array(rng.state, np.uint64),
        "numsteps": numsteps, "ray_index": index, "counter": counter, "startt": startt,
        "sha.coords": sha(coords), "coords_head": head, "cells_head": cells[:64], "enc_head": enc,
        "sh_head": N.sh_encode_c(head[:, 4:]), "net_head": net[:512], "sha.net": sha(net), "rgb": rgb, "T": T,
    }
    np.savez_compressed(os.path.join(HERE, "ngp.npz"), **out)
    print("ngp.npz:", os.path.getsize(os.path.join(HERE, "ngp.npz")), "bytes;", coords.shape[0], "samples")


if __name__ == "__main__":
    main()
