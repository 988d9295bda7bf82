"""Generates the committed golden fixtures from oracle (a) (oracle/tensorf_oracle.py, torch CPU).

The reference itself cannot run here (Jittor absent), so these vectors are outputs of the
restated CPU path, not of the reference: parity stays "unpinned" at the Jittor boundary.  They pin the ORACLE
(so that a later edit of the restatement, of torch, or of the synthetic generator cannot silently move the
target) and give the GPU tests data-only expectations that travel without /root/reference.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jittor_myc_nerfs_amd import rays as R, synthetic  # noqa: E402
from oracle import tensorf_oracle as TO  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

TINY = dict(gridSize=[16, 20, 24], aabb=[[-1.5, -1.2, -1.0], [1.5, 1.2, 1.0]], near_far=[2.0, 6.0], step_ratio=0.5,
            N_samples=48, cam_radius=4.0, img=8, camera_angle_x=0.6911, seed=7)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tiny_scene(alpha=False, ref=False, npp=None):
    return synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=TINY["seed"],
                                       alpha_grid=[12, 10, 14] if alpha else None, ref=ref, npp=npp)


def main_npp(hyper):
    """(v) NerfPlusPlus (models/nerfplusplus.py) on the tiny scene, bounding sphere radius 6: the background network's arrays, the two
    random draws the reference takes per call (injected), and the render with its foreground / background parts."""
    arrs = tiny_scene(npp=6.0)
    sc = TO.scene_from_arrays(arrs, **hyper)
    rays = tiny_rays()
    g = np.random.default_rng(17)
    rf = g.random((rays.shape[0], TINY["N_samples"])).astype(np.float32)
    rb = g.random((rays.shape[0], 512)).astype(np.float32)
    d = dump_to_np(TO.execute_npp(sc, rays, N_samples=TINY["N_samples"], rand_fg=rf, rand_bg=rb, dump=True))
    out = dict(rays=rays.numpy(), rand_fg=rf, rand_bg=rb, **{f"out.{k}": d[k] for k in ("rgb_map", "depth_map", "fg_rgb_map", "bg_rgb_map", "bg_lambda",
                                                                                       "z_vals", "valid", "app_mask", "weight")},
               **{f"scene.{k}": v for k, v in arrs.items() if k.startswith("bg")})
    np.savez_compressed(os.path.join(HERE, "tiny_npp.npz"), **out)
    print("tiny_npp.npz", os.path.getsize(os.path.join(HERE, "tiny_npp.npz")) // 1024, "KiB; valid/ray", float(d["valid"].sum(1).mean()),
          "app", int(d["app_mask"].sum()), "bg_lambda>0:", int((d["bg_lambda"] > 0).sum()), "of", rays.shape[0])

REF_KEYS = ("W1", "b1") + tuple(f"{n}_{s}" for n in ("normal", "diffuse", "specular", "rho") for s in ("W", "b"))


def main_ref(hyper):
    """(iv) REFTensoRF (models/REFTensoRF.py) on the tiny scene: the arrays it adds to tiny_dump's scene, the full render,
    the appearance-branch intermediates at the shaded samples, and one masked + jittered + black-background variant."""
    arrs = tiny_scene(ref=True)
    sc = TO.scene_from_arrays(arrs, **hyper)
    rays = tiny_rays()
    d = dump_to_np(TO.execute(sc, rays, white_bg=True, N_samples=TINY["N_samples"], dump=True))
    am = torch.tensor(d["app_mask"].astype(bool))
    xyz_n = torch.tensor(d["xyz_norm"])[am]
    dirs = rays[:, None, 3:6].expand(-1, TINY["N_samples"], -1)[am]
    f, rgb_d, tint, normal, rho = TO.compute_appfeature_ref(sc, xyz_n)
    nn = TO.jt_normalize(normal)
    dot = ((-dirs) * nn).sum(1, keepdim=True)
    refl = 2 * dot * nn - (-dirs)
    rgb_s, mlp_in = TO.mlp_render_fea_ref(sc, refl, f, -dot, return_in=True)
    out = dict(rays=rays.numpy(), rgb_map=d["rgb_map"], depth_map=d["depth_map"], acc_map=d["acc_map"], app_mask=d["app_mask"],
               rgb=d["rgb"], weight=d["weight"], penalty=np.float32(sc.penalty.item()),
               app_xyz_norm=xyz_n.numpy(), app_dirs=dirs.numpy(), app_feature=f.numpy(), rgb_d=rgb_d.numpy(), specular_tint=tint.numpy(),
               normal_vector=normal.numpy(), rho=rho.numpy(), reflection=refl.numpy(), dot_product=dot.numpy(), mlp_in=mlp_in.numpy(),
               rgb_s=rgb_s.numpy(), **{f"scene.{k}": arrs[k] for k in REF_KEYS})
    arrs_a = tiny_scene(alpha=True, ref=True)
    jit = np.random.default_rng(5).random(rays.shape[0]).astype(np.float32)
    sca = TO.scene_from_arrays(arrs_a, **hyper)
    da = dump_to_np(TO.execute(sca, rays, white_bg=False, N_samples=TINY["N_samples"], jitter=jit, dump=True))
    out.update({"wb0_am1_jit.jitter": jit, "wb0_am1_jit.rgb_map": da["rgb_map"], "wb0_am1_jit.depth_map": da["depth_map"],
                "wb0_am1_jit.app_mask": da["app_mask"]})
    np.savez_compressed(os.path.join(HERE, "tiny_ref.npz"), **out)
    print("tiny_ref.npz", os.path.getsize(os.path.join(HERE, "tiny_ref.npz")) // 1024, "KiB; app", int(d["app_mask"].sum()),
          "rgb range", float(d["rgb"].min()), float(d["rgb"].max()), "penalty", float(sc.penalty))


def tiny_rays():
    M = R.sphere_poses(4, TINY["cam_radius"])[1]
    return R.frame_rays(M, TINY["img"], TINY["img"], TINY["camera_angle_x"])


def edge_rays(aabb):
    """Hand-built rays for the edge cases SURVEY.md §8c lists (tensorBase.py:345-360 behaviours)."""
    lo, hi = np.asarray(aabb[0], np.float32), np.asarray(aabb[1], np.float32)
    r = []
    r.append([0.2, 0.1, 4.0, 0.0, 0.0, -1.0])                  # d.x = d.y = 0 exactly -> vec substitution 1e-6
    r.append([0.0, 4.0, 0.3, 0.0, -1.0, 0.0])                  # axis-aligned along -y
    r.append([0.1, -0.2, 0.05, 0.577, 0.577, 0.578])           # origin inside the box -> t_min clamps to near
    r.append([5.0, 5.0, 5.0, 0.0, 0.0, 1.0])                   # misses the box entirely -> all invalid, rgb = white
    r.append([4.0, 0.0, 0.0, -1.0, 0.0, 0.0])                  # enters through the +x face, grazing centre line
    r.append([float(hi[0]), 3.0, 0.0, 0.0, -1.0, 0.0])         # travels exactly ON the hi x face (index W-1, x1 padded)
    r.append([float(lo[0]), 3.0, 0.2, 0.0, -1.0, 0.0])         # exactly on the lo x face
    r.append([0.0, 0.0, -4.0, 1e-9, 0.0, 1.0])                 # tiny (non-zero) direction component
    r.append([2.5, 2.5, 2.5, -0.577, -0.577, -0.578])          # corner-to-corner diagonal
    r.append([0.3, 3.5, 0.1, 0.0, -1.0, 0.0])
    return np.asarray(r, np.float32)


def dump_to_np(d):
    out = {}
    for k, v in d.items():
        a = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        out[k] = a.astype(np.uint8) if a.dtype == np.bool_ else a
    return out


def main():
    torch.manual_seed(0)
    hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"])
    if "--only-ref" in sys.argv:
        return main_ref(hyper)
    if "--only-npp" in sys.argv:
        return main_npp(hyper)

    # (i) tiny scene, full per-sample dump, white_bg on, no alpha mask
    arrs = tiny_scene()
    sc = TO.scene_from_arrays(arrs, **hyper)
    rays = tiny_rays()
    d = dump_to_np(TO.execute(sc, rays, white_bg=True, N_samples=TINY["N_samples"], dump=True))
    xyz_n = torch.tensor(d["xyz_norm"][d["app_mask"].astype(bool)])
    f, h = TO.compute_appfeature(sc, xyz_n, return_h=True)
    dirs = rays[:, None, 3:6].expand(-1, TINY["N_samples"], -1)[torch.tensor(d["app_mask"].astype(bool))]
    rgb_s, mlp_in = TO.mlp_render_fea(sc, dirs, f, return_in=True)
    np.savez_compressed(os.path.join(HERE, "tiny_dump.npz"), rays=rays.numpy(), step=np.float32(sc.stepSize.item()),
                        app_feature=f.numpy(), app_h=h.numpy(), mlp_in=mlp_in.numpy(), app_rgb=rgb_s.numpy(),
                        app_dirs=dirs.numpy(), app_xyz_norm=xyz_n.numpy(),
                        **{f"scene.{k}": v for k, v in arrs.items()}, **{f"out.{k}": v for k, v in d.items()})

    # (iii) edge cases: white_bg on/off x alpha mask on/off, jitter on the last variant
    arrs_a = tiny_scene(alpha=True)
    er = edge_rays(TINY["aabb"])
    all_rays = torch.cat([torch.tensor(er), rays[::5]])
    jit = np.random.default_rng(3).random(all_rays.shape[0]).astype(np.float32)
    edge = dict(rays=all_rays.numpy(), jitter=jit, alpha_volume=arrs_a["alpha_volume"], alpha_aabb=arrs_a["alpha_aabb"])
    for name, A, wb, j in (("wb1_am0", arrs, True, None), ("wb0_am0", arrs, False, None),
                           ("wb1_am1", arrs_a, True, None), ("wb0_am1_jit", arrs_a, False, jit)):
        scx = TO.scene_from_arrays(A, **hyper)
        dd = dump_to_np(TO.execute(scx, all_rays, white_bg=wb, N_samples=TINY["N_samples"], jitter=j, dump=True))
        for k in ("rgb_map", "depth_map", "acc_map", "t_min", "valid", "bbox_valid", "cell", "weight", "app_mask", "z_vals", "sigma"):
            edge[f"{name}.{k}"] = dd[k]
    np.savez_compressed(os.path.join(HERE, "tiny_edge.npz"), **edge)

    # (ii) config-1 scale: BASELINE.json configs[0] — 128^3, 64x64, 192 samples; inputs regenerated from the seed
    B = synthetic.SCENE_B
    arrs1 = synthetic.make_scene_arrays(B["gridSize"], B["aabb"])
    hyper1 = dict(synthetic.HYPER, near_far=B["near_far"], step_ratio=B["step_ratio"])
    sc1 = TO.scene_from_arrays(arrs1, **hyper1)
    M = R.sphere_poses(8, B["cam_radius"])[0]
    rays1 = R.frame_rays(M, B["img_wh"][1], B["img_wh"][0], B["camera_angle_x"])
    d1 = dump_to_np(TO.execute(sc1, rays1, white_bg=True, N_samples=B["N_samples"], dump=True))
    np.savez_compressed(os.path.join(HERE, "config1.npz"), rays=rays1.numpy(),
                        scene_sha=np.array([f"{k}:{sha(v)}" for k, v in sorted(arrs1.items())]),
                        step=np.float32(sc1.stepSize.item()), nSamples=sc1.nSamples,
                        rgb_map=d1["rgb_map"], depth_map=d1["depth_map"], acc_map=d1["acc_map"],
                        valid_bits=np.packbits(d1["valid"]), app_bits=np.packbits(d1["app_mask"]),
                        n_valid=int(d1["valid"].sum()), n_app=int(d1["app_mask"].sum()))
    main_ref(hyper)
    main_npp(hyper)
    for fn in ("tiny_dump.npz", "tiny_edge.npz", "config1.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")
    print("tiny: valid", d["valid"].sum(), "app", d["app_mask"].sum(), "acc range", d["acc_map"].min(), d["acc_map"].max())
    print("config1: valid", d1["valid"].sum(), "app", d1["app_mask"].sum(), "acc mean", d1["acc_map"].mean())


if __name__ == "__main__":
    main()
