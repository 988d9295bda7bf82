#!/usr/bin/env python3
"""GPU box: gradients of a six-frequency scene's training step — fused step vs eager chain (HIP Linears) vs eager chain (torch matmul) vs the CPU oracle's autograd."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TINY  # noqa: E402
from jittor_myc_nerfs_amd import TensorVMSplit, synthetic, autograd_ops  # noqa: E402
from oracle import tensorf_oracle as TO  # noqa: E402


def main():
    vpe, fpe = int(sys.argv[1]), int(sys.argv[2])
    rep = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    hyper = dict(synthetic.HYPER, near_far=TINY["near_far"], step_ratio=TINY["step_ratio"], view_pe=vpe, fea_pe=fpe)
    arrs = synthetic.make_scene_arrays(TINY["gridSize"], TINY["aabb"], seed=5, view_pe=vpe, fea_pe=fpe)
    rays = dict(np.load(os.path.join(ROOT, "tests", "golden", "tiny_dump.npz")))["rays"]
    rays = np.concatenate([rays] * rep).copy()
    rays[:, :3] += 0.01 * np.random.default_rng(3).standard_normal((rays.shape[0], 3)).astype(np.float32)
    cw = np.random.default_rng(12).standard_normal((rays.shape[0], 3)).astype(np.float32)
    S = TINY["N_samples"]

    def model():
        m = TensorVMSplit(arrs["aabb"], [int(x) for x in arrs["gridSize"]], "cuda", density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27,
                          near_far=hyper["near_far"], shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=hyper["density_shift"],
                          distance_scale=hyper["distance_scale"], rayMarch_weight_thres=hyper["rayMarch_weight_thres"], pos_pe=6, view_pe=vpe, fea_pe=fpe,
                          featureC=128, step_ratio=hyper["step_ratio"], fea2denseAct=hyper["fea2denseAct"])
        m.load_arrays(arrs)
        m.eps_T = 0.0
        return m

    def run(static, min_rows=None):
        if min_rows is not None:
            autograd_ops._HIP_MM_MIN_ROWS = min_rows
        m = model()
        m.static_training = static
        rgb, _ = m.render_rays_autograd(torch.tensor(rays, device="cuda"), white_bg=True, N_samples=S)
        (rgb * torch.tensor(cw, device="cuda")).sum().backward()
        mlp = m.renderModule.mlp
        return {"rgb": rgb.detach().cpu(), "basis": m.basis_mat.weight.grad.cpu(), "W1": mlp[0].weight.grad.cpu(), "b1": mlp[0].bias.grad.cpu(), "W2": mlp[2].weight.grad.cpu(),
                "app_plane0": m.app_plane[0].grad.cpu(), "den_plane0": m.density_plane[0].grad.cpu()}
    default_rows = autograd_ops._HIP_MM_MIN_ROWS
    res = {"fused": run(True), "eager_hip": run(False, 1), "eager_torch": run(False, 1 << 40)}
    autograd_ops._HIP_MM_MIN_ROWS = default_rows
    from test_gpu_training import _oracle_with_grads
    sc, leaves = _oracle_with_grads(arrs, hyper)
    rgb_o, _ = TO.execute(sc, torch.tensor(rays), white_bg=True, N_samples=S)
    (rgb_o * torch.tensor(cw)).sum().backward()
    res["oracle"] = {"rgb": rgb_o.detach(), "basis": leaves["basis_mat"].grad, "W1": leaves["W1"].grad, "b1": leaves["b1"].grad, "W2": leaves["W2"].grad,
                     "app_plane0": leaves["app_plane.0"].grad, "den_plane0": leaves["density_plane.0"].grad}
    print(f"view_pe {vpe} fea_pe {fpe}  rays {rays.shape[0]}")
    for k in ("rgb", "basis", "W1", "b1", "W2", "app_plane0", "den_plane0"):
        o = res["oracle"][k]
        print(f"{k:>10}: " + "  ".join(f"{n} {float((res[n][k] - o).abs().max()) / max(float(o.abs().max()), 1e-9):.2e}" for n in ("fused", "eager_hip", "eager_torch")))
    if res["fused"]["W1"].shape[1] > 0:
        d = (res["fused"]["W1"] - res["oracle"]["W1"]).abs().max(0).values
        print("W1 columns with the largest error:", torch.topk(d, 8).indices.tolist(), "of", res["fused"]["W1"].shape[1])


if __name__ == "__main__":
    main()
