"""One 800x800 frame of NerfPlusPlus (TensorVMSplit 300^3 foreground with explicit depths + the 512-sample background network), timed."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from jittor_myc_nerfs_amd import NerfPlusPlus, OctreeRender_trilinear_fast, synthetic
A, H = synthetic.SCENE_A, synthetic.HYPER
arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], npp=6.0)
m = NerfPlusPlus(arrs["aabb"], A["gridSize"], "cuda", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27, near_far=A["near_far"],
                 shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=H["density_shift"], distance_scale=H["distance_scale"],
                 rayMarch_weight_thres=H["rayMarch_weight_thres"], pos_pe=6, view_pe=2, fea_pe=2, featureC=128, step_ratio=A["step_ratio"],
                 fea2denseAct=H["fea2denseAct"])
m.load_arrays(arrs)
rays = bench.frames(A)[0].cuda()
S = A["N_samples"]
MODES = ("f32", "f16act", "f16")          # model.mlp_arith (include/tvr.h TVR_ARITH_*): products per k-step of the appearance network AND of the background network
ref_pic = None
for mode in MODES[1:] + MODES[:1]:        # (the default mode last: the detailed lines below are its)
    m.mlp_arith = mode
    with torch.no_grad():
        OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=S, white_bg=False)      # (the whole frame: its 24 GB of background temporaries are allocated here, not in the timed frame)
        torch.manual_seed(0); torch.cuda.synchronize(); t0 = time.perf_counter()
        pic = OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=S, white_bg=False)[0]
        torch.cuda.synchronize(); t_mode = time.perf_counter() - t0
        n = 65536
        u = torch.randn(n, 512, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        pts = torch.cat([u / u.norm(dim=-1, keepdim=True), torch.rand(n, 512, 1, device="cuda")], -1)
        v = rays[:n, 3:6] / rays[:n, 3:6].norm(dim=-1, keepdim=True)
        out = m._mlpnet(pts, v); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): m._mlpnet(pts, v)
        torch.cuda.synchronize(); t_bg = (time.perf_counter() - t0) / 3
    print(f"mlp_arith {mode:7s}: {t_mode * 1e3:6.1f} ms / frame; background network kernel {n * 512 / t_bg / 1e9:.2f} G samples/s ({t_bg * 1e3:.1f} ms per {n * 512 / 1e6:.1f} M samples)")
m.mlp_arith = "f32"
with torch.no_grad():
    OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=S, white_bg=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rgb, _, _, _, _ = OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=S, white_bg=False)
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    # the foreground alone (same call sequence without the background network)
    z = m._fg_depths(rays[:, :3], rays[:, 3:6], S)
    m._render_z(rays, z, S, 1e-4); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m._render_z(rays, z, S, 1e-4)
    torch.cuda.synchronize(); t_fg = (time.perf_counter() - t0) / 5
print(f"NerfPlusPlus 800x800 x {S} fg samples + 512 bg samples: {t_all * 1e3:.0f} ms / frame ({rays.shape[0] * S / t_all:.3e} fg ray-samples/s); "
      f"foreground kernels alone (tvr_render_z, one call) {t_fg * 1e3:.1f} ms; rgb range {float(rgb.min()):.3f}..{float(rgb.max()):.3f}")
# the background network kernel alone: 65 536 rays x 512 samples per call (tvr_mlpnet_forward)
with torch.no_grad():
    n = 65536
    u = torch.randn(n, 512, 3, device="cuda")
    pts = torch.cat([u / u.norm(dim=-1, keepdim=True), torch.rand(n, 512, 1, device="cuda")], -1)
    v = rays[:n, 3:6] / rays[:n, 3:6].norm(dim=-1, keepdim=True)
    m._mlpnet(pts, v); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m._mlpnet(pts, v)
    torch.cuda.synchronize(); t_bg = (time.perf_counter() - t0) / 3
    lam = m._render_z(rays, z, S, 1e-4)[2]
print(f"background network kernel: {n * 512 / t_bg / 1e9:.2f} G samples/s ({t_bg * 1e3:.1f} ms per {n * 512 / 1e6:.1f} M samples; a full frame has 327.7 M); "
      f"rays with bg_lambda > 0.1 on this scene: {float((lam > 0.1).float().mean()) * 100:.0f} %")
