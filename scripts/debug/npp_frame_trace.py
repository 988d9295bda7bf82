"""One NerfPlusPlus frame (f32) rendered 3 times; run under rocprofv3 --kernel-trace --stats to see what a frame is made of."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from jittor_myc_nerfs_amd import NerfPlusPlus, OctreeRender_trilinear_fast, synthetic
A, H = synthetic.SCENE_A, synthetic.HYPER
arrs = synthetic.make_scene_arrays(A["gridSize"], A["aabb"], npp=6.0)
m = NerfPlusPlus(arrs["aabb"], A["gridSize"], "cuda", density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27, near_far=A["near_far"],
                 shadingMode="MLP_Fea", alphaMask_thres=1e-4, density_shift=H["density_shift"], distance_scale=H["distance_scale"],
                 rayMarch_weight_thres=H["rayMarch_weight_thres"], pos_pe=6, view_pe=2, fea_pe=2, featureC=128, step_ratio=A["step_ratio"], fea2denseAct=H["fea2denseAct"])
m.load_arrays(arrs)
rays = bench.frames(A)[0].cuda()
with torch.no_grad():
    for _ in range(3):
        OctreeRender_trilinear_fast(rays, m, chunk=4096, N_samples=A["N_samples"], white_bg=False)
torch.cuda.synchronize()
print("done")
