"""GPU box: the shade kernel's matrix phase alone (tvr_mlp_render: features given, no gather, no basis product) and its feature half
(tvr_app_feature), ms per call and cycles per 32-entry tile and SIMD (TVR_LIB_PATH selects a variant library)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
m, arrs, A = bench.build_model(torch.device("cuda"))
n = 4_000_000
xyz = (torch.rand(n, 3, device="cuda") * 2 - 1) * 1.4
vd = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=1)
feat = torch.randn(n, 27, device="cuda")
def timeit(name, fn, mfma):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles_per_simd = n / 32 / 1024
    print(f"{name:18s} {ms:7.3f} ms for {n} samples = {ms * 1e3 / tiles_per_simd:6.2f} us per tile and SIMD ({mfma} MFMAs = {mfma * 32} cycles of matrix pipe)")
timeit("mlp_render", lambda: m._mlp_render(vd, feat), 216)
timeit("app_feature", lambda: m.compute_appfeature(xyz), 27)
