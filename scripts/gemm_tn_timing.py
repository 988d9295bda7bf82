import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jittor_myc_nerfs_amd import _lib as L
for M, Ka, Kb in ((356000, 128, 150), (356000, 128, 128), (356000, 3, 128), (356000, 27, 144)):
    A = torch.randn((M, Ka), device="cuda"); B = torch.randn((M, Kb), device="cuda"); out = torch.empty((Ka, Kb), device="cuda")
    sc = torch.empty(L.lib().tvr_gemm_tn_scratch_bytes(Ka, Kb, M), dtype=torch.uint8, device="cuda")
    f = lambda: L.check(L.lib().tvr_gemm_tn(A.data_ptr(), Ka, Ka, B.data_ptr(), Kb, Kb, M, out.data_ptr(), sc.data_ptr(), sc.numel(), None), "g")
    g = lambda: A.t() @ B
    for fn, name in ((f, "tvr_gemm_tn"), (g, "library")):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{M}x{Ka}x{Kb} {name:12s} {dt * 1e3:.3f} ms  {2 * M * Ka * Kb / dt / 1e12:.1f} TFLOP/s")
