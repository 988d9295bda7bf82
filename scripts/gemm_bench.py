"""GPU box: tvr_gemm_tn (dW = dY^T X) at the training step's shapes, ms per call (TVR_LIB_PATH selects a variant library)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jittor_myc_nerfs_amd import _lib as L
for M, Ka, Kb in ((356_123, 128, 150), (356_123, 128, 128), (700_000, 128, 150), (700_000, 128, 128), (356_123, 3, 128), (356_123, 27, 144), (2_100_000, 128, 128)):
    A = torch.randn((M, Ka), device="cuda"); B = torch.randn((M, Kb), device="cuda")
    out = torch.empty((Ka, Kb), device="cuda")
    sc = torch.empty(L.lib().tvr_gemm_tn_scratch_bytes(Ka, Kb, M), dtype=torch.uint8, device="cuda")
    f = lambda: L.check(L.lib().tvr_gemm_tn(A.data_ptr(), Ka, Ka, B.data_ptr(), Kb, Kb, M, out.data_ptr(), sc.data_ptr(), sc.numel(), None), "gemm")
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"gemm_tn {M} x {Ka} x {Kb}: {ms:.3f} ms  {2.0 * M * Ka * Kb / ms / 1e9:.1f} TFLOP/s  {(Ka + Kb) * 4.0 * M / ms / 1e6:.0f} GB/s")
