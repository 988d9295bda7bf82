#!/usr/bin/env python3
"""Round 6: does a rank's share of the N-way split gain from two shares in flight (render.FrameStream: share k on stream k % 2) the way whole frames do?
   python scripts/share_stream_experiment.py  ->  ms per share, serial loop vs two in flight, N = 2, 4, 8 (rank 0's 512-ray tiles of the 8 bench poses)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from jittor_myc_nerfs_amd import FrameStream, shard_indices  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model, arrs, A = bench.build_model(dev)
    S = A["N_samples"]
    fr = bench.frames(A)
    for N in (2, 4, 8):
        idx = shard_indices(fr[0].shape[0], 0, N, 512)
        shares = [f[idx].contiguous().to(dev) for f in fr]
        n = shares[0].shape[0]
        out = [(torch.empty((n, 3), device=dev), torch.empty((n,), device=dev)) for _ in range(2)]
        res = {}
        for rnd in range(3):
            for name in ("serial", "two in flight"):
                fs = FrameStream(model, white_bg=True, N_samples=S)
                def run(k):
                    if name == "serial":
                        for i in range(k):
                            model.render_rays(shares[i % 8], white_bg=True, N_samples=S, out=out[i % 2])
                    else:
                        for i in range(k):
                            fs.submit(shares[i % 8], out=out[i % 2])
                        fs.flush()
                run(8)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(64)
                torch.cuda.synchronize()
                res.setdefault(name, []).append((time.perf_counter() - t0) / 64 * 1e3)
        print(f"N = {N}: {n} rays per share; serial {min(res['serial']):.3f} ms, two in flight {min(res['two in flight']):.3f} ms ({100 * (min(res['two in flight']) / min(res['serial']) - 1):+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
