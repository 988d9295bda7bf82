#!/usr/bin/env python3
"""GPU box: rank 0's share of the 8-way split of the bench frame (80 384 rays) as a stream of frames through ShardedFramePipeline's RCCL branch — a ONE-member `nccl`
group: the streams an 8-GPU rank has (caller, render stream(s), the exchange's side stream, RCCL's own), no wire.  usage: share_rccl_timing.py [one|two] [steps] [N = 8] [piece rays: -1 = the library's default, 0 = one launch set per call]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from jittor_myc_nerfs_amd import ShardedFramePipeline, shard_indices  # noqa: E402


def main():
    two = (sys.argv[1] if len(sys.argv) > 1 else "two") == "two"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    piece_rays = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(36300 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    saved = os.dup(1)
    os.dup2(2, 1)                                         # (RCCL prints its banner on stdout)
    dist.init_process_group("nccl", device_id=dev)
    model, arrs, A = bench.build_model(dev, "TensorVMSplit")
    S = A["N_samples"]
    fr = [f.to(dev) for f in bench.frames(A)]
    if piece_rays >= 0:
        model.render_piece_rays = piece_rays
    idx = shard_indices(fr[0].shape[0], 0, N, 512).to(dev)
    subs = [f.index_select(0, idx).contiguous() for f in fr]
    n = subs[0].shape[0]
    pipe = ShardedFramePipeline(model, n, 0, 1, tile=512, white_bg=True, N_samples=S, exchange="dist", two_in_flight=two)
    want = model.render_rays(subs[3], white_bg=True, N_samples=S)
    for k in range(6):
        pipe.submit(k % 8, subs[k % 8])
    pipe.flush()
    torch.cuda.synchronize()
    best, all_ = 1e9, []
    for blk in range(5):
        t0 = time.perf_counter()
        got = None
        for k in range(steps):
            prev = pipe.submit(k % 8, subs[k % 8])
            if k == 4:
                got = (prev[0].clone(), prev[1].clone())  # frame 3
        pipe.flush()
        torch.cuda.synchronize()
        all_.append((time.perf_counter() - t0) / steps * 1e3)
        best = min(all_)
    ok = torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    os.dup2(saved, 1)
    print(f"{n} rays per share (N = {N}), pieces {'default' if piece_rays < 0 else piece_rays}, {'two shares' if pipe.two else 'one share'} in flight, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: "
          f"{best:.3f} ms per step (fastest of 5 blocks of {steps}; median {sorted(all_)[2]:.3f}); frame == plain render: {ok}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
