"""CPU, world_size 2 over gloo: the N>1 path of render_sharded (tile interleave + one all_gather + un-permute)
returns exactly the single-process image.  The per-rank renderer here is the C oracle (tests may use it as a
stand-in renderer; the product path passes the HIP field's render_rays)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, TINY


def _worker(rank, world, port, arrs, hyper, rays_np, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jittor_myc_nerfs_amd import render_sharded
    from oracle import c_oracle as CO, tensorf_oracle as TO
    sc = TO.scene_from_arrays(arrs, **hyper)
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)

    def render_fn(r):
        o = co.render(r.numpy(), TINY["N_samples"])
        return torch.from_numpy(o["rgb_map"]), torch.from_numpy(o["depth_map"])

    rgb, depth = render_sharded(torch.from_numpy(rays_np), render_fn, rank, world, tile=16)
    out_q.put((rank, rgb.numpy(), depth.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_render_sharded_world2_equals_single(tiny_arrays, hyper_tiny, tiny_edge):
    from oracle import c_oracle as CO, tensorf_oracle as TO
    rays = np.concatenate([tiny_edge["rays"]] * 4)[:77]            # ragged: 77 rays, tile 16 -> 5 tiles, last one short
    sc = TO.scene_from_arrays(tiny_arrays, **hyper_tiny)
    single = CO.COracle(tiny_arrays, step=float(sc.stepSize), **hyper_tiny).render(rays, TINY["N_samples"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tiny_arrays, hyper_tiny, rays, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, rgb, depth in res:
        assert np.array_equal(rgb, single["rgb_map"]), f"rank {rank}: gathered image != single-process image"
        assert np.array_equal(depth, single["depth_map"])
