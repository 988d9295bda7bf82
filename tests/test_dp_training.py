"""Data-parallel training: GradBucket host logic on CPU (world_size 2 over gloo), and on the GPU the property that
makes DP correct — the rank-averaged gradient of two half batches equals the single-process gradient of the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, TINY, make_model


def _cpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jittor_myc_nerfs_amd import GradBucket
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b = GradBucket(net)
    x = torch.arange(40, dtype=torch.float32).view(8, 5) / 10 + rank           # different data per rank
    b.zero()
    net(x).square().mean().backward()
    local = b.flat.clone()
    b.all_reduce_mean()
    q.put((rank, local.numpy(), b.flat.numpy().copy(), [p.grad.data_ptr() - b.flat.data_ptr() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_bucket_one_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    mean = (res[0][1] + res[1][1]) / 2
    for rank, local, reduced, offs in res:
        assert np.allclose(reduced, mean, atol=1e-7) and not np.allclose(local, mean)
        assert offs == [0, 35 * 4, 42 * 4, 63 * 4]                                # grads are views into the one bucket, in order


def test_grad_bucket_guards():
    from jittor_myc_nerfs_amd import GradBucket, shard_batch
    net = torch.nn.Linear(3, 2)
    b = GradBucket(net)
    b.check()
    b.all_reduce_mean()                                                           # no process group: a no-op
    net.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        b.check()
    assert [shard_batch(10, r, 4) for r in range(4)] == [slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 10)]
    assert shard_batch(2, 3, 4) == slice(2, 2)


def _gpu_worker(rank, world, port, arrs, hyper, rays_np, cw_np, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)                  # gloo stages the CUDA bucket through the host: both
    from conftest import make_model as mk                                          # ranks share the box's single GPU here
    from jittor_myc_nerfs_amd import GradBucket, shard_batch
    m = mk(arrs, hyper)
    m.eps_T = 0.0
    b = GradBucket(m)
    sl = shard_batch(rays_np.shape[0], rank, world)
    rays, cw = torch.tensor(rays_np[sl], device="cuda"), torch.tensor(cw_np[sl], device="cuda")
    b.zero()
    rgb, _ = m.render_rays_autograd(rays, white_bg=True, N_samples=TINY["N_samples"])
    ((rgb * cw).sum(-1).mean() + 1e-3 * m.density_L1()).backward()
    b.all_reduce_mean()
    q.put((rank, b.flat.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_dp_gradient_equals_full_batch_gradient(tiny_dump, tiny_arrays, hyper_tiny):
    from jittor_myc_nerfs_amd import GradBucket
    rays_np = tiny_dump["rays"]
    cw_np = np.random.default_rng(3).standard_normal((rays_np.shape[0], 3)).astype(np.float32)
    m = make_model(tiny_arrays, hyper_tiny)
    m.eps_T = 0.0
    b = GradBucket(m)
    b.zero()
    rgb, _ = m.render_rays_autograd(torch.tensor(rays_np, device="cuda"), white_bg=True, N_samples=TINY["N_samples"])
    ((rgb * torch.tensor(cw_np, device="cuda")).sum(-1).mean() + 1e-3 * m.density_L1()).backward()
    full = b.flat.cpu().numpy()
    assert b.numel == sum(p.numel() for p in m.parameters()) and np.abs(full).max() > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, tiny_arrays, hyper_tiny, rays_np, cw_np, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])                                   # every rank ends with the same gradient
    err = np.abs(res[0][1] - full).max() / np.abs(full).max()
    print(f"DP (2 ranks, half batches, one all-reduce) vs full batch: rel max err {err:.2e}")
    assert err < 2e-5                                                             # fp32 atomics order + summation order only
