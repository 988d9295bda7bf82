"""Data-parallel training step for the reference's loop (tensorf-myc/train.py:219-261; SURVEY §8 f1: "adds an all-reduce of ≈69 MB
grads for DP training").  One process per GPU; every rank holds the whole field (70 MB) and renders its share of the ray batch; the
gradients of ALL parameters live in ONE contiguous fp32 bucket, so a step needs a single RCCL all-reduce (`backend="nccl"` on ROCm)
of ≈70 MB — 2·(N-1)/N·70 MB over the xGMI ring, about a millisecond at 8 GPUs — instead of one collective per tensor.

    bucket = GradBucket(tensorf)                     # after construction and after every upsample_volume_grid / shrink
    for it in range(n_iters):
        bucket.zero()
        rays, rgbs = my_share_of_the_batch(rank, world)
        rgb_map, *_ = OctreeRender_trilinear_fast(rays, tensorf, chunk=..., N_samples=nSamples, white_bg=True, is_train=True)
        loss = torch.mean((rgb_map - rgbs) ** 2) + regularisers
        loss.backward()
        bucket.all_reduce_mean()                     # no-op when torch.distributed is not initialised / world size 1
        optimizer.step()
The regularisers depend on the parameters only, so every rank computes the same value and the mean leaves their gradient unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class GradBucket:
    """Makes every parameter's `.grad` a view into one flat fp32 buffer (allocated on the parameters' device)."""

    def __init__(self, module: torch.nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("GradBucket needs all parameters in fp32 on one device")
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self) -> None:
        """Use instead of optimizer.zero_grad() (whose set_to_none default would detach the views)."""
        self.flat.zero_()

    def check(self) -> None:
        """Raises if something replaced a `.grad` (e.g. optimizer.zero_grad(set_to_none=True), or the parameters were re-created)."""
        off = 0
        for p in self.params:
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off:
                raise RuntimeError("a parameter's .grad no longer aliases the bucket: rebuild the GradBucket (after upsample_volume_grid / "
                                   "shrink) and clear gradients with bucket.zero()")
            off += p.numel()

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Average the gradients over the ranks of `group` with ONE all-reduce of the whole bucket."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        self.check()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / world)


def shard_batch(n: int, rank: int, world: int) -> slice:
    """Rank's contiguous share of an n-ray training batch (train.py draws the batch at random, so any fixed split is unbiased)."""
    per = (n + world - 1) // world
    return slice(min(rank * per, n), min((rank + 1) * per, n))


def make_graphed_step(step_fn, device=None, warmup: int = 2):
    """Capture one whole training step — forward, backward, optimizer — as a hipGraph, the safe way, and return a callable that replays it.

        opt = torch.optim.Adam(tensorf.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99), capturable=True, foreach=True)
        def step():                                   # static tensors in, a static loss tensor out
            opt.zero_grad(set_to_none=False)
            rgb, _ = tensorf.render_rays_autograd(rays, white_bg=True, N_samples=S, jitter=jitter)
            loss = torch.mean((rgb - target) ** 2)
            loss.backward(); opt.step()
            return loss
        replay = make_graphed_step(step)
        for it in range(n): rays.copy_(...); target.copy_(...); jitter.uniform_(); loss = replay()

    What this does that a bare `with torch.cuda.graph(g): step()` does not: the `warmup` eager steps run on a SIDE stream.  Autograd state created by an eager
    step on the default stream (the parameters' AccumulateGrad nodes) makes the capture wait on the legacy default stream; on this ROCm that ends in a
    segmentation fault inside hipStreamEndCapture (gpurun_out/r3k/fused.log).  The fused training forward refuses such a capture with a Python error instead
    (autograd_ops._FusedStepFn); this helper is how to do it right.  Requirements are PyTorch's own: the step touches only static tensors, the optimizer is
    `capturable=True`, nothing inside reads the host.  Returns `replay` with attributes `.graph` (the torch.cuda.CUDAGraph) and `.output` (step_fn's static result)."""
    if warmup < 1:
        raise ValueError("make_graphed_step needs at least one eager warm-up step (buffers, the packed scene and autograd state are created there)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step_fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    from . import field as _field
    graph = torch.cuda.CUDAGraph()
    _field.GRAPH_HELPER_CAPTURING[0] = True
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (other threads — a process group's watchdog — may call the runtime meanwhile)
            out = step_fn()
    finally:
        _field.GRAPH_HELPER_CAPTURING[0] = False

    def replay():
        graph.replay()
        _field.GRAPH_REPLAYS[0] += 1          # models whose tvr_scene_update lives in this graph re-pack on their next host-driven call, and only then
        return out
    replay.graph, replay.output = graph, out
    return replay
