"""world_size 2 over gloo: the N>1 path of render_sharded (tile interleave + one all_gather + one index gather) returns exactly the
single-process image.
  * CPU (not gpu): the per-rank renderer is the C oracle (a stand-in renderer: the host logic is what is under test);
  * -m gpu: the per-rank renderer is the HIP field (`render_rays` through the C-ABI), two spawned ranks sharing the box's one card —
    BASELINE configs[2] in miniature, and bench.py's own multi-rank step function (weak and strong mode) under TVR_BENCH_BACKEND=gloo."""
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


def _worker_cases(rank, world, port, arrs, hyper, rays_np, cases, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jittor_myc_nerfs_amd import render_sharded
    from oracle import c_oracle as CO, tensorf_oracle as TO
    sc = TO.scene_from_arrays(arrs, **hyper)
    co = CO.COracle(arrs, step=float(sc.stepSize), **hyper)

    def render_fn(r):
        if r.shape[0] == 0:                                          # a rank that owns no tile of this frame
            return torch.zeros((0, 3)), torch.zeros((0,))
        o = co.render(r.numpy(), TINY["N_samples"])
        return torch.from_numpy(o["rgb_map"]), torch.from_numpy(o["depth_map"])

    out = {}
    for n, tile in cases:
        rgb, depth = render_sharded(torch.from_numpy(rays_np[:n]), render_fn, rank, world, tile=tile)
        out[(n, tile)] = (rgb.numpy(), depth.numpy())
    out_q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_render_sharded_world8_ragged_tiles_equal_single(tiny_arrays, hyper_tiny, tiny_edge, tiny_dump):
    """EIGHT ranks as eight processes over gloo (the driver's N = 8 layout), host logic only — the per-rank renderer is the C oracle:
    ragged frames whose tiles do not divide among the ranks (20 tiles of 4 over 8 ranks with a short last tile; 9 tiles of 16: one rank renders two, the rest
    one; 2 tiles for 8 ranks: six ranks render nothing and still take part in the ONE all_gather) all come back as the single-process image, on every rank."""
    from oracle import c_oracle as CO, tensorf_oracle as TO
    rays = np.concatenate([tiny_edge["rays"], tiny_dump["rays"]] * 3)[:130]
    cases = [(77, 4), (130, 16), (5, 4), (64, 8)]
    sc = TO.scene_from_arrays(tiny_arrays, **hyper_tiny)
    single = CO.COracle(tiny_arrays, step=float(sc.stepSize), **hyper_tiny).render(rays, TINY["N_samples"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_cases, args=(r, 8, port, tiny_arrays, hyper_tiny, rays, cases, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(8))
    for rank, out in res:
        for (n, tile), (rgb, depth) in out.items():
            assert rgb.shape == (n, 3) and depth.shape == (n,)
            assert np.array_equal(rgb, single["rgb_map"][:n]), f"rank {rank}, {n} rays in tiles of {tile}: gathered image != single-process image"
            assert np.array_equal(depth, single["depth_map"][:n])


# ---- the HIP renderer behind render_sharded (BASELINE configs[2]) ------------------------------------------------------------
def _hip_worker(rank, world, port, arrs, hyper, rays_np, tiles, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)       # gloo stages device tensors through the host: both ranks share cuda:0
    from conftest import make_model as mk
    from jittor_myc_nerfs_amd import render_sharded
    m = mk(arrs, hyper)
    m.eps_T = 0.0
    rays = torch.tensor(rays_np, device="cuda")
    out = {}
    for tile in tiles:
        rgb, depth = render_sharded(rays, lambda r, out=None: m.render_rays(r, white_bg=True, N_samples=TINY["N_samples"], out=out), rank, world, tile=tile)
        out[tile] = (rgb.cpu().numpy(), depth.cpu().numpy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_render_sharded_world2_hip_renderer_equals_single(tiny_arrays, hyper_tiny, tiny_dump, tiny_edge):
    from conftest import make_model
    rng = np.random.default_rng(5)
    base = np.concatenate([tiny_dump["rays"], tiny_edge["rays"]])
    rays = np.concatenate([base] * 140)[:9001].copy()                  # ragged: 9001 rays -> 563 tiles of 16 (last one short), 3 tiles of 4096
    rays[:, :3] += 0.02 * rng.standard_normal((rays.shape[0], 3)).astype(np.float32)
    m = make_model(tiny_arrays, hyper_tiny)
    m.eps_T = 0.0
    rgb1, depth1 = m.render_rays(torch.tensor(rays, device="cuda"), white_bg=True, N_samples=TINY["N_samples"])
    rgb1, depth1 = rgb1.cpu().numpy(), depth1.cpu().numpy()
    assert rgb1.std() > 0.01                                            # a picture, not a constant
    ctx = mp.get_context("spawn")                                       # fresh child processes: no re-exec of a GPU-initialised one
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, tiny_arrays, hyper_tiny, rays, (16, 4096), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, out in res:
        for tile, (rgb, depth) in out.items():
            assert np.array_equal(rgb, rgb1), f"rank {rank}, tile {tile}: gathered HIP image != single-process HIP image"
            assert np.array_equal(depth, depth1)


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("scaling,img", [("weak", 96), ("strong", 96), ("strong", 800), ("default", 96)])
def test_bench_step_world2_on_one_card(scaling, img, tmp_path):
    """bench.py's multi-rank path exactly as the driver launches it (torch.distributed.run, 2 ranks) except for the rehearsal backend (and, for
    three of the cases, a small frame): `--check` makes every rank compare the gathered frame(s) with its own single-rank render of the same
    rays, bit for bit.  ("strong", 800) is BASELINE configs[2]'s own frame: the 640 000 rays of scene A split in two.  "default" passes no
    --scaling: N > 1 must then be the strong split of ONE frame (SURVEY 8e's metric), not N frames."""
    import json
    import subprocess
    env = dict(os.environ, TVR_BENCH_BACKEND="gloo", TVR_BENCH_IMG=str(img), HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 35500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--check", "--no-cpu-baseline"]
    if scaling != "default":
        cmd += ["--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    want = "strong" if scaling == "default" else scaling
    assert d["n_gpus"] == 2 and d["scaling"] == want and d["check"] == "gathered == single-rank render, bit for bit"
    assert d["config"]["rays_per_step"] == (img * img * (2 if want == "weak" else 1))
    if want == "strong":
        ss = d["strong_split"]
        # (a rehearsal: two ranks on ONE card over host-staged gloo — the ratio is only checked to be a number, it is no scaling figure)
        assert ss["N"] == 2 and ss["t1_ms"] > 0 and ss["tN_ms"] > 0 and 0.0 < ss["t1_over_N_tN"] < 1.5
        assert d["config"]["rays_per_rank"] <= (img * img + 4095) // 4096 // 2 * 4096 + 4096


# ---- the RCCL backend itself, as far as ONE card can run it: a one-member `nccl` process group -----------------------------------------
def _rccl_worker(port, arrs, hyper, rays_np, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from conftest import make_model as mk
    from jittor_myc_nerfs_amd import render_sharded
    from jittor_myc_nerfs_amd.training import GradBucket
    m = mk(arrs, hyper)
    m.eps_T = 0.0
    rays = torch.tensor(rays_np, device="cuda")
    rgb1, depth1 = m.render_rays(rays, white_bg=True, N_samples=TINY["N_samples"])
    out = {"backend": dist.get_backend()}
    for tile in (16, 4096):
        rgb, depth = render_sharded(rays, lambda r, out=None: m.render_rays(r, white_bg=True, N_samples=TINY["N_samples"], out=out), 0, 1, tile=tile,
                                    exchange_at_world1=True)
        out[tile] = bool(torch.equal(rgb, rgb1) and torch.equal(depth, depth1))
    # the frame STREAM (ShardedFramePipeline): the exchange of frame k on a side stream behind frame k + 1's kernels, frames returned one submit late, with and
    # without the hipGraph capture of the per-rank render — every frame bit for bit the plain render
    from jittor_myc_nerfs_amd import ShardedFramePipeline, shard_indices
    sets = [rays, rays.flip(0).contiguous(), torch.roll(rays, 7, 0).contiguous()]
    plain = [m.render_rays(r, white_bg=True, N_samples=TINY["N_samples"]) for r in sets]
    # (round 6: optionally two shares in flight on two render streams with their own scratch slots; the default is the one-stream form)
    for graph, two in ((False, True), (False, False), (True, False)):
        pipe = ShardedFramePipeline(m, rays.shape[0], 0, 1, tile=16, white_bg=True, N_samples=TINY["N_samples"], exchange="dist", graph=graph, two_in_flight=two)
        assert pipe.two == two
        subs = [r.index_select(0, shard_indices(r.shape[0], 0, 1, 16).cuda()).contiguous() for r in sets]
        ok, got = True, []
        for i in range(7):                                 # more frames than buffers: both send buffers come round several times
            prev = pipe.submit(i % 3, subs[i % 3])
            if i == 0:
                ok = ok and prev is None
            else:
                got.append((i - 1, prev[0].clone(), prev[1].clone()))
        last = pipe.flush()
        got.append((6, last[0].clone(), last[1].clone()))
        torch.cuda.synchronize()
        for i, rgb_i, depth_i in got:
            ok = ok and bool(torch.equal(rgb_i, plain[i % 3][0]) and torch.equal(depth_i, plain[i % 3][1]))
        out["pipeline_graph" if graph else ("pipeline" if two else "pipeline_one_in_flight")] = ok
    # the training side's one collective: the flat gradient bucket through an RCCL all_reduce (a one-member SUM returns its input)
    b = GradBucket(m)
    b.flat.copy_(torch.arange(b.numel, device="cuda", dtype=torch.float32) % 97)
    want = b.flat.clone()
    dist.all_reduce(b.flat, op=dist.ReduceOp.SUM)
    b.all_reduce_mean()
    out["bucket"] = bool(torch.equal(b.flat, want))
    torch.cuda.synchronize()
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_render_sharded_over_rccl_with_one_rank(tiny_arrays, hyper_tiny, tiny_dump, tiny_edge):
    """`backend="nccl"` IS RCCL on ROCm.  Two ranks cannot share a card under RCCL, so the branch the 8-GPU run takes (device-side
    all_gather_into_tensor on the send buffer, no host staging) is run here with ONE rank: same code, a one-member communicator."""
    base = np.concatenate([tiny_dump["rays"], tiny_edge["rays"]])
    rays = np.concatenate([base] * 140)[:9001].copy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(33500 + os.getpid() % 2000, tiny_arrays, hyper_tiny, rays, q))
    p.start()
    out = q.get(timeout=400)
    p.join(120)
    assert p.exitcode == 0
    assert out["backend"] == "nccl" and out[16] and out[4096] and out["bucket"], out
    assert out["pipeline"] and out["pipeline_one_in_flight"] and out["pipeline_graph"], out


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_exchange_over_rccl_with_one_rank():
    """bench.py's N > 1 step (send buffer, RCCL all_gather_into_tensor, un-permute, barriers, max-over-ranks) on the full 800x800 frame with a
    one-member `nccl` group, `--check`ed bit for bit against the plain render."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TVR_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--one-rank-exchange", "--check", "--no-cpu-baseline", "--pmc", "off"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["exchange_backend"].startswith("nccl") and d["check"] == "gathered == single-rank render, bit for bit"
    assert d["n_gpus"] == 1 and d["config"]["rays_per_step"] == 640000 and d["strong_split"]["N"] == 1


def test_shard_gather_index_is_the_inverse_of_the_tile_interleave():
    """CPU: image[i] = gathered[inv[i]] for every (rays, world, tile) — ragged last tiles, more ranks than tiles, one rank; and the strided
    un-permute of the [rgb block | depth block] send buffers (shard_send_views / shard_unpermute) returns the same frame."""
    from jittor_myc_nerfs_amd import shard_capacity, shard_gather_index, shard_indices, shard_send_views, shard_unpermute
    for R, w, t in [(77, 2, 16), (9001, 2, 4096), (640000, 8, 4096), (640000, 3, 4096), (5, 4, 16), (100, 1, 16), (1, 2, 4096)]:
        cap = shard_capacity(R, w, t)
        frame_rgb, frame_depth = torch.arange(R * 3, dtype=torch.float32).view(R, 3), -torch.arange(R, dtype=torch.float32)
        bufs = []
        for r in range(w):
            idx = shard_indices(R, r, w, t)
            buf = torch.full((4 * cap,), float("nan"))
            v_rgb, v_depth = shard_send_views(buf, cap, idx.numel())
            v_rgb.copy_(frame_rgb[idx])
            v_depth.copy_(frame_depth[idx])
            bufs.append(buf)
        rgb, depth = shard_unpermute(torch.cat(bufs), R, w, cap, t)
        assert torch.equal(rgb, frame_rgb) and torch.equal(depth, frame_depth), (R, w, t)
    for R, w, t in [(77, 2, 16), (640000, 8, 4096), (640000, 3, 4096), (5, 4, 16), (4096 * 8, 8, 4096), (100, 1, 16), (1, 2, 4096)]:
        cap = shard_capacity(R, w, t)
        g = torch.full((w * cap,), -1, dtype=torch.long)
        for r in range(w):
            idx = shard_indices(R, r, w, t)
            assert idx.numel() <= cap
            g[r * cap:r * cap + idx.numel()] = idx
        assert torch.equal(g[shard_gather_index(R, w, t)], torch.arange(R)), (R, w, t)
