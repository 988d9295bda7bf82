"""Ray-batch render loop and its multi-GPU sharding.

  * OctreeRender_trilinear_fast — same signature and return tuple as tensorf-myc/renderer.py:12-27
    (`(rgb [R,3], None, depth [R], None, None)`); each chunk is ONE tvr_render call (three kernels) on the current
    stream, with no host synchronisation between chunks (the reference syncs and garbage-collects per chunk, :23-25).
  * render_sharded — new functionality (the reference is single-GPU): rays are cut into fixed-size
    tiles dealt round-robin to the ranks (interleaving evens out empty-vs-dense image regions), each rank renders
    its tiles straight into its all_gather send buffer, and ONE all_gather (RCCL over xGMI when the backend is "nccl") returns every
    pixel to every rank; two strided copies undo the interleave.
    Per-ray results do not depend on the partition, so the gathered image equals the single-GPU image bit for bit.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


SHARD_TILE = 512                        # rays per shard tile, dealt round-robin to the ranks.  Round 4 (scripts/shard_balance.py, the 8 bench poses, 8 ranks): the slowest
                                        # rank carries 1.020 x the mean march work with 4096-ray tiles (1.024 x the rays), 1.0045 x with 512; the kernels lose nothing
                                        # to the shorter runs of consecutive rays (scripts/emulate8_ab.sh)
_INFER_SCRATCH_BUDGET = 16 << 30        # bytes of queue scratch an inference call may use (40 B per ray-sample, worst case)


def OctreeRender_trilinear_fast(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True, is_train=False,
                                device='cuda'):
    """tensorf-myc/renderer.py:12-27.  `chunk` bounds memory in the reference (every chunk materialises [chunk,S,3] tensors and
    is followed by a host sync, :23-25).  Here a chunk is one enqueue of three kernels with no host sync, and per-ray results
    do not depend on how rays are batched (tested bit for bit), so inference merges chunks up to a scratch budget — the
    reference's evaluation() passes chunk=1024 (625 calls per 800x800 frame), which would otherwise be host-bound.  Training
    calls (is_train under autograd) keep the caller's chunk: there it is the optimisation batch."""
    N_rays_all = rays.shape[0]
    if not (is_train and torch.is_grad_enabled()) and N_rays_all > chunk:
        S = N_samples if N_samples > 0 else getattr(tensorf, "nSamples", 1024)
        cap = min(_INFER_SCRATCH_BUDGET // (40 * S), ((1 << 32) - 1) // S)
        cap = min(cap, getattr(tensorf, "max_render_chunk", cap))      # e.g. NerfPlusPlus: its background geometry holds [chunk,512,*] tensors
        chunk = max(chunk, min(N_rays_all, (cap // 4096) * 4096 if cap >= 4096 else cap))
    rgbs, depth_maps = [], []
    for chunk_idx in range(N_rays_all // chunk + int(N_rays_all % chunk > 0)):
        rays_chunk = rays[chunk_idx * chunk:(chunk_idx + 1) * chunk]
        rgb_map, depth_map = tensorf(rays_chunk, is_train=is_train, white_bg=white_bg, ndc_ray=ndc_ray, N_samples=N_samples)
        rgbs.append(rgb_map)
        depth_maps.append(depth_map)
    return torch.cat(rgbs), None, torch.cat(depth_maps), None, None


def N_to_reso(n_voxels, bbox):
    """tensorf-myc/utils.py:56-59: grid resolution for a voxel budget inside an aabb."""
    xyz_min, xyz_max = bbox
    xyz_min, xyz_max = torch.as_tensor(xyz_min, dtype=torch.float32), torch.as_tensor(xyz_max, dtype=torch.float32)
    voxel_size = ((xyz_max - xyz_min).prod() / n_voxels).pow(1 / 3)
    return ((xyz_max - xyz_min) / voxel_size).long().tolist()


def cal_n_samples(reso, step_ratio=0.5):
    """tensorf-myc/utils.py:61-62."""
    import numpy as np
    return int(np.linalg.norm(reso) / step_ratio)


def shard_indices(n_rays: int, rank: int, world: int, tile: int = SHARD_TILE) -> torch.Tensor:
    """Indices of the rays rank `rank` renders: tiles rank, rank+world, rank+2*world, ... of `tile` rays each."""
    n_tiles = (n_rays + tile - 1) // tile
    if rank >= n_tiles:
        return torch.zeros(0, dtype=torch.long)
    mine = torch.arange(rank, n_tiles, world)
    idx = (mine[:, None] * tile + torch.arange(tile)[None, :]).reshape(-1)
    return idx[idx < n_rays]


def shard_capacity(n_rays: int, world: int, tile: int = SHARD_TILE) -> int:
    """Rays in the largest shard (rank 0's); every rank pads to this so one equal-size all_gather suffices."""
    n_tiles = (n_rays + tile - 1) // tile
    return ((n_tiles + world - 1) // world) * tile


_GATHER_INDEX_CACHE = {}


def shard_gather_index(n_rays: int, world: int, tile: int = SHARD_TILE, device=None) -> torch.Tensor:
    """inv [n_rays] with  image[i] = gathered[inv[i]]  for the all_gather layout (rank r's rows at r*cap .. r*cap + its ray count):
    ray i lies in tile t = i // tile, which rank t % world renders as its (t // world)-th tile.  The un-permute after the
    all_gather is then ONE index_select, whatever the world size (cached per shape and device)."""
    key = (n_rays, world, tile, str(device))
    inv = _GATHER_INDEX_CACHE.get(key)
    if inv is None:
        cap = shard_capacity(n_rays, world, tile)
        i = torch.arange(n_rays)
        t = i // tile
        inv = (t % world) * cap + (t // world) * tile + (i % tile)
        if device is not None:
            inv = inv.to(device)
        if len(_GATHER_INDEX_CACHE) > 16:
            _GATHER_INDEX_CACHE.clear()
        _GATHER_INDEX_CACHE[key] = inv
    return inv


def shard_send_views(buf: torch.Tensor, cap: int, n_mine: int):
    """Views of a flat [4 cap] fp32 send buffer: (rgb [n_mine,3] inside its first 3 cap floats, depth [n_mine] inside its last cap).  The render
    kernels write the pixels straight into them (render_rays(out=...)): no pad copy in front of the all_gather."""
    return buf[:3 * cap].view(cap, 3)[:n_mine], buf[3 * cap:][:n_mine]


def shard_unpermute(gathered: torch.Tensor, n_rays: int, world: int, cap: int, tile: int = SHARD_TILE):
    """gathered [world, 4 cap] (every rank's send buffer, rank-major) -> (rgb [n_rays,3], depth [n_rays]) in ray order.  Rank r's t-th tile is
    tile t * world + r of the frame, so the frame is the [tiles-per-rank, world] transpose of the gathered [world, tiles-per-rank] tile grid:
    two strided copies (rgb, depth), whatever the world size; the padding tiles land behind the last ray and are cut off."""
    L = cap // tile
    g = gathered.view(world, 4 * cap)
    rgb = g[:, :3 * cap].reshape(world, L, tile, 3).transpose(0, 1).reshape(-1, 3)[:n_rays]
    depth = g[:, 3 * cap:].reshape(world, L, tile).transpose(0, 1).reshape(-1)[:n_rays]
    return rgb, depth


def render_sharded(rays: torch.Tensor, render_fn: Callable[..., Tuple[torch.Tensor, torch.Tensor]],
                   rank: int, world: int, tile: int = SHARD_TILE, group=None, exchange_at_world1: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Render `rays` [R,6] (the full batch, present on every rank) across `world` ranks.

    render_fn(rays_subset, out=(rgb [n,3], depth [n])) renders on this rank's device INTO `out` (field.render_rays does; a render_fn without
    an `out` parameter is called plainly and its result copied).  Returns the full (rgb [R,3], depth [R]) on every rank after ONE all_gather of
    the [4 cap] fp32 send buffers (rgb block, then depth block) and two strided copies that undo the tile interleave (shard_unpermute).
    world == 1 renders plainly — unless `exchange_at_world1` (a rehearsal: send buffer, all_gather and un-permute run on a one-member group,
    which is how a single card exercises the RCCL branch)."""
    import inspect
    import torch.distributed as dist
    R = rays.shape[0]
    if world == 1 and not exchange_at_world1:
        return render_fn(rays)
    idx = shard_indices(R, rank, world, tile).to(rays.device)
    cap = shard_capacity(R, world, tile)
    mine = rays.new_zeros((4 * cap,))
    if idx.numel():
        views = shard_send_views(mine, cap, idx.numel())
        sub = rays.index_select(0, idx)
        if "out" in inspect.signature(render_fn).parameters:
            render_fn(sub, out=views)
        else:
            rgb, depth = render_fn(sub)
            views[0].copy_(rgb)
            views[1].copy_(depth)
    gathered = rays.new_empty((world * 4 * cap,))
    if dist.get_backend(group) == "gloo" and gathered.is_cuda:          # 1-GPU rehearsals: gloo has no all_gather_into_tensor for device tensors
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        gathered = torch.cat(parts)
    else:
        dist.all_gather_into_tensor(gathered, mine, group=group)
    rgb, depth = shard_unpermute(gathered, R, world, cap, tile)
    return rgb.contiguous(), depth.contiguous()


class FrameStream:
    """A stream of frames on ONE card with two frames in flight (round 5).  Frame k is rendered on stream k % 2 into its own scratch and output buffers; the four launches of
    a frame stay in order on their stream, and the persistent kernels of the OTHER frame fill the CUs this frame's kernels leave as they drain (and the chip's power budget
    while this frame is in its march).  Measured (scripts/overlap_stream.py, profiles/r05_overlap_stream.txt): 18.8 ms per frame against 19.7 for the serial loop on the
    bench frame, -4.4 %; limiting the kernels' grids to disjoint CU sets is WORSE (19.8 - 31.5 ms).  Pixels are those of render_rays, bit for bit (the kernels and their
    per-ray orders are the same; tests/test_gpu_parity.py::test_frame_stream_two_in_flight).  `submit()` returns the frame submitted one call earlier (None at the first
    call), `flush()` the last one: the returned tensors stay valid until the same slot is submitted again, two calls later.  A throughput device: a single frame's latency
    is not shortened."""

    def __init__(self, model, white_bg: bool = True, N_samples: int = -1, eps_T=None):
        if not getattr(model, "render_rays_is_the_frame", False):
            raise TypeError(f"FrameStream renders through model.render_rays, which is the whole frame for TensorVMSplit / REFTensoRF scenes only; "
                            f"{type(model).__name__} composes its picture in forward() (NerfPlusPlus: the background network) — render such frames one after the other")
        self.model, self.white_bg, self.S, self.eps_T = model, white_bg, N_samples, eps_T
        self.dev = model.device
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(2)]
        self.done = [torch.cuda.Event() for _ in range(2)]
        self.out = [None, None]
        self._own = [None, None]
        self.busy = [False, False]
        self.k = 0
        self._drain_next = False

    def submit(self, rays, out=None):
        """`out` (optional): (rgb [n,3], depth [n]) to render into — e.g. slices of one frame-sized pair when the submits are the chunks of a frame (a chunk loop with two
        calls in flight); the pair is what a later submit() / flush() hands back for this call.  Without it the slot's own buffers are used and come round again two submits later."""
        b = self.k % 2
        self.k += 1
        cur = torch.cuda.current_stream(self.dev)
        n = rays.shape[0]
        if out is not None:
            self.out[b] = out
        elif self.out[b] is None or self.out[b][0].shape[0] != n or self._own[b] is not self.out[b]:
            self.out[b] = self._own[b] = (torch.empty((n, 3), dtype=torch.float32, device=self.dev), torch.empty((n,), dtype=torch.float32, device=self.dev))
        # Two frames share the scene's packed images.  A call that will re-pack them (parameters changed), convert the fp16 copies, settle the range check or run the
        # arithmetic gate's probe renders must not overlap a frame that reads them: it waits for everything in flight, and so does the frame after it.
        settled = bool(getattr(self.model, "scene_settled", lambda: False)())
        if not settled or self._drain_next:
            cur.wait_stream(self.streams[0])
            cur.wait_stream(self.streams[1])
        self._drain_next = not settled
        st = self.streams[b]
        st.wait_stream(cur)                        # the rays are ready, and the caller's stream has been handed this slot's previous frame (two calls ago)
        with torch.cuda.stream(st):
            self.model.render_rays(rays, white_bg=self.white_bg, N_samples=self.S, eps_T=self.eps_T, out=self.out[b], scratch_slot=b)
            self.done[b].record(st)
        self.busy[b] = True
        prev = None
        if self.busy[b ^ 1]:                       # hand out the frame submitted one call earlier.  The caller's stream waits for it only AFTER this frame has been
            cur.wait_event(self.done[b ^ 1])       # queued behind the caller's earlier work: waiting first would put every frame behind the one before it
            prev = self.out[b ^ 1]
        return prev

    def flush(self):
        b = (self.k - 1) % 2
        if self.k == 0 or not self.busy[b]:
            return None
        torch.cuda.current_stream(self.dev).wait_event(self.done[b])
        if self.busy[b ^ 1]:
            torch.cuda.current_stream(self.dev).wait_event(self.done[b ^ 1])
        self.busy = [False, False]
        return self.out[b]


class ShardedFramePipeline:
    """A stream of frames, each split over the ranks (render_sharded's layout), with the two things a rank's 2.6 ms share of an 800x800 frame needs at N = 8:

      * the exchange of frame k (ONE all_gather of the [4 cap] send buffers + the two strided un-permute copies) runs on a SIDE stream behind frame k + 1's
        kernels: two send / receive buffer pairs, events in both directions.  `submit()` therefore returns the frame submitted one call EARLIER (None at
        the first call); `flush()` returns the last one;
      * optionally (`graph=True`) the per-rank render (header clear, march, shade, composite: four launches) is captured ONCE per distinct ray set as a hipGraph
        and replayed.  Measured on one MI355X (scripts/graph_cost.py, scripts/emulate8_ab.sh; profiles/r04_split_step_ab.txt): the four plain launches of a
        rank's 2.6 ms share already run back to back without gaps (2.538 ms per call against 2.541 ms of kernels), a replay takes 2.51 - 2.54 ms — no gain, hence
        off by default.  (The "0.12 ms of launch gaps" of round 3's emulation were the statistics atomics and clock probes of the timed launches.)

    Pixels are those of render_sharded — bit for bit the single-rank frame (tests/test_sharding_gloo.py).  `exchange`: "dist" (torch.distributed group),
    "local" (single process: this rank's buffer is copied into slot 0 of an [N, 4 cap] receive buffer and un-permuted — the device-side cost of the exchange
    at N-way sizes, scripts/strong_emulation.py) or None (render only)."""

    def __init__(self, model, n_rays: int, rank: int, world: int, tile: int = SHARD_TILE, white_bg: bool = True, N_samples: int = -1, eps_T=None,
                 group=None, exchange: Optional[str] = "dist", graph: bool = False, two_in_flight: bool = False):
        import torch.distributed as dist
        self.model, self.R, self.rank, self.world, self.tile = model, int(n_rays), rank, world, tile
        self.white_bg, self.S, self.eps_T, self.group, self.exchange, self.use_graph = white_bg, N_samples, eps_T, group, exchange, graph
        self.dev = model.device
        self.cap = shard_capacity(self.R, world, tile)
        self.n_mine = shard_indices(self.R, rank, world, tile).numel()
        self.mine = [torch.zeros((4 * self.cap,), device=self.dev) for _ in range(2)]
        self.gathered = [torch.empty((world * 4 * self.cap,), device=self.dev) for _ in range(2)] if exchange else [None, None]
        self.views = [shard_send_views(m, self.cap, self.n_mine) for m in self.mine]
        # Round 6, OPTIONAL (`two_in_flight=True`, bench.py --two-shares-in-flight): frame k is rendered on render stream k % 2 into scratch slot k % 2, so the march of frame
        # k + 1 runs beside the shade kernel of frame k and fills what its drain leaves (FrameStream's mechanism across the frames of the split stream: an 80 000-ray share is too
        # small for pieces INSIDE a call to settle, include/tvr.h PIECES, but a stream of such shares never drains).  Measured (profiles/r06_share_in_flight_ab.txt): with the
        # device-side exchange emulated locally — four streams on HIP's four hardware queues — the 8-way share takes 2.51 ms instead of 2.59 (-3.3 %; 4-way -3.2 %, 2-way +-0),
        # i.e. t1 / (8 t8) = 0.95 - 0.956; through a ONE-member RCCL group — five streams: RCCL brings its own — 2.40 / 2.50 / 2.56 ms in three processes against a steady
        # 2.455 - 2.461 (which two streams share a queue differs from process to process), and 2.47 against 2.43 with GPU_MAX_HW_QUEUES=8.  Not reproducible where it matters
        # and not measurable on eight cards here: OFF by default.  Never under `graph` (a captured render belongs to one stream) or off the GPU.
        self.two = bool(two_in_flight) and not graph and str(self.dev).startswith("cuda") and bool(getattr(model, "render_rays_is_the_frame", False))
        self.rstreams = [torch.cuda.Stream(self.dev) for _ in range(2)] if self.two else None
        self._drain_next = False
        self.side = torch.cuda.Stream(self.dev) if exchange else None
        self.rendered = [torch.cuda.Event() for _ in range(2)]       # frame in buffer b is rendered (compute stream -> side stream)
        self.exchanged = [torch.cuda.Event() for _ in range(2)]      # buffer b's exchange has finished (side stream -> compute stream)
        self.busy = [False, False]
        # the un-permuted frames land in static (padded) buffers: no allocation on the side stream, and the returned tensors are views that stay valid until
        # the buffer comes round again, two submits later
        L_ = self.cap // tile
        self.out_rgb = [torch.empty((L_ * world * tile, 3), device=self.dev) for _ in range(2)] if exchange else None
        self.out_depth = [torch.empty((L_ * world * tile,), device=self.dev) for _ in range(2)] if exchange else None
        self.out = [None, None]
        self.graphs = {}
        self.k = 0
        self._waited = None          # (buffer index, stream) of the exchange event the current stream was last made to wait for, if nothing was enqueued since
        self._gloo = bool(exchange == "dist" and dist.is_initialized() and dist.get_backend(group) == "gloo")

    def _render(self, sub, b):
        if self.two:
            self.model.render_rays(sub, white_bg=self.white_bg, N_samples=self.S, eps_T=self.eps_T, out=self.views[b], scratch_slot=b)
        else:
            self.model.render_rays(sub, white_bg=self.white_bg, N_samples=self.S, eps_T=self.eps_T, out=self.views[b])

    def _replay(self, key, sub, b):
        if not self.use_graph or self.n_mine == 0:
            if self.n_mine:
                self._render(sub, b)
            return
        g = self.graphs.get((key, b))
        if g is None:
            # warm up on a side stream (scratch, packed scene, the fp16-range decision: everything that allocates or reads the host happens here), then capture
            warm = torch.cuda.Stream(self.dev)
            warm.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(warm):
                self._render(sub, b)
            torch.cuda.current_stream(self.dev).wait_stream(warm)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            # thread_local: a process group's watchdog thread queries events while we capture; under the default (global) mode that call is an error for
            # it and the capture (one hang in seven runs of the one-rank RCCL test before this)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._render(sub, b)
            self.graphs[(key, b)] = (g, sub)                          # (the graph reads `sub`'s storage: keep it alive)
            g = self.graphs[(key, b)]
        g[0].replay()

    def prepare(self, key, sub_rays):
        """Capture the graphs of ray set `key` for BOTH send buffers now (a frame stream alternates buffers, so a key meets either); otherwise the first
        submit() that pairs a key with a buffer pays the capture (a warm-up render, a device synchronise and the capture itself: ~3 ms)."""
        if self.use_graph and self.n_mine:
            for b in range(2):
                if (key, b) not in self.graphs:
                    if self.busy[b]:
                        torch.cuda.current_stream(self.dev).wait_event(self.exchanged[b])
                    self._replay(key, sub_rays, b)
            torch.cuda.synchronize(self.dev)

    def _exchange(self, b):
        import torch.distributed as dist
        if self.exchange == "dist":
            if self._gloo and self.mine[b].is_cuda:                    # 1-GPU rehearsals: gloo has no all_gather_into_tensor for device tensors
                parts = [torch.empty_like(self.mine[b]) for _ in range(self.world)]
                dist.all_gather(parts, self.mine[b], group=self.group)
                self.gathered[b].copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(self.gathered[b], self.mine[b], group=self.group)
        else:
            self.gathered[b][:4 * self.cap].copy_(self.mine[b])
        # un-permute (shard_unpermute's two strided copies, into the static buffers): the frame is the [tiles-per-rank, world] transpose of the gathered tile grid
        W_, L_, T_, cap = self.world, self.cap // self.tile, self.tile, self.cap
        g = self.gathered[b].view(W_, 4 * cap)
        self.out_rgb[b].view(L_, W_, T_, 3).copy_(g[:, :3 * cap].view(W_, L_, T_, 3).transpose(0, 1))
        self.out_depth[b].view(L_, W_, T_).copy_(g[:, 3 * cap:].view(W_, L_, T_).transpose(0, 1))
        self.out[b] = (self.out_rgb[b][:self.R], self.out_depth[b][:self.R])

    def submit(self, key, sub_rays):
        """Enqueue the frame whose rays of THIS rank are `sub_rays` [n_mine,6] (shard_indices order; a static tensor per `key`: the graph captured for a key
        reads that storage).  Returns the (rgb [R,3], depth [R]) of the frame submitted one call earlier, or None."""
        b = self.k % 2
        cur = torch.cuda.current_stream(self.dev)
        rs = cur
        if self.two:
            # (as FrameStream: a call that re-packs the scene, converts fp16 copies or runs the arithmetic gate's probes must not overlap a frame that reads the packed images)
            settled = bool(getattr(self.model, "scene_settled", lambda: False)())
            if not settled or self._drain_next:
                cur.wait_stream(self.rstreams[0])
                cur.wait_stream(self.rstreams[1])
            self._drain_next = not settled
            rs = self.rstreams[b]
            rs.wait_stream(cur)                                       # the rays are ready; and (through the caller's stream) the frame handed out by the previous submit
            if self.busy[b]:
                rs.wait_event(self.exchanged[b] if self.exchange else self.rendered[b])
            with torch.cuda.stream(rs):
                self._replay(key, sub_rays, b)
                self.rendered[b].record(rs)
            # (the exchange stays on the side stream: riding on the frame's own render stream — one stream fewer — it delays that stream's next frame; measured +4 % through a
            #  one-member RCCL group and -0.8 % instead of -3.3 % in the local emulation, profiles/r06_share_in_flight_ab.txt)
            self._waited = None
        else:
            if self.busy[b] and self._waited != (b, cur.cuda_stream):     # (the previous submit already made this stream wait for that very event: one barrier packet less per frame)
                cur.wait_event(self.exchanged[b])                         # the exchange that last read this send buffer is done before it is rendered into again
            self._waited = None
            self._replay(key, sub_rays, b)
        prev = None
        if self.exchange:
            if not self.two:
                self.rendered[b].record(cur)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.rendered[b])
                self._exchange(b)
                self.exchanged[b].record(self.side)
            self.busy[b] = True
            pb = 1 - b
            if self.k > 0:
                cur.wait_event(self.exchanged[pb])                    # frame k - 1's pixels are complete for whatever the caller enqueues next
                prev = self.out[pb]
                self._waited = (pb, cur.cuda_stream)
        else:
            if self.two:
                self.busy[b] = True
                if self.k > 0:
                    cur.wait_event(self.rendered[1 - b])
            prev = (self.views[1 - b][0], self.views[1 - b][1]) if self.k > 0 else None
        self.k += 1
        return prev

    def flush(self):
        """The last submitted frame (waits, on the current stream, for its exchange)."""
        if self.k == 0:
            return None
        b = (self.k - 1) % 2
        if not self.exchange:
            if self.two:
                torch.cuda.current_stream(self.dev).wait_event(self.rendered[b])
                if self.k > 1:
                    torch.cuda.current_stream(self.dev).wait_event(self.rendered[1 - b])
            return self.views[b][0], self.views[b][1]
        torch.cuda.current_stream(self.dev).wait_event(self.exchanged[b])
        if self.two and self.k > 1:
            torch.cuda.current_stream(self.dev).wait_event(self.exchanged[1 - b])
        return self.out[b]
