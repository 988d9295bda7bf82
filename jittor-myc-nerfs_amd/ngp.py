"""Host side of the alt path (BASELINE configs[4]): JNeRF's Instant-NGP inference surface on the HIP kernels of
csrc/tvr_ngp.hip (C-ABI: include/tvr_ngp.h).  No CPU fallback: every op raises if libtvr.so is missing or a call fails.

Mirrors, under jnerf-myc/python/jnerf/:
  HashEncoder          models/position_encoders/hash_encoder/hash_encoder.py:9-30 (+ grid_encode.py:17-39 level table)
  SHEncoder            models/position_encoders/sh_encoder/sh_encoder.py:9-53
  NGPNetworks          models/networks/ngp_network.py:41-96 (plain-Linear branch, the fp32 configs Car.py / Easyship.py)
  DensityGridSampler   models/samplers/density_grid_sampler/density_grid_sampler.py:17-162 (inference: sample / rays2rgb / bitfield)
  render_img           runner/runner.py:195-228 (the 4096-ray slab loop) — and render_frame, the same image in one pass
  NerfRays             dataset/dataset.py:267-292,313-320 (ray generation for one camera in the NGP convention)
State-dict keys are the reference's (`pos_encoder.m_grid`, `density_mlp.0.weight`, ..., `density_grid`, `density_grid_bitfield`).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

NERF_GRIDSIZE, NERF_CASCADES, MAX_STEP = 128, 5, 1024
NERF_SCALE = 0.33                                   # dataset.py:14
_PCG_MULT = 0x5851F42D4C957F2D
_M64 = (1 << 64) - 1


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _need_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise L.TvrError(f"{what}: tensors must live on the GPU (there is no CPU fallback)")


# ------------------------------------------------------------------------------------------------------------------ RNG
class Pcg32:
    """The process-global `pcg32 rng{1337}` of the reference (ops/code_ops/global_vars.py:14-17, pcg32.h): the sampling kernel is
    handed its state and the host advances it by 2^32 after every sampling call (ray_sampler.py:61)."""

    def __init__(self, initstate: int = 1337, initseq: int = 1):
        self.state, self.inc = 0, ((initseq << 1) | 1) & _M64
        self._next()
        self.state = (self.state + initstate) & _M64
        self._next()

    def _next(self) -> int:
        old = self.state
        self.state = (old * _PCG_MULT + self.inc) & _M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def next_uint(self) -> int:
        return self._next()

    def advance(self, delta: int = 1 << 32) -> None:
        cur_mult, cur_plus, acc_mult, acc_plus = _PCG_MULT, self.inc, 1, 0
        delta &= _M64
        while delta > 0:
            if delta & 1:
                acc_mult = (acc_mult * cur_mult) & _M64
                acc_plus = (acc_plus * cur_mult + cur_plus) & _M64
            cur_plus = ((cur_mult + 1) * cur_plus) & _M64
            cur_mult = (cur_mult * cur_mult) & _M64
            delta >>= 1
        self.state = (acc_mult * self.state + acc_plus) & _M64


global_rng = Pcg32(1337)


# ------------------------------------------------------------------------------------------------------------------ encoders
def grid_levels(aabb_scale: int, n_levels: int = 16, base_resolution: int = 16, log2_hashmap_size: int = 19,
                desired_resolution: float = 2048.0) -> Tuple[np.ndarray, np.ndarray, float]:
    """`GridEncode.__init__` (grid_encode.py:17-39): per-level first entry and the kernel's per-level scale, in fp32 like the
    reference (`jt.exp(jt.log(..))`, `jt.pow(2, ..)`; `exp2f(level * log2_per_level_scale) * base_resolution - 1`)."""
    f32 = np.float32
    per_level_scale = float(np.exp(np.log(f32(desired_resolution * aabb_scale / base_resolution)) / f32(n_levels - 1)))
    log2s = float(np.log2(f32(per_level_scale)))
    offsets = np.zeros(n_levels + 1, np.uint32)
    scale = np.zeros(n_levels, np.float32)
    off = 0
    for i in range(n_levels):
        s = np.power(f32(2), f32(i * log2s)) * f32(base_resolution) - f32(1.0)
        res = int(np.ceil(s)) + 1
        n = min((res ** 3 + 7) // 8 * 8, 1 << log2_hashmap_size)
        offsets[i] = off
        off += n
        scale[i] = np.exp2(f32(i) * f32(np.log2(per_level_scale))) * f32(base_resolution) - f32(1.0)
    offsets[n_levels] = off
    return offsets, scale, per_level_scale


def _grid_cfg(offsets, scale) -> L.NgpGridCfg:
    g = L.NgpGridCfg()
    for i in range(17):
        g.offsets[i] = int(offsets[i])
    for i in range(16):
        g.scale[i] = float(scale[i])
    return g


class HashEncoder(torch.nn.Module):
    def __init__(self, aabb_scale: int = 1, n_pos_dims: int = 3, n_features_per_level: int = 2, n_levels: int = 16, base_resolution: int = 16,
                 log2_hashmap_size: int = 19):
        super().__init__()
        if (n_pos_dims, n_features_per_level, n_levels) != (3, 2, 16):
            raise NotImplementedError("the kernels are built for 3-D positions, 16 levels x 2 features (hash_encoder.py:18-19)")
        self.aabb_scale = aabb_scale
        self.offsets, self.scale, self.per_level_scale = grid_levels(aabb_scale, n_levels, base_resolution, log2_hashmap_size)
        self.m_n_params = int(self.offsets[-1]) * n_features_per_level
        self.m_grid = torch.nn.Parameter(torch.empty(self.m_n_params).uniform_(-1e-4, 1e-4))          # hash_encoder.py:23-24
        self.out_dim = n_features_per_level * n_levels
        self.cfg = _grid_cfg(self.offsets, self.scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _need_gpu(self.m_grid, "HashEncoder")
        x = x.detach().to(self.m_grid.device, torch.float32)
        if x.stride(-1) != 1:
            x = x.contiguous()
        n = x.shape[0]
        out = torch.empty(n, self.out_dim, device=x.device)
        L.check(L.lib().tvr_ngp_hash_encode(C.byref(self.cfg), self.m_grid.data_ptr(), x.data_ptr(), x.stride(0) if n else 3, n, out.data_ptr(),
                                            _stream(x.device)), "tvr_ngp_hash_encode")
        return out

    execute = forward


class SHEncoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.m_sh_degree, self.out_dim = 4, 16

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _need_gpu(x, "SHEncoder")
        x = x.detach().to(torch.float32)
        if x.stride(-1) != 1:
            x = x.contiguous()
        n = x.shape[0]
        out = torch.empty(n, 16, device=x.device)
        L.check(L.lib().tvr_ngp_sh_encode(x.data_ptr(), x.stride(0) if n else 3, n, out.data_ptr(), _stream(x.device)), "tvr_ngp_sh_encode")
        return out

    execute = forward


class NGPNetworks(torch.nn.Module):
    """`NGPNetworks` (ngp_network.py:41-96), plain-Linear branch.  `forward(pos, dir)` is ONE fused kernel (hash grid + SH + both
    MLPs on fp32 MFMAs); the encoders stay available as separate modules, as in the reference."""

    def __init__(self, aabb_scale: int = 1, use_fully: bool = True, density_hidden_layer: int = 1, density_n_neurons: int = 64, rgb_hidden_layer: int = 2,
                 rgb_n_neurons: int = 64):
        super().__init__()
        if (density_hidden_layer, density_n_neurons, rgb_hidden_layer, rgb_n_neurons) != (1, 64, 2, 64):
            raise NotImplementedError("the fused kernel is built for the 32-64-16 / 32-64-64-3 networks (ngp_network.py:43)")
        self.use_fully, self.using_fp16 = use_fully, False
        self.pos_encoder = HashEncoder(aabb_scale)
        self.dir_encoder = SHEncoder()
        Lin = torch.nn.Linear
        self.density_mlp = torch.nn.Sequential(Lin(32, 64, bias=False), torch.nn.ReLU(), Lin(64, 16, bias=False))
        self.rgb_mlp = torch.nn.Sequential(Lin(32, 64, bias=False), torch.nn.ReLU(), Lin(64, 64, bias=False), torch.nn.ReLU(), Lin(64, 3, bias=False))
        self._packed: Optional[torch.Tensor] = None
        self._sig = None

    def _weights(self):
        return [self.density_mlp[0].weight, self.density_mlp[2].weight, self.rgb_mlp[0].weight, self.rgb_mlp[2].weight, self.rgb_mlp[4].weight]

    def packed(self, force: bool = False) -> torch.Tensor:
        ws = self._weights()
        _need_gpu(ws[0], "NGPNetworks")
        sig = tuple((w.data_ptr(), w._version) for w in ws)
        if force or self._packed is None or sig != self._sig:
            dev = ws[0].device
            cs = [w.detach().to(torch.float32).contiguous() for w in ws]
            p = L.NgpNetParams(*[c.data_ptr() for c in cs])
            nbytes = L.lib().tvr_ngp_net_packed_bytes()
            if self._packed is None or self._packed.device != dev:
                self._packed = torch.empty(nbytes // 4, device=dev)
            L.check(L.lib().tvr_ngp_net_pack(C.byref(p), self._packed.data_ptr(), nbytes, _stream(dev)), "tvr_ngp_net_pack")
            self._sig = sig
        return self._packed

    def _run(self, pos: torch.Tensor, dirs: torch.Tensor, n_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        grid = self.pos_encoder.m_grid
        _need_gpu(grid, "NGPNetworks")
        n = pos.shape[0]
        if out is None:
            out = torch.empty(n, 4, device=grid.device)
        if n:
            L.check(L.lib().tvr_ngp_network(C.byref(self.pos_encoder.cfg), grid.data_ptr(), self.packed().data_ptr(), pos.data_ptr(), pos.stride(0),
                                            dirs.data_ptr(), dirs.stride(0), n, None if n_dev is None else n_dev.data_ptr(), out.data_ptr(),
                                            _stream(grid.device)), "tvr_ngp_network")
        return out

    @staticmethod
    def _rows(x: torch.Tensor, device) -> torch.Tensor:
        x = x.detach().to(device, torch.float32)
        return x if (x.dim() == 2 and x.stride(1) == 1 and x.shape[1] >= 3) else x.contiguous()

    def forward(self, pos_input: torch.Tensor, dir_input: torch.Tensor) -> torch.Tensor:
        """[n,3] positions in [0,1] and directions warped to [0,1] -> [n,4] = (rgb raw, density raw) (`execute_`, ngp_network.py:78-85).
        Row-strided views (the sampler's `coords[:, :3]`, `coords[:, 4:]`) are read in place."""
        dev = self.pos_encoder.m_grid.device
        return self._run(self._rows(pos_input, dev), self._rows(dir_input, dev))

    execute = forward

    def density(self, pos_input: torch.Tensor) -> torch.Tensor:
        """`density` (ngp_network.py:87-90): [n,3] -> [n,1]."""
        p = self._rows(pos_input, self.pos_encoder.m_grid.device)
        return self._run(p, p)[:, 3:4]

    def set_fp16(self):
        pass                                                            # fp16 = False in the shipped scene configs


# ------------------------------------------------------------------------------------------------------------------ sampler
class DensityGridSampler(torch.nn.Module):
    """Inference half of `DensityGridSampler` (density_grid_sampler.py): `sample` -> `model` -> `rays2rgb(inference=True)`.
    Buffers keep the reference's names so a converted `ckpt['sampler']` loads with `load_state_dict`."""

    def __init__(self, model: NGPNetworks, aabb_scale: int = 1, n_rays_per_batch: int = 4096, near_distance: float = 0.2,
                 cone_angle_constant: float = 0.00390625, const_dt: bool = True, background_color: Sequence[float] = (1.0, 1.0, 1.0),
                 rng: Optional[Pcg32] = None):
        super().__init__()
        if aabb_scale > (1 << (NERF_CASCADES - 1)):
            raise ValueError(f"NeRF dataset's aabb_scale must <= {1 << (NERF_CASCADES - 1)}, but now is {aabb_scale}")     # :55-58
        self.__dict__["model"] = model                                  # not a sub-module (the reference reaches it through the config)
        self.aabb_scale, self.n_rays_per_batch = aabb_scale, n_rays_per_batch
        self.aabb_range = (0.5 - aabb_scale / 2, 0.5 + aabb_scale / 2)  # dataset.py:214-215
        self.near_distance, self.cone_angle_constant, self.const_dt = near_distance, cone_angle_constant, const_dt
        self.background_color = [float(c) for c in background_color]
        self.rng = rng if rng is not None else global_rng
        self.MAX_STEP = MAX_STEP
        n = NERF_CASCADES * NERF_GRIDSIZE ** 3
        self.register_buffer("density_grid", torch.zeros(n))
        self.register_buffer("density_grid_bitfield", torch.zeros(n // 8, dtype=torch.uint8))
        self.register_buffer("density_grid_mean", torch.zeros(1))
        self._coords = self._rays_numsteps = self._counter = None
        self._scratch: Optional[torch.Tensor] = None

    # -- maintenance
    def update_bitfield(self) -> None:
        """`update_bitfield` (update_bitfield.py:14-31) from `density_grid`."""
        _need_gpu(self.density_grid, "DensityGridSampler")
        dev = self.density_grid.device
        L.check(L.lib().tvr_ngp_update_bitfield(self.density_grid.data_ptr(), self.density_grid_bitfield.data_ptr(), self.density_grid_mean.data_ptr(),
                                                None, 0, _stream(dev)), "tvr_ngp_update_bitfield")

    # -- sampling
    def _cfg(self, slab_rays: int) -> L.NgpMarchCfg:
        c = L.NgpMarchCfg()
        c.aabb_lo[:] = [self.aabb_range[0]] * 3
        c.aabb_hi[:] = [self.aabb_range[1]] * 3
        c.near_distance, c.cone_angle, c.const_dt = self.near_distance, self.cone_angle_constant, int(self.const_dt)
        c.rng_state, c.rng_inc, c.slab_rays = self.rng.state, self.rng.inc, slab_rays
        return c

    def _sample_raw(self, rays_o: torch.Tensor, rays_d: torch.Tensor, max_samples: int, slab_rays: int = 0, want_index: bool = False):
        dev = self.density_grid_bitfield.device
        _need_gpu(self.density_grid_bitfield, "DensityGridSampler")
        o = rays_o.detach().to(dev, torch.float32).contiguous()
        d = rays_d.detach().to(dev, torch.float32).contiguous()
        R = o.shape[0]
        coords = torch.empty(max_samples, 7, device=dev)
        numsteps = torch.empty(R, 2, dtype=torch.int32, device=dev)
        index = torch.empty(R, dtype=torch.int32, device=dev) if want_index else None
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        need = L.lib().tvr_ngp_sample_scratch_bytes(R)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != dev:
            self._scratch = L.dev_bytes(need, dev, what="tvr_ngp scratch")
        cfg = self._cfg(slab_rays)
        L.check(L.lib().tvr_ngp_sample(C.byref(cfg), o.data_ptr(), d.data_ptr(), R, self.density_grid_bitfield.data_ptr(), coords.data_ptr(), max_samples,
                                       numsteps.data_ptr(), None if index is None else index.data_ptr(), counter.data_ptr(), self._scratch.data_ptr(),
                                       self._scratch.numel(), _stream(dev)), "tvr_ngp_sample")
        n_slabs = 1 if slab_rays == 0 else (R + slab_rays - 1) // slab_rays
        for _ in range(n_slabs):
            self.rng.advance()                                          # ray_sampler.py:61, once per reference call
        return coords, index, numsteps, counter

    def sample(self, img_ids, rays_o: torch.Tensor, rays_d: torch.Tensor, rgb_target=None, is_training: bool = False):
        """`sample` (density_grid_sampler.py:133-146), inference branch: returns (coords_pos [n,3], coords_dir [n,3]) — views of the
        sampler's [n,7] rows.  Like the reference this reads the sample count back (`.item()`, ray_sampler.py:69)."""
        if is_training:
            raise NotImplementedError("training-time sampling (density-grid update, compaction) is outside this round's scope")
        R = rays_o.shape[0]
        coords, index, numsteps, counter = self._sample_raw(rays_o, rays_d, R * self.MAX_STEP, want_index=True)
        samples = int(counter[1].item())
        coords = coords[:samples]
        self._coords, self._rays_numsteps, self._counter, self._rays_index = coords, numsteps, counter, index
        return coords[:, :3], coords[:, 4:]

    def rays2rgb(self, network_outputs: torch.Tensor, training_background_color=None, inference: bool = False) -> torch.Tensor:
        """`rays2rgb` (density_grid_sampler.py:163-190) -> `CalcRgb.inference` (calc_rgb.py:118-150)."""
        if not inference:
            raise NotImplementedError("only inference=True (CalcRgb.inference) is built")
        assert network_outputs.shape[0] == self._coords.shape[0]
        return self._composite(network_outputs, self._coords, self._rays_numsteps,
                               self.background_color if training_background_color is None else training_background_color)

    def _composite(self, net_out, coords, numsteps, bg) -> torch.Tensor:
        dev = numsteps.device
        R = numsteps.shape[0]
        rgb = torch.empty(R, 3, device=dev)
        o = net_out.detach().to(dev, torch.float32).contiguous()
        bgc = (C.c_float * 3)(*[float(b) for b in bg])
        L.check(L.lib().tvr_ngp_composite(o.data_ptr(), coords.data_ptr(), numsteps.data_ptr(), R, C.byref(bgc), rgb.data_ptr(), _stream(dev)), "tvr_ngp_composite")
        return rgb

    # -- one frame without the slab loop
    def render_frame(self, rays_o: torch.Tensor, rays_d: torch.Tensor, stats: Optional[dict] = None, profile: Optional[dict] = None) -> torch.Tensor:
        """The image `render_img` produces, in two launches and no host read (`tvr_ngp_render`): the march over all rays (each ray draws
        the jitter it would get in its 4096-ray slab), then one kernel that walks every ray's steps through the encoders and networks
        and composites them in order, stopping where `compute_rgbs_inference` breaks (T < 1e-4).  Nothing in between is materialised.
        `profile` (a dict) switches to the measuring entry point: per-kernel milliseconds, at the price of waiting for the frame."""
        dev = self.density_grid_bitfield.device
        _need_gpu(self.density_grid_bitfield, "DensityGridSampler")
        o = rays_o.detach().to(dev, torch.float32).contiguous()
        d = rays_d.detach().to(dev, torch.float32).contiguous()
        R = o.shape[0]
        rgb = torch.empty(R, 3, device=dev)
        need = L.lib().tvr_ngp_render_scratch_bytes(R)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != dev:
            self._scratch = L.dev_bytes(need, dev, what="tvr_ngp scratch")
        st = torch.zeros(2, dtype=torch.int64, device=dev) if stats is not None else None
        cfg = self._cfg(self.n_rays_per_batch)
        bgc = (C.c_float * 3)(*self.background_color)
        m = self.model
        args = (C.byref(cfg), C.byref(m.pos_encoder.cfg), m.pos_encoder.m_grid.data_ptr(), m.packed().data_ptr(), o.data_ptr(), d.data_ptr(),
                R, self.density_grid_bitfield.data_ptr(), C.byref(bgc), rgb.data_ptr(), None if st is None else st.data_ptr(),
                self._scratch.data_ptr(), self._scratch.numel(), _stream(dev))
        if profile is None:
            L.check(L.lib().tvr_ngp_render(*args), "tvr_ngp_render")
        else:                                                           # HIP events around the two kernels; waits for completion
            ms = (C.c_float * 2)()
            L.check(L.lib().tvr_ngp_render_profiled(*args, C.byref(ms)), "tvr_ngp_render_profiled")
            profile.update(march_ms=float(ms[0]), render_ms=float(ms[1]))
        for _ in range((R + self.n_rays_per_batch - 1) // self.n_rays_per_batch):
            self.rng.advance()                                          # as many advances as the slab loop would make
        if stats is not None:
            ev, tot = (int(v) for v in st.tolist())
            stats.update(evaluated=ev, samples=tot)
        return rgb

    def render_frame_rows(self, rays_o: torch.Tensor, rays_d: torch.Tensor, samples_per_ray_hint: int = 256, stats: Optional[dict] = None) -> torch.Tensor:
        """The same image through the three stand-alone entry points over the whole frame (rows of every step, network outputs of every
        row, compositing): what `render_img` does without the slabs.  The only host read is the total at the end, to detect that the row
        buffer was too small (then it is re-rendered larger).  Kept as the cross-check of `render_frame` and as the bulk user of the
        row-level ABI."""
        R = rays_o.shape[0]
        state0 = (self.rng.state, self.rng.inc)
        cap = max(1, min(R * self.MAX_STEP, R * samples_per_ray_hint))
        while True:
            coords, _, numsteps, counter = self._sample_raw(rays_o, rays_d, cap, slab_rays=self.n_rays_per_batch)
            out = torch.empty(cap, 4, device=coords.device)
            self.model._run(coords[:, :3], coords[:, 4:], n_dev=counter[1:], out=out)
            rgb = self._composite(out, coords, numsteps, self.background_color)
            total = int(counter[1].item())
            if total <= cap:
                if stats is not None:
                    stats.update(samples=total, capacity=cap)
                return rgb
            del coords, out
            self.rng.state, self.rng.inc = state0                       # same jitter on the retry
            cap = min(R * self.MAX_STEP, int(total * 1.05) + 1024)


def render_img(sampler: DensityGridSampler, model: NGPNetworks, rays_o: torch.Tensor, rays_d: torch.Tensor) -> torch.Tensor:
    """`Runner.render_img` (runner.py:209-222): slabs of n_rays_per_batch rays, the last one padded with rays of ones."""
    R, B = rays_o.shape[0], sampler.n_rays_per_batch
    imgs = torch.empty(R + B, 3, device=rays_o.device)
    for pixel in range(0, R, B):
        o, d = rays_o[pixel:pixel + B], rays_d[pixel:pixel + B]
        if o.shape[0] < B:
            pad = torch.ones(B - o.shape[0], 3, dtype=o.dtype, device=o.device)
            o, d = torch.cat([o, pad]), torch.cat([d, pad])
        pos, dirs = sampler.sample(None, o, d)
        out = model(pos, dirs)
        imgs[pixel:pixel + B] = sampler.rays2rgb(out, inference=True)
    return imgs[:R]


# ------------------------------------------------------------------------------------------------------------------ rays
def matrix_nerf2ngp(matrix, scale: float = NERF_SCALE, offset=(0.5, 0.5, 0.5), correct_pose=(-1, -1, 1)) -> np.ndarray:
    """dataset.py:313-320: Blender camera-to-world [3or4,4] -> NGP convention [3,4]."""
    m = np.array(matrix, np.float32)[:3].copy()
    for k in range(3):
        m[:, k] *= correct_pose[k]
    m[:, 3] = m[:, 3] * np.float32(scale) + np.asarray(offset, np.float32)
    return m[[1, 2, 0]]


def generate_rays(xform, W: int, H: int, focal: Sequence[float], principal=(0.5, 0.5), device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`generate_rays_total_test` (dataset.py:267-292) for one camera: rays_o, rays_d [H*W,3] fp32."""
    x = torch.as_tensor(np.asarray(xform, np.float32), device=device)
    gx = (torch.linspace(0, H - 1, H, device=device) + 0.5) / H
    gy = (torch.linspace(0, W - 1, W, device=device) + 0.5) / W
    a, b = torch.meshgrid(gx, gy, indexing="ij")
    xy = torch.stack([a, b], -1).permute(1, 0, 2).reshape(-1, 2)
    res = torch.tensor([W, H], dtype=torch.float32, device=device)
    d = torch.cat([(xy - torch.tensor(principal, dtype=torch.float32, device=device)) * res / torch.tensor(list(focal), dtype=torch.float32, device=device),
                   torch.ones(H * W, 1, device=device)], -1)
    d = (x[:, :3] @ d[:, :, None])[:, :, 0]
    d = d / torch.clamp(torch.sqrt((d * d).sum(-1, keepdim=True)), min=1e-12)
    return x[:, 3].expand_as(d).contiguous(), d.contiguous()


def fov_to_focal_length(resolution: int, degrees: float) -> float:
    return 0.5 * resolution / math.tan(0.5 * degrees * math.pi / 180)      # dataset.py:17-18


def load_scene_arrays(model: NGPNetworks, sampler: Optional[DensityGridSampler], arrs: Dict[str, np.ndarray]) -> None:
    """Fill a model / sampler from a flat dict keyed like the reference's state dicts (`synthetic.make_ngp_scene_arrays`, or a
    converted `ckpt['model']` / `ckpt['sampler']`)."""
    with torch.no_grad():
        model.pos_encoder.m_grid.copy_(torch.as_tensor(arrs["grid" if "grid" in arrs else "pos_encoder.m_grid"]).reshape(-1))
        for name, mod in (("density_mlp.0", model.density_mlp[0]), ("density_mlp.2", model.density_mlp[2]), ("rgb_mlp.0", model.rgb_mlp[0]),
                          ("rgb_mlp.2", model.rgb_mlp[2]), ("rgb_mlp.4", model.rgb_mlp[4])):
            mod.weight.copy_(torch.as_tensor(arrs[name + ".weight"]))
        model._sig = None
        if sampler is not None:
            if "density_grid" in arrs:
                sampler.density_grid.copy_(torch.as_tensor(arrs["density_grid"]))
            if "density_grid_bitfield" in arrs:
                sampler.density_grid_bitfield.copy_(torch.as_tensor(arrs["density_grid_bitfield"]))
            elif sampler.density_grid.is_cuda:
                sampler.update_bitfield()


# ------------------------------------------------------------------------------------------------------------------ formats either side
def load_jnerf_checkpoint(path: str, model: NGPNetworks, sampler: Optional[DensityGridSampler] = None) -> int:
    """`Runner.load_ckpt` (runner/runner.py:137-144) for inference: the file `Runner.save_ckpt` (:127-135) writes with `jt.save` is a pickle
    of `{'global_step', 'model': state_dict, 'sampler': state_dict, optimizer states...}` with every jt.Var turned into a numpy array
    (Jittor's published behaviour, restated; Jittor itself is absent here).  Fills `model` from `ckpt['model']` (`pos_encoder.m_grid`,
    `density_mlp.{0,2}.weight`, `rgb_mlp.{0,2,4}.weight`) and `sampler` from `ckpt['sampler']` (`density_grid`, `density_grid_bitfield`,
    `density_grid_mean`); fp16 checkpoints (`fp16 = True` configs) are widened to fp32.  Returns `global_step`."""
    from .field import _ArrayUnpickler                       # numpy arrays + builtin containers only: a downloaded checkpoint is untrusted input
    with open(path, "rb") as f:
        ckpt = _ArrayUnpickler.load(f)
    m = {k: np.asarray(v) for k, v in ckpt["model"].items()}
    # `fp16 = True` configs (ngp_comp.py:100) build the networks as FMLP (ngp_network.py:9-39): each keeps ONE flat `con_weights` =
    # concat_i(dweights[i].T.reshape(-1)) with dweights[i] of shape (in, out) and the last layer zero-padded to 16 outputs.  dweights[i].T is
    # the [out, in] matrix of the equivalent bias-free Linear; they are evaluated here in fp32 (the FMLP kernels themselves ship without
    # source, so their fp16 rounding is not reproduced).
    for prefix, shapes, names in (("density_mlp", [(64, 32), (16, 64)], ["0", "2"]), ("rgb_mlp", [(64, 32), (64, 64), (16, 64)], ["0", "2", "4"])):
        key = prefix + ".con_weights"
        if key in m and prefix + ".0.weight" not in m:
            flat, off = m[key].astype(np.float32).reshape(-1), 0
            if flat.size != sum(a * b for a, b in shapes):
                raise ValueError(f"{path}: {key} has {flat.size} entries, expected {sum(a * b for a, b in shapes)}")
            for (o, i), nm in zip(shapes, names):
                m[f"{prefix}.{nm}.weight"] = flat[off:off + o * i].reshape(o, i)
                off += o * i
            if prefix == "rgb_mlp":
                m["rgb_mlp.4.weight"] = m["rgb_mlp.4.weight"][:3]          # drop the zero padding rows
    want = {"pos_encoder.m_grid": (model.pos_encoder.m_n_params,), "density_mlp.0.weight": (64, 32), "density_mlp.2.weight": (16, 64),
            "rgb_mlp.0.weight": (64, 32), "rgb_mlp.2.weight": (64, 64), "rgb_mlp.4.weight": (3, 64)}
    for k, shape in want.items():
        if k not in m:
            raise KeyError(f"{path}: 'model' has no '{k}'")
        if tuple(m[k].shape) != shape:
            raise ValueError(f"{path}: {k} has shape {tuple(m[k].shape)}, this model (aabb_scale {model.pos_encoder.aabb_scale}) needs {shape}")
    load_scene_arrays(model, None, {k: m[k].astype(np.float32) for k in want})
    if sampler is not None and "sampler" in ckpt:
        s = {k: np.asarray(v) for k, v in ckpt["sampler"].items()}
        with torch.no_grad():
            for k in ("density_grid", "density_grid_bitfield", "density_grid_mean"):
                if k in s:
                    buf = getattr(sampler, k)
                    buf.copy_(torch.as_tensor(s[k].reshape(-1)[:buf.numel()].astype(np.uint8 if k.endswith("bitfield") else np.float32)).reshape(buf.shape))
    return int(ckpt.get("global_step", 0))


class NerfRays:
    """Test-set rays of a `NerfDataset` (dataset/dataset.py:82-228, mode 'test', `have_img=False`): `transforms_test.json` in the NeRF
    synthetic format -> NGP-convention camera matrices (`matrix_nerf2ngp`: `correct_pose`, `scale`, `offset`, axis cycle) and focal
    lengths (`fl_x` / `camera_angle_x`, :187-203); `rays(i)` is `generate_rays_total_test` for image i."""

    def __init__(self, root_dir: str, mode: str = "test", H: int = 800, W: int = 800, correct_pose=(1, -1, -1), aabb_scale: Optional[int] = None,
                 scale: Optional[float] = None, offset=None, device=None):
        import json
        import os
        with open(os.path.join(root_dir, f"transforms_{mode}.json")) as f:
            meta = json.load(f)
        self.H, self.W, self.device = int(meta.get("h", H)), int(meta.get("w", W)), device
        self.scale = NERF_SCALE if scale is None else scale
        self.offset = [0.5, 0.5, 0.5] if offset is None else offset
        self.aabb_scale = meta.get("aabb_scale", 1) if aabb_scale is None else aabb_scale
        self.aabb_range = (0.5 - self.aabb_scale / 2, 0.5 + self.aabb_scale / 2)

        def focal(res, axis):
            if "fl_" + axis in meta:
                return float(meta["fl_" + axis])
            if "camera_angle_" + axis in meta:
                return fov_to_focal_length(res, meta["camera_angle_" + axis] * 180 / math.pi)
            return 0.0
        fx, fy = focal(self.W, "x"), focal(self.H, "y")
        if fx == 0 and fy == 0:
            raise RuntimeError("Couldn't read fov.")
        self.focal = (fx or fy, fy or fx)
        self.principal = (meta.get("cx", self.W / 2) / self.W, meta.get("cy", self.H / 2) / self.H)
        self.transforms = [matrix_nerf2ngp(np.asarray(fr["transform_matrix"], np.float32), self.scale, self.offset, correct_pose) for fr in meta["frames"]]
        self.n_images = len(self.transforms)

    def rays(self, img_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return generate_rays(self.transforms[img_id], self.W, self.H, self.focal, self.principal, device=self.device)
