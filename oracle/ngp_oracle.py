"""oracle/ngp_oracle.py — CPU restatement of the reference's alt path: JNeRF Instant-NGP inference.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (see oracle/ngp_oracle.c header: no Jittor, no CUDA toolchain, two kernels ship without source).

Two implementations that check each other:
  (b) oracle/ngp_oracle.c through ctypes — scalar C, the one that can march rays (bit-exact sample positions);
  (a) the vectorised numpy functions below for the encoders, the networks and the compositing.
Plus the host-side bookkeeping of the reference restated: hash-grid level table (`grid_encode.py:17-39`), ray generation
(`dataset/dataset.py:267-292`, `:313-320`), the render loop (`runner/runner.py:195-228`) and the process-global RNG
(`ops/code_ops/global_vars.py:14-17`, `ray_sampler.py:61`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libngp_oracle.so")
_FP = C.POINTER(C.c_float)
_U8P = C.POINTER(C.c_uint8)
_I32P = C.POINTER(C.c_int32)
_U32P = C.POINTER(C.c_uint32)

NERF_GRIDSIZE, NERF_CASCADES, NERF_STEPS = 128, 5, 1024
MIN_CONE_STEPSIZE = np.float32(np.float32(1.73205080757) / np.float32(1024))
NERF_SCALE = 0.33                                                   # dataset.py:14


class _March(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("near_distance", C.c_float), ("cone_angle", C.c_float),
                ("const_dt", C.c_int32), ("slab_rays", C.c_uint32)]


class _Grid(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("offsets", C.c_uint32 * 33), ("scale", C.c_float * 32)]


class _Net(C.Structure):
    _fields_ = [("d0", _FP), ("d1", _FP), ("c0", _FP), ("c1", _FP), ("c2", _FP)]


class _Rng(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ngp_oracle_sample.restype = C.c_int64
        L.ngp_oracle_sample.argtypes = [C.POINTER(_March), _FP, _FP, C.c_int64, _U8P, C.c_uint64, C.c_uint64, C.c_uint32, _FP, _I32P,
                                        _I32P, _U32P, _FP]
        L.ngp_oracle_update_bitfield.restype = None
        L.ngp_oracle_update_bitfield.argtypes = [_FP, _U8P, _FP]
        L.ngp_oracle_hash_encode.restype = None
        L.ngp_oracle_hash_encode.argtypes = [C.POINTER(_Grid), _FP, _FP, C.c_int64, _FP, _U32P]
        L.ngp_oracle_sh_encode.restype = None
        L.ngp_oracle_sh_encode.argtypes = [_FP, C.c_int64, _FP]
        L.ngp_oracle_network.restype = None
        L.ngp_oracle_network.argtypes = [C.POINTER(_Grid), _FP, C.POINTER(_Net), _FP, C.c_int64, _FP]
        L.ngp_oracle_composite.restype = None
        L.ngp_oracle_composite.argtypes = [_FP, _FP, _I32P, C.c_int64, _FP, _FP, _FP]
        L.ngp_rng_seed.argtypes = [C.POINTER(_Rng), C.c_uint64, C.c_uint64]
        L.ngp_rng_advance.argtypes = [C.POINTER(_Rng), C.c_uint64]
        L.ngp_rng_next_uint.restype = C.c_uint32
        L.ngp_rng_next_uint.argtypes = [C.POINTER(_Rng)]
        L.ngp_rng_next_float.restype = C.c_float
        L.ngp_rng_next_float.argtypes = [C.POINTER(_Rng)]
        L.ngp_morton3d.restype = C.c_uint32
        L.ngp_morton3d.argtypes = [C.c_uint32] * 3
        L.ngp_cascaded_grid_idx_at.restype = C.c_uint32
        L.ngp_cascaded_grid_idx_at.argtypes = [_FP, C.c_uint32]
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_FP):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------ RNG (global_vars.py:14-17: `pcg32 rng{1337}`)
class Pcg32:
    """PCG32 (ops/op_include/pcg32/pcg32.h).  `Pcg32(1337)` is the reference's process-global generator; every sampling call hands
    the current state to the kernel and then advances it by 2**32 (`ray_sampler.py:61`)."""

    def __init__(self, initstate: int = 1337, initseq: int = 1):
        self.r = _Rng()
        lib().ngp_rng_seed(C.byref(self.r), initstate, initseq)

    @property
    def state(self) -> Tuple[int, int]:
        return int(self.r.state), int(self.r.inc)

    def advance(self, delta: int = 1 << 32):
        lib().ngp_rng_advance(C.byref(self.r), delta & 0xFFFFFFFFFFFFFFFF)

    def next_uint(self) -> int:
        return int(lib().ngp_rng_next_uint(C.byref(self.r)))

    def next_float(self) -> float:
        return float(lib().ngp_rng_next_float(C.byref(self.r)))


# ------------------------------------------------------------------ hash-grid level table (grid_encode.py:17-39, HashEncode.h:142-145)
def grid_levels(aabb_scale: int, n_levels: int = 16, base_resolution: int = 16, log2_hashmap_size: int = 19,
                desired_resolution: float = 2048.0) -> Dict[str, np.ndarray]:
    """Offsets (in entries) and per-level scales.  The reference evaluates these in fp32 (`jt.exp(jt.log(..))`, `jt.pow(2, ..)`,
    `exp2f` in the kernel); so does this, with numpy fp32 ops."""
    f32 = np.float32
    per_level_scale = float(np.exp(np.log(f32(desired_resolution * aabb_scale / base_resolution)) / f32(n_levels - 1)))
    log2s = float(np.log2(f32(per_level_scale)))
    offsets = np.zeros(n_levels + 1, np.uint32)
    off = 0
    for i in range(n_levels):
        scale = np.power(f32(2), f32(i * log2s)) * f32(base_resolution) - f32(1.0)
        resolution = int(np.ceil(scale)) + 1
        params = resolution ** 3
        params = (params + 7) // 8 * 8
        params = min(params, 1 << log2_hashmap_size)
        offsets[i] = off
        off += params
    offsets[n_levels] = off
    kscale = np.array([np.exp2(f32(l) * f32(np.log2(per_level_scale))) * f32(base_resolution) - f32(1.0) for l in range(n_levels)],
                      np.float32)
    return {"offsets": offsets, "scale": kscale, "n_params": int(off) * 2, "per_level_scale": per_level_scale}


def _grid_struct(levels) -> _Grid:
    g = _Grid()
    n = len(levels["scale"])
    g.n_levels = n
    for i in range(n + 1):
        g.offsets[i] = int(levels["offsets"][i])
    for i in range(n):
        g.scale[i] = float(levels["scale"][i])
    return g


# ------------------------------------------------------------------ (b) C oracle wrappers
def aabb_range(aabb_scale: float) -> Tuple[float, float]:
    return (0.5 - aabb_scale / 2, 0.5 + aabb_scale / 2)                 # dataset.py:214-215


def sample(rays_o, rays_d, bitfield, aabb_scale, rng_state, near_distance=0.2, cone_angle=0.00390625, const_dt=True,
           max_samples: Optional[int] = None, slab_rays: int = 0):
    """`RaySampler.execute` (ray_sampler.py:20-72).  Returns coords [n,7], rays_index [R], numsteps [R,2], counter [2], startt [R]."""
    o, d = _f(rays_o), _f(rays_d)
    R = o.shape[0]
    bits = np.ascontiguousarray(bitfield, np.uint8)
    assert bits.size == NERF_GRIDSIZE ** 3 * NERF_CASCADES // 8
    cfg = _March()
    lo, hi = aabb_range(aabb_scale)
    cfg.lo[:] = [lo] * 3
    cfg.hi[:] = [hi] * 3
    cfg.near_distance, cfg.cone_angle, cfg.const_dt, cfg.slab_rays = near_distance, cone_angle, int(const_dt), slab_rays
    if max_samples is None:
        max_samples = R * NERF_STEPS                                     # ray_sampler.py:15
    numsteps = np.zeros((R, 2), np.int32)
    index = np.zeros(R, np.int32)
    counter = np.zeros(2, np.uint32)
    startt = np.zeros(R, np.float32)
    args = (C.byref(cfg), _p(o), _p(d), R, _p(bits, _U8P), rng_state[0], rng_state[1], max_samples)
    total = lib().ngp_oracle_sample(*args, None, _p(numsteps, _I32P), _p(index, _I32P), _p(counter, _U32P), _p(startt))
    kept = int(numsteps[:, 0].sum())
    coords = np.zeros((min(int(total), max_samples), 7), np.float32)    # cudaMemsetAsync(coords_out, 0) `ray_sampler.py:50`
    if kept:
        lib().ngp_oracle_sample(*args, _p(coords), _p(numsteps, _I32P), _p(index, _I32P), _p(counter, _U32P), _p(startt))
    return coords, index, numsteps, counter, startt


def update_bitfield(density_grid):
    g = _f(density_grid)
    assert g.size == NERF_GRIDSIZE ** 3 * NERF_CASCADES
    bits = np.zeros(g.size // 8, np.uint8)
    mean = np.zeros(1, np.float32)
    lib().ngp_oracle_update_bitfield(_p(g), _p(bits, _U8P), _p(mean))
    return bits, float(mean[0])


def hash_encode_c(levels, grid, pos, want_cells=False):
    pos, grid = _f(pos), _f(grid)
    n, L = pos.shape[0], len(levels["scale"])
    out = np.zeros((n, 2 * L), np.float32)
    cells = np.zeros((n, L, 3), np.uint32) if want_cells else None
    g = _grid_struct(levels)
    lib().ngp_oracle_hash_encode(C.byref(g), _p(grid), _p(pos), n, _p(out), _p(cells, _U32P) if want_cells else None)
    return (out, cells) if want_cells else out


def sh_encode_c(dir01):
    d = _f(dir01)
    out = np.zeros((d.shape[0], 16), np.float32)
    lib().ngp_oracle_sh_encode(_p(d), d.shape[0], _p(out))
    return out


def network_c(levels, scene: Dict[str, np.ndarray], coords):
    c = _f(coords)
    keep = {k: _f(scene[k]) for k in ("grid", "density_mlp.0.weight", "density_mlp.2.weight", "rgb_mlp.0.weight", "rgb_mlp.2.weight",
                                      "rgb_mlp.4.weight")}
    net = _Net(_p(keep["density_mlp.0.weight"]), _p(keep["density_mlp.2.weight"]), _p(keep["rgb_mlp.0.weight"]),
               _p(keep["rgb_mlp.2.weight"]), _p(keep["rgb_mlp.4.weight"]))
    out = np.zeros((c.shape[0], 4), np.float32)
    g = _grid_struct(levels)
    lib().ngp_oracle_network(C.byref(g), _p(keep["grid"]), C.byref(net), _p(c), c.shape[0], _p(out))
    return out


def composite_c(net_out, coords, numsteps, bg=(1.0, 1.0, 1.0)):
    o, c, ns = _f(net_out), _f(coords), np.ascontiguousarray(numsteps, np.int32)
    R = ns.shape[0]
    rgb = np.zeros((R, 3), np.float32)
    T = np.zeros(R, np.float32)
    b = _f(bg)
    lib().ngp_oracle_composite(_p(o), _p(c), _p(ns, _I32P), R, _p(b), _p(rgb), _p(T))
    return rgb, T


# ------------------------------------------------------------------ (a) numpy restatement of encoders / networks / compositing
def hash_encode(levels, grid, pos):
    """`kernel_grid` (HashEncode.h:117-199) vectorised over samples.  fp32 throughout; the accumulation uses separately rounded
    multiply-add (numpy has no fmaf), so it agrees with (b) to ~1 ulp of the partial sums, not bit for bit."""
    f32 = np.float32
    pos, grid = _f(pos), _f(grid).reshape(-1, 2)
    n, L = pos.shape[0], len(levels["scale"])
    out = np.zeros((n, 2 * L), f32)
    for l in range(L):
        tab = grid[int(levels["offsets"][l]):int(levels["offsets"][l + 1])]
        size = np.uint32(tab.shape[0])
        scale = f32(levels["scale"][l])
        res = np.uint32(int(np.ceil(scale)) + 1)
        p = (pos.astype(np.float64) * np.float64(scale) + 0.5).astype(f32)        # fmaf(x, scale, 0.5): exact product, one rounding
        c = np.floor(p).astype(np.int64)
        f = p - c.astype(f32)
        c = c.astype(np.uint32)
        acc = np.zeros((n, 2), f32)
        for idx in range(8):
            w = np.ones(n, f32)
            q = []
            for k in range(3):
                if idx & (1 << k):
                    w = w * f[:, k]
                    q.append(c[:, k] + np.uint32(1))
                else:
                    w = w * (f32(1) - f[:, k])
                    q.append(c[:, k])
            stride, index, dim = 1, np.zeros(n, np.uint32), 0                      # uint32 arithmetic as in grid_index
            with np.errstate(over="ignore"):
                while dim < 3 and stride <= int(size):
                    index = index + q[dim] * np.uint32(stride)
                    stride = (stride * int(res)) & 0xFFFFFFFF
                    dim += 1
                if int(size) < stride:
                    index = q[0] ^ (q[1] * np.uint32(19349663)) ^ (q[2] * np.uint32(83492791))
            e = index % size
            acc = acc + w[:, None] * tab[e]
        out[:, 2 * l:2 * l + 2] = acc
    return out


def sh_encode(dir01):
    """`kernel_sh` degree 4 (SphericalEncode.h:60-100)."""
    f32 = np.float32
    d = _f(dir01)
    x, y, z = d[:, 0] * f32(2) - f32(1), d[:, 1] * f32(2) - f32(1), d[:, 2] * f32(2) - f32(1)
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = np.empty((d.shape[0], 16), f32)
    o[:, 0] = f32(0.28209479177387814)
    o[:, 1] = f32(-0.48860251190291987) * y
    o[:, 2] = f32(0.48860251190291987) * z
    o[:, 3] = f32(-0.48860251190291987) * x
    o[:, 4] = f32(1.0925484305920792) * xy
    o[:, 5] = f32(-1.0925484305920792) * yz
    o[:, 6] = f32(0.94617469575755997) * z2 - f32(0.31539156525251999)
    o[:, 7] = f32(-1.0925484305920792) * xz
    o[:, 8] = f32(0.54627421529603959) * x2 - f32(0.54627421529603959) * y2
    o[:, 9] = f32(0.59004358992664352) * y * (f32(-3.0) * x2 + y2)
    o[:, 10] = f32(2.8906114426405538) * xy * z
    o[:, 11] = f32(0.45704579946446572) * y * (f32(1.0) - f32(5.0) * z2)
    o[:, 12] = f32(0.3731763325901154) * z * (f32(5.0) * z2 - f32(3.0))
    o[:, 13] = f32(0.45704579946446572) * x * (f32(1.0) - f32(5.0) * z2)
    o[:, 14] = f32(1.4453057213202769) * z * (x2 - y2)
    o[:, 15] = f32(0.59004358992664352) * x * (-x2 + f32(3.0) * y2)
    return o


def network(levels, scene, pos, dir01):
    """`NGPNetworks.execute_` (ngp_network.py:78-85), plain-Linear branch `:60-68` (fp16=False in Car.py / Easyship.py)."""
    enc = hash_encode(levels, scene["grid"], pos)
    h = np.maximum(enc @ _f(scene["density_mlp.0.weight"]).T, 0)
    den = h @ _f(scene["density_mlp.2.weight"]).T
    x = np.concatenate([den, sh_encode(dir01)], -1)
    h = np.maximum(x @ _f(scene["rgb_mlp.0.weight"]).T, 0)
    h = np.maximum(h @ _f(scene["rgb_mlp.2.weight"]).T, 0)
    rgb = h @ _f(scene["rgb_mlp.4.weight"]).T
    return np.concatenate([rgb, den[:, :1]], -1).astype(np.float32)


def unwarp_dt(dt):
    mx = MIN_CONE_STEPSIZE * np.float32(1 << (NERF_CASCADES - 1))
    return np.float32(dt) * (mx - MIN_CONE_STEPSIZE) + MIN_CONE_STEPSIZE


def composite(net_out, coords, numsteps, bg=(1.0, 1.0, 1.0)):
    """`compute_rgbs_inference` restated (python loop over rays; small inputs only)."""
    f32 = np.float32
    R = numsteps.shape[0]
    out = np.zeros((R, 3), f32)
    for i in range(R):
        n, base = int(numsteps[i, 0]), int(numsteps[i, 1])
        T, c, j = f32(1), np.zeros(3, f32), 0
        while j < n:
            if T < f32(1e-4):
                break
            o = net_out[base + j]
            dt = unwarp_dt(coords[base + j, 3])
            alpha = f32(1) - np.exp(-np.exp(o[3]) * dt, dtype=f32)
            w = alpha * T
            c = c + w * (f32(1) / (f32(1) + np.exp(-o[:3], dtype=f32)))
            T = T * (f32(1) - alpha)
            j += 1
        if j == n:
            c = c + T * _f(bg)
        out[i] = c
    return out


# ------------------------------------------------------------------ callers either side
def matrix_nerf2ngp(matrix, scale=NERF_SCALE, offset=(0.5, 0.5, 0.5), correct_pose=(-1, -1, 1)):
    """dataset.py:313-320 on a [3,4] (or [4,4]) camera-to-world matrix."""
    m = np.array(matrix, np.float32)[:3].copy()
    m[:, 0] *= correct_pose[0]
    m[:, 1] *= correct_pose[1]
    m[:, 2] *= correct_pose[2]
    m[:, 3] = m[:, 3] * np.float32(scale) + np.array(offset, np.float32)
    return m[[1, 2, 0]]


def generate_rays(xform, W, H, focal, principal=(0.5, 0.5)):
    """`generate_rays_total_test` (dataset.py:267-292) for one camera.  xform [3,4] in NGP convention; focal (fx, fy) in pixels."""
    f32 = np.float32
    gx = (np.linspace(0, H - 1, H, dtype=f32) + f32(0.5)) / f32(H)
    gy = (np.linspace(0, W - 1, W, dtype=f32) + f32(0.5)) / f32(W)
    a, b = np.meshgrid(gx, gy, indexing="ij")                        # jt.meshgrid: 'ij'
    xy = np.stack([a, b], -1).transpose(1, 0, 2).reshape(-1, 2)
    res = np.array([W, H], f32)
    d = np.concatenate([(xy - np.array(principal, f32)) * res / np.array(focal, f32), np.ones((H * W, 1), f32)], -1)
    d = (np.array(xform, f32)[:, :3] @ d[:, :, None])[:, :, 0]
    d = d / np.maximum(np.sqrt((d * d).sum(-1, keepdims=True)), f32(1e-12))
    o = np.broadcast_to(np.array(xform, f32)[:, 3], d.shape).copy()
    return o, d.astype(f32)


def render_img(scene, levels, rays_o, rays_d, aabb_scale, rng: Pcg32, n_rays_per_batch=4096, bg=(1.0, 1.0, 1.0), **march_kw):
    """`Runner.render_img` loop (runner.py:209-222): 4096-ray slabs, the tail padded with rays of ones; one RNG advance per slab."""
    R = rays_o.shape[0]
    img = np.empty((R + n_rays_per_batch, 3), np.float32)
    for p in range(0, R, n_rays_per_batch):
        o, d = rays_o[p:p + n_rays_per_batch], rays_d[p:p + n_rays_per_batch]
        if o.shape[0] < n_rays_per_batch:
            pad = n_rays_per_batch - o.shape[0]
            o = np.concatenate([o, np.ones((pad, 3), np.float32)])
            d = np.concatenate([d, np.ones((pad, 3), np.float32)])
        coords, _, numsteps, _, _ = sample(o, d, scene["density_grid_bitfield"], aabb_scale, rng.state, **march_kw)
        rng.advance()
        out = network_c(levels, scene, coords)
        rgb, _ = composite_c(out, coords, numsteps, bg)
        img[p:p + n_rays_per_batch] = rgb
    return img[:R]
