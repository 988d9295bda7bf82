"""Seeded synthetic TensorVMSplit scenes — numpy only, no device code.

The reference's own initialisation (0.1*randn with density_shift=-10, tensoRF.py:148-149) renders
nothing (softplus(-10) ~ 4.5e-5), so benchmarks and parity tests use a soft opaque blob instead:
    density plane_i[c] = |N(0,1)| * g(u) * g(v),  density line_i[c] = |N(0,1)| * g(w),
    g(s) = exp(-s^2 / (2*0.35^2)) on normalised coordinates s in [-1,1]
so sigma_feature ~ 30.6*exp(-r^2/(2*0.35^2)) (r = normalised radius): ~30 at the centre, < 5 outside
r ~ 0.67.  Appearance factors are 0.1*N(0,1) (the reference's init scale), basis / MLP weights
U(+-1/sqrt(fan_in)), b3 = 0 (tensorBase.py:74).  Parameter shapes follow tensoRF.py:154-164.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)
SEED = 20211202          # tensorf-myc/train.py:396


def _g(n: int, sigma: float) -> np.ndarray:
    s = np.linspace(-1.0, 1.0, n)
    return np.exp(-s * s / (2 * sigma * sigma))


def make_scene_arrays(gridSize: Sequence[int], aabb, seed: int = SEED,
                      density_n_comp=(16, 16, 16), appearance_n_comp=(48, 48, 48), app_dim: int = 27,
                      featureC: int = 128, view_pe: int = 2, fea_pe: int = 2, blob_sigma: float = 0.35,
                      alpha_grid: Optional[Sequence[int]] = None, ref: bool = False, npp: Optional[float] = None) -> Dict[str, np.ndarray]:
    """Return a flat dict of fp32 arrays in the reference's parameter layout.  ref=True adds REFTensoRF's parameters
    (models/REFTensoRF.py:86-96 and the 151-input W1 of MLPRender_Fea_Ref) from a second generator, so that the arrays shared
    with the TensorVMSplit scene of the same seed are identical.  npp=<radii> adds NerfPlusPlus's background network
    (models/nerfplusplus.py:147-163: bg_freq 4, bg_view_freq 2, bg_D 4, W 128) from a third generator."""
    rng = np.random.default_rng(seed)
    g = [int(x) for x in gridSize]
    out: Dict[str, np.ndarray] = {"aabb": np.asarray(aabb, np.float32).reshape(2, 3), "gridSize": np.asarray(g, np.int32)}
    gs = [_g(n, blob_sigma) for n in g]
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        v = VEC_MODE[i]
        c = density_n_comp[i]
        pl = np.abs(rng.standard_normal((1, c, g[m1], g[m0]))) * gs[m1][None, None, :, None] * gs[m0][None, None, None, :]
        ln = np.abs(rng.standard_normal((1, c, g[v], 1))) * gs[v][None, None, :, None]
        out[f"density_plane.{i}"] = pl.astype(np.float32)
        out[f"density_line.{i}"] = ln.astype(np.float32)
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        v = VEC_MODE[i]
        c = appearance_n_comp[i]
        out[f"app_plane.{i}"] = (0.1 * rng.standard_normal((1, c, g[m1], g[m0]))).astype(np.float32)
        out[f"app_line.{i}"] = (0.1 * rng.standard_normal((1, c, g[v], 1))).astype(np.float32)

    def U(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-b, b, size=shape).astype(np.float32)

    K = int(sum(appearance_n_comp))
    nin = app_dim + 3 + 2 * fea_pe * app_dim + 2 * view_pe * 3
    out["basis_mat"] = U((app_dim, K), K)
    # basis output feeds sin/cos(f*2^k): scale it up so the PE terms are exercised over a few radians
    out["basis_mat"] *= np.float32(64.0)
    out["W1"], out["b1"] = U((featureC, nin), nin), U((featureC,), nin)
    out["W2"], out["b2"] = U((featureC, featureC), featureC), U((featureC,), featureC)
    out["W3"], out["b3"] = U((3, featureC), featureC) * np.float32(4.0), np.zeros((3,), np.float32)
    if ref:
        r2 = np.random.default_rng(seed + 1)

        def U2(shape, fan_in, scale=1.0):
            b = 1.0 / np.sqrt(fan_in)
            return (r2.uniform(-b, b, size=shape) * scale).astype(np.float32)
        out["W1"], out["b1"] = U2((featureC, nin + 1), nin + 1), U2((featureC,), nin + 1)
        # h = plane*line is O(1e-2): scale the heads like the basis so that normal / colour / tint vary over the scene
        out["normal_W"], out["normal_b"] = U2((3, K), K, 64.0), U2((3,), K)
        out["diffuse_W"], out["diffuse_b"] = U2((3, K), K, 16.0), U2((3,), K) + np.float32(0.3)
        out["specular_W"], out["specular_b"] = U2((1, K), K, 64.0), U2((1,), K) + np.float32(0.5)
        out["rho_W"], out["rho_b"] = U2((1, K), K, 64.0), U2((1,), K) + np.float32(0.5)
    if npp is not None:
        r3 = np.random.default_rng(seed + 2)

        def U3(shape, fan_in, scale=1.0):
            b = 1.0 / np.sqrt(fan_in)
            return (r3.uniform(-b, b, size=shape) * scale).astype(np.float32)
        bg_freq, bg_view_freq, bg_D, W = 4, 2, 4, 128
        ch_p, ch_v = 4 + 4 * 2 * bg_freq, 3 + 3 * 2 * bg_view_freq
        out.update({"bg.radii": np.float32(npp), "bg.bg_freq": np.int32(bg_freq), "bg.bg_view_freq": np.int32(bg_view_freq), "bg.bg_D": np.int32(bg_D)})
        dim = ch_p
        for i in range(bg_D):
            out[f"bg_net.base_layers.{i}.0.weight"], out[f"bg_net.base_layers.{i}.0.bias"] = U3((W, dim), dim), U3((W,), dim)
            dim = W + (ch_p if (i == int(bg_D / 2) and i != bg_D - 1) else 0)
        out["bg_net.sigma_layers.0.weight"], out["bg_net.sigma_layers.0.bias"] = U3((1, dim), dim, 20.0), U3((1,), dim)   # visible background density
        out["bg_net.base_remap_layers.0.weight"], out["bg_net.base_remap_layers.0.bias"] = U3((256, dim), dim), U3((256,), dim)
        out["bg_net.rgb_layers.0.weight"], out["bg_net.rgb_layers.0.bias"] = U3((W // 2, 256 + ch_v), 256 + ch_v), U3((W // 2,), 256 + ch_v)
        out["bg_net.rgb_layers.2.weight"], out["bg_net.rgb_layers.2.bias"] = U3((3, W // 2), W // 2, 4.0), U3((3,), W // 2)
    if alpha_grid is not None:
        ag = [int(x) for x in alpha_grid]                      # (gx, gy, gz); volume stored (gz, gy, gx)
        zs, ys, xs = [np.linspace(-1, 1, n) for n in (ag[2], ag[1], ag[0])]
        r2 = zs[:, None, None] ** 2 + ys[None, :, None] ** 2 + xs[None, None, :] ** 2
        sf = 30.6 * np.exp(-r2 / (2 * blob_sigma ** 2))
        out["alpha_volume"] = (sf > 0.5).astype(np.float32)    # generous (dilated) occupancy of the blob
        out["alpha_aabb"] = out["aabb"].copy()
    return out


# BASELINE.json configs: hyper-parameters that go with the arrays.
SCENE_A = dict(gridSize=[300, 300, 300], aabb=[[-1.5] * 3, [1.5] * 3], near_far=[2.0, 6.0], step_ratio=0.5,
               cam_radius=4.0, camera_angle_x=0.6911, N_samples=512, img_wh=(800, 800))
SCENE_B = dict(gridSize=[128, 128, 128], aabb=[[-5.0] * 3, [5.0] * 3], near_far=[5.0, 40.0], step_ratio=0.5,
               cam_radius=13.0, camera_angle_x=0.6911, N_samples=192, img_wh=(64, 64))
HYPER = dict(density_shift=-10.0, distance_scale=25.0, rayMarch_weight_thres=1e-4, fea2denseAct="softplus",
             view_pe=2, fea_pe=2)


# ---------------------------------------------------------------------------------------------------------------------
# Alt path (BASELINE configs[4]): a seeded Instant-NGP scene in JNeRF's layout
def _expand_bits(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(x, y, z) -> np.ndarray:
    """Morton code of integer cell coordinates < 1024 (the order JNeRF keeps its 128^3 density grid in)."""
    with np.errstate(over="ignore"):
        return _expand_bits(x) | (_expand_bits(y) << np.uint32(1)) | (_expand_bits(z) << np.uint32(2))


def make_ngp_scene_arrays(level_offsets: Sequence[int], seed: int = SEED, shell=(0.12, 0.30), speckle: float = 0.02,
                          density_gain: float = 24.0) -> Dict[str, np.ndarray]:
    """fp32 arrays keyed like JNeRF's state dicts (`NGPNetworks` / `HashEncoder` / `DensityGridSampler`):
    `grid` (= pos_encoder.m_grid, level-major, 2 features per entry; level_offsets[-1] entries), the five bias-free Linear weights,
    and `density_grid` [5*128^3] (Morton order per cascade).  The reference's initialisation (grid U(+-1e-4)) renders nothing,
    so: grid U(-1,1); Linear weights U(+-sqrt(6/(fan_in+fan_out))), the density head scaled by `density_gain` so that raw
    densities spread over several e-folds; occupancy = a hollow ball around the scene centre (0.5,0.5,0.5), radii `shell` in
    unit-cube units, plus a `speckle` fraction of random cells inside radius 0.45 (cascade 0; coarser cascades carry the same
    geometry at their own cell size)."""
    rng = np.random.default_rng(seed + 5)
    out: Dict[str, np.ndarray] = {}
    out["grid"] = rng.uniform(-1.0, 1.0, size=int(level_offsets[-1]) * 2).astype(np.float32)

    def U(n_out, n_in, gain=1.0):
        b = np.sqrt(6.0 / (n_in + n_out))
        return (rng.uniform(-b, b, size=(n_out, n_in)) * gain).astype(np.float32)
    out["density_mlp.0.weight"] = U(64, 32)
    out["density_mlp.2.weight"] = U(16, 64, density_gain)
    out["rgb_mlp.0.weight"] = U(64, 32, 0.25)
    out["rgb_mlp.2.weight"] = U(64, 64)
    out["rgb_mlp.4.weight"] = U(3, 64, 4.0)
    G = 128
    ax = np.arange(G, dtype=np.uint32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    code = morton3d(X, Y, Z).reshape(-1)
    dg = np.zeros(5 * G ** 3, np.float32)
    for c in range(5):
        size = float(1 << c)
        ctr = [(A.astype(np.float32) + 0.5) / G * size + (0.5 - size / 2) for A in (X, Y, Z)]
        r = np.sqrt((ctr[0] - 0.5) ** 2 + (ctr[1] - 0.5) ** 2 + (ctr[2] - 0.5) ** 2).reshape(-1)
        occ = (r >= shell[0]) & (r <= shell[1])
        if c == 0 and speckle > 0:
            occ |= (r < 0.45) & (rng.random(G ** 3) < speckle)
        lvl = np.zeros(G ** 3, np.float32)
        lvl[code] = occ.astype(np.float32)
        dg[c * G ** 3:(c + 1) * G ** 3] = lvl
    out["density_grid"] = dg
    return out
