"""CPU oracle (a): op-for-op torch-CPU restatement of the TensoRF render path.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it,
and only as the checker / CPU baseline.  The shipped path is the HIP library.

PARITY UNPINNED.  The reference (FREDZEL2020/jittor-MYC-NeRFs, tensorf-myc) is
Python over Jittor; Jittor is not installed, not vendored and cannot be fetched,
and the reference holds no tests / golden vectors for this path (SURVEY.md §4,
§8c).  This file restates the reference arithmetic from the source text; it is
cross-checked against an independent scalar-C restatement (oracle/tvr_oracle.c)
and uses torch's grid_sample (align_corners=True, zeros padding), which has the
semantics the reference's F.grid_sample calls rely on.

Every function cites the reference lines (relative to /root/reference/) that it
follows.  All arithmetic is fp32 on CPU.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

MAT_MODE = ((0, 1), (0, 2), (1, 2))   # tensorf-myc/models/tensorBase.py:168
VEC_MODE = (2, 1, 0)                  # tensorf-myc/models/tensorBase.py:169


def _t(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.detach().to("cpu", dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


class OracleScene:
    """Plain container for everything TensorBase.__init__ / TensorVMSplit hold.

    Parameter shapes follow tensorf-myc/models/tensoRF.py:154-164
    (planes (1,C,grid[mat1],grid[mat0]), lines (1,C,grid[vec],1)) and the
    hyper-parameter set of tensorf-myc/models/tensorBase.py:141-145.
    """

    def __init__(self, aabb, gridSize, density_plane, density_line, app_plane, app_line,
                 basis_mat, mlp, near_far=(2.0, 6.0), step_ratio=0.5, density_shift=-10.0,
                 distance_scale=25.0, rayMarch_weight_thres=1e-4, fea2denseAct="softplus",
                 view_pe=2, fea_pe=2, alpha_volume=None, alpha_aabb=None, ref=None, npp=None, jittor_semantics=False):
        self.aabb = _t(aabb).reshape(2, 3)
        self.gridSize = [int(g) for g in gridSize]
        self.density_plane = [_t(p) for p in density_plane]
        self.density_line = [_t(p) for p in density_line]
        self.app_plane = [_t(p) for p in app_plane]
        self.app_line = [_t(p) for p in app_line]
        self.basis_mat = _t(basis_mat)                      # [app_dim, sum(app_n_comp)]
        self.mlp = {k: _t(v) for k, v in mlp.items()}       # W1,b1,W2,b2,W3,b3 (Linear: y = x W^T + b)
        self.near_far = (float(near_far[0]), float(near_far[1]))
        # True: raw2alpha / softplus in the formulations Jittor is believed to use instead of torch's — a MODEL of the unverifiable part
        self.jittor_semantics = bool(jittor_semantics)
        self.step_ratio = float(step_ratio)
        self.density_shift = float(density_shift)
        self.distance_scale = float(distance_scale)
        self.thres = float(rayMarch_weight_thres)
        self.fea2denseAct = fea2denseAct
        self.view_pe = int(view_pe)
        self.fea_pe = int(fea_pe)
        self.alpha_volume = None if alpha_volume is None else _t(alpha_volume)
        self.alpha_aabb = None if alpha_aabb is None else _t(alpha_aabb).reshape(2, 3)
        # REFTensoRF (tensorf-myc/models/REFTensoRF.py:86-96): {normal,diffuse,specular,rho}_{W,b}; mlp["W1"] is then [128,151]
        self.ref = None if ref is None else {k: _t(v) for k, v in ref.items()}
        self.penalty = torch.zeros(())
        # NerfPlusPlus (tensorf-myc/models/nerfplusplus.py:147-163): {"radii", "bg_freq", "bg_view_freq", "bg_D", "net": {name: array}} with
        # the MLPNet parameters under their module names (base_layers.i.0.weight, sigma_layers.0.weight, base_remap_layers.0.weight, rgb_layers.0/2.*)
        self.npp = None
        if npp is not None:
            self.npp = dict(npp, net={k: _t(v) for k, v in npp["net"].items()})
        self.update_stepSize()

    # tensorf-myc/models/tensorBase.py:197-209
    def update_stepSize(self):
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        g = torch.tensor(self.gridSize, dtype=torch.int32)
        self.units = self.aabbSize / (g - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio            # fp32 scalar tensor
        self.aabbDiag = torch.sqrt(torch.sum(torch.pow(self.aabbSize, 2)))
        self.nSamples = int((self.aabbDiag / self.stepSize).item()) + 1


# tensorf-myc/models/tensorBase.py:9-15
def positional_encoding(positions, freqs):
    freq_bands = (2 ** torch.arange(freqs).float())
    pts = (positions[..., None] * freq_bands).reshape(positions.shape[:-1] + (freqs * positions.shape[-1],))
    return torch.cat([torch.sin(pts), torch.cos(pts)], dim=-1)


# tensorf-myc/models/tensorBase.py:17-24
def raw2alpha(sigma, dist, jittor_semantics=False):
    alpha = 1.0 - torch.exp(-sigma * dist)
    t = torch.cat([torch.ones(alpha.shape[0], 1), 1.0 - alpha + 1e-10], -1)
    if jittor_semantics:
        # Jittor 1.3.x is believed to evaluate jt.cumprod as exp(cumsum(log(x))) — not verifiable offline; this branch MODELS that
        # formulation so that a test can bound what it would move (tests/test_oracle.py::test_jittor_formulations_stay_inside_the_parity_bar)
        T = torch.exp(torch.cumsum(torch.log(t), -1))
    else:
        T = torch.cumprod(t, -1)
    weights = alpha * T[:, :-1]
    return alpha, weights, T[:, -1:]


def softplus_jittor(x, beta=1.0, threshold=20.0):
    """jt.nn.softplus is believed to be  log(1 + exp(min(beta x, threshold))) / beta + max(x - threshold / beta, 0)  — no log1p, no branch."""
    return torch.log(1.0 + torch.exp(torch.clamp(beta * x, max=threshold))) / beta + torch.clamp(x - threshold / beta, min=0.0)


# tensorf-myc/models/tensorBase.py:223-224
def normalize_coord(sc: OracleScene, xyz):
    return (xyz - sc.aabb[0]) * sc.invaabbSize - 1


# tensorf-myc/models/tensorBase.py:39-59 (AlphaGridMask)
def alpha_sample(sc: OracleScene, xyz):
    aabb = sc.alpha_aabb
    inv = 1.0 / (aabb[1] - aabb[0]) * 2
    q = (xyz - aabb[0]) * inv - 1
    if q.shape[0] == 0:
        return torch.zeros(0)
    vol = sc.alpha_volume.view(1, 1, *sc.alpha_volume.shape[-3:])
    return F.grid_sample(vol, q.view(1, -1, 1, 1, 3), align_corners=True).view(-1)


# tensorf-myc/models/tensorBase.py:340-360
def sample_ray(sc: OracleScene, rays_o, rays_d, N_samples=-1, jitter=None):
    N_samples = N_samples if N_samples > 0 else sc.nSamples
    near, far = sc.near_far
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    rate_a = (sc.aabb[1] - rays_o) / vec
    rate_b = (sc.aabb[0] - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).max(-1).values.clamp(min=near, max=far)
    rng = torch.arange(N_samples)[None].float()
    if jitter is not None:                       # is_train: one u ~ U[0,1) per ray (:351-353), injected
        rng = rng.repeat(rays_d.shape[-2], 1)
        rng = rng + _t(jitter).view(-1, 1)
    step = sc.stepSize * rng
    interpx = t_min[..., None] + step
    rays_pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
    mask_outbbox = ((sc.aabb[0] > rays_pts) | (rays_pts > sc.aabb[1])).any(dim=-1)
    return rays_pts, interpx, ~mask_outbbox, t_min


def _coords(xyz):
    # tensorf-myc/models/tensoRF.py:212-214
    plane = torch.stack((xyz[..., MAT_MODE[0]], xyz[..., MAT_MODE[1]], xyz[..., MAT_MODE[2]])).detach().view(3, -1, 1, 2)
    line = torch.stack((xyz[..., VEC_MODE[0]], xyz[..., VEC_MODE[1]], xyz[..., VEC_MODE[2]]))
    line = torch.stack((torch.zeros_like(line), line), dim=-1).detach().view(3, -1, 1, 2)
    return plane, line


# tensorf-myc/models/tensoRF.py:209-225
def compute_densityfeature(sc: OracleScene, xyz):
    plane, line = _coords(xyz)
    sigma_feature = torch.zeros((xyz.shape[0],))
    for i in range(3):
        p = F.grid_sample(sc.density_plane[i], plane[[i]], align_corners=True).view(-1, xyz.shape[0])
        l = F.grid_sample(sc.density_line[i], line[[i]], align_corners=True).view(-1, xyz.shape[0])
        sigma_feature = sigma_feature + torch.sum(p * l, dim=0)
    return sigma_feature


# tensorf-myc/models/tensoRF.py:228-244
def compute_appfeature(sc: OracleScene, xyz, return_h=False):
    plane, line = _coords(xyz)
    ps, ls = [], []
    for i in range(3):
        ps.append(F.grid_sample(sc.app_plane[i], plane[[i]], align_corners=True).view(-1, xyz.shape[0]))
        ls.append(F.grid_sample(sc.app_line[i], line[[i]], align_corners=True).view(-1, xyz.shape[0]))
    h = (torch.cat(ps) * torch.cat(ls)).T
    f = h @ sc.basis_mat.T
    return (f, h) if return_h else f


# tensorf-myc/models/tensorBase.py:444-448
def feature2density(sc: OracleScene, x):
    if sc.fea2denseAct == "softplus":
        if getattr(sc, "jittor_semantics", False):
            return softplus_jittor(x + sc.density_shift)
        return F.softplus(x + sc.density_shift)
    return F.relu(x)


# tensorf-myc/models/tensorBase.py:62-86 (MLPRender_Fea.execute)
def mlp_render_fea(sc: OracleScene, viewdirs, features, return_in=False):
    indata = [features, viewdirs]
    if sc.fea_pe > 0:
        indata += [positional_encoding(features, sc.fea_pe)]
    if sc.view_pe > 0:
        indata += [positional_encoding(viewdirs, sc.view_pe)]
    mlp_in = torch.cat(indata, dim=-1)
    m = sc.mlp
    h1 = torch.relu(mlp_in @ m["W1"].T + m["b1"])
    h2 = torch.relu(h1 @ m["W2"].T + m["b2"])
    rgb = torch.sigmoid(h2 @ m["W3"].T + m["b3"])
    return (rgb, mlp_in) if return_in else rgb


# tensorf-myc/models/REFTensoRF.py:107-133 (REFTensoRF.compute_appfeature)
def compute_appfeature_ref(sc: OracleScene, xyz):
    _, h = compute_appfeature(sc, xyz, return_h=True)
    r = sc.ref
    appfeatures = h @ sc.basis_mat.T
    normal_vector = h @ r["normal_W"].T + r["normal_b"]
    rgb_d = h @ r["diffuse_W"].T + r["diffuse_b"]
    specular_tint = torch.relu(h @ r["specular_W"].T + r["specular_b"])
    rho = torch.relu(h @ r["rho_W"].T + r["rho_b"])
    return appfeatures, rgb_d, specular_tint, normal_vector, rho


# jt.normalize(x, dim=-1) as REFTensoRF.py:217 calls it.  Jittor (third-party, not in /root/reference; jnerf-myc/requirements.txt
# pins jittor>=1.3.4.13) defines normalize(input, p=2, dim=1, eps=1e-30) = input / input.norm(p, dim, True, eps) with
# norm(p=2) = sqrt(max(sum(x^2), eps)) (python/jittor/misc.py): restated from that published definition.
def jt_normalize(x):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-30))


# tensorf-myc/models/REFTensoRF.py:5-28 (MLPRender_Fea_Ref.execute; k is accepted and unused there)
def mlp_render_fea_ref(sc: OracleScene, viewdirs, features, dot_product, k=None, return_in=False):
    indata = [dot_product, features, viewdirs]
    if sc.fea_pe > 0:
        indata += [positional_encoding(features, sc.fea_pe)]
    if sc.view_pe > 0:
        indata += [positional_encoding(viewdirs, sc.view_pe)]
    mlp_in = torch.cat(indata, dim=-1)
    m = sc.mlp
    h1 = torch.relu(mlp_in @ m["W1"].T + m["b1"])
    h2 = torch.relu(h1 @ m["W2"].T + m["b2"])
    rgb = torch.sigmoid(h2 @ m["W3"].T + m["b3"])
    return (rgb, mlp_in) if return_in else rgb


# tensorf-myc/models/REFTensoRF.py:212-239: the appearance branch of REFTensoRF.execute for the samples of app_mask
def shade_ref(sc: OracleScene, xyz_n_sel, views_sel, weight_sel):
    app_features, rgb_d, specular_tint, normal_vector, rho = compute_appfeature_ref(sc, xyz_n_sel)
    normal_vector = jt_normalize(normal_vector)                        # :217
    d = -views_sel                                                     # :219
    dot_product = (d * normal_vector).sum(dim=1)                       # :221
    dot_product = dot_product[:, None]                                 # :222-223 (broadcast to [M,1])
    reflection = 2 * dot_product * normal_vector - d                   # :225
    rgb_s = mlp_render_fea_ref(sc, reflection, app_features, -dot_product, 1 / rho)   # :229
    valid_rgbs = specular_tint * torch.clamp(rgb_s, min=0) + rgb_d     # :232
    penalty = torch.relu(-dot_product)                                 # :237
    penalty = (penalty * penalty).squeeze(-1)                          # :238
    sc.penalty = torch.sum(weight_sel * penalty, -1)                   # :239
    return valid_rgbs


# tensorf-myc/models/tensorBase.py:476-536 (ndc_ray=False branch); with sc.ref set, tensorf-myc/models/REFTensoRF.py:174-256,
# which differs only in the appearance branch (shade_ref)
def execute(sc: OracleScene, rays_chunk, white_bg=True, N_samples=-1, jitter=None, dump=False, sampler=None):
    rays_chunk = _t(rays_chunk)
    viewdirs = rays_chunk[:, 3:6]
    if sampler is not None:                         # a subclass's sample_ray override (NerfPlusPlus, nerfplusplus.py:239-269)
        xyz, z_vals, ray_valid = sampler(rays_chunk[:, :3], viewdirs)
        t_min = torch.zeros(rays_chunk.shape[0])
    else:
        xyz, z_vals, ray_valid, t_min = sample_ray(sc, rays_chunk[:, :3], viewdirs, N_samples, jitter)
    bbox_valid = ray_valid.clone()
    dists = torch.cat((z_vals[:, 1:] - z_vals[:, :-1], torch.zeros_like(z_vals[:, :1])), dim=-1)
    if dists.shape[0] != xyz.shape[0]:              # is_train=False: z_vals is [1,S] + [N,1] broadcast already
        dists = dists.expand(xyz.shape[0], -1)
    viewdirs_e = viewdirs.view(-1, 1, 3).expand(xyz.shape)

    if sc.alpha_volume is not None:                 # :491-496
        alphas = alpha_sample(sc, xyz[ray_valid])
        alpha_mask = alphas > 0
        ray_invalid = ~ray_valid
        ray_invalid[ray_valid] |= (~alpha_mask)
        ray_valid = ~ray_invalid

    sigma = torch.zeros(xyz.shape[:-1])
    rgb = torch.zeros((*xyz.shape[:2], 3))
    sigma_feature_full = torch.zeros(xyz.shape[:-1])
    xyz_n = normalize_coord(sc, xyz)                # :503 (done unconditionally here; values unused if none valid)
    if ray_valid.any():
        sf = compute_densityfeature(sc, xyz_n[ray_valid])
        sigma[ray_valid] = feature2density(sc, sf)
        sigma_feature_full[ray_valid] = sf

    alpha, weight, bg_weight = raw2alpha(sigma, dists * sc.distance_scale, getattr(sc, "jittor_semantics", False))
    app_mask = weight > sc.thres
    if app_mask.any():
        if sc.ref is not None:
            rgb[app_mask] = shade_ref(sc, xyz_n[app_mask], viewdirs_e[app_mask], weight[app_mask])
        else:
            app_features = compute_appfeature(sc, xyz_n[app_mask])
            rgb[app_mask] = mlp_render_fea(sc, viewdirs_e[app_mask], app_features)

    acc_map = torch.sum(weight, -1)
    rgb_map = torch.sum(weight[..., None] * rgb, -2)
    if white_bg:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    rgb_map = rgb_map.clamp(0, 1)
    depth_map = torch.sum(weight * z_vals, -1)
    depth_map = depth_map + (1.0 - acc_map) * rays_chunk[..., -1]   # :531 — last ray column (d_z), kept as is

    if not dump:
        return rgb_map, depth_map
    g = torch.tensor(sc.gridSize, dtype=torch.float32)
    fidx = ((xyz_n + 1) / 2) * (g - 1)             # grid_sample align_corners=True un-normalisation, per axis
    return dict(rgb_map=rgb_map, depth_map=depth_map, acc_map=acc_map, t_min=t_min,
                z_vals=z_vals.expand(xyz.shape[0], -1).contiguous(), xyz=xyz, xyz_norm=xyz_n,
                bbox_valid=bbox_valid, valid=ray_valid, cell=torch.floor(fidx).to(torch.int32),
                sigma_feature=sigma_feature_full, sigma=sigma, alpha=alpha, weight=weight,
                bg_weight=bg_weight, app_mask=app_mask, rgb=rgb)


# ---- NerfPlusPlus (tensorf-myc/models/nerfplusplus.py) -------------------------------------------------------------------------
HUGE_NUMBER = 1e10      # nerfplusplus.py:4
TINY_NUMBER = 1e-6      # nerfplusplus.py:5


# nerfplusplus.py:7-56 (Embedder, log_sampling=True, include_input=True, periodic_fns=(sin, cos))
def embed(x, max_freq_log2, n_freqs):
    freq_bands = (2.0 ** torch.linspace(0.0, float(max_freq_log2), n_freqs)).tolist()
    out = [x]
    for f in freq_bands:
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, dim=-1)


# nerfplusplus.py:66-140 (MLPNet with W=128, skips=[int(D/2)], use_viewdirs=True as set_nerfplusplus builds it, :158-161)
def mlpnet(net, D, skips, inp, input_ch, input_ch_viewdirs):
    lin = lambda name, x: x @ net[name + ".weight"].T + net[name + ".bias"]
    input_pts = inp[..., :input_ch]
    base = torch.relu(lin("base_layers.0.0", input_pts))
    for i in range(D - 1):
        if i in skips:
            base = torch.cat((input_pts, base), dim=-1)
        base = torch.relu(lin(f"base_layers.{i + 1}.0", base))
    sigma = torch.abs(lin("sigma_layers.0", base))
    base_remap = lin("base_remap_layers.0", base)
    input_viewdirs = inp[..., -input_ch_viewdirs:]
    h = torch.relu(lin("rgb_layers.0", torch.cat((base_remap, input_viewdirs), dim=-1)))
    rgb = torch.sigmoid(lin("rgb_layers.2", h))
    return rgb, sigma.squeeze(-1)


# nerfplusplus.py:178-194
def intersect_sphere(ray_o, ray_d, radii):
    d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
    p = ray_o + d1.unsqueeze(-1) * ray_d
    ray_d_cos = 1.0 / torch.norm(ray_d, dim=-1)
    p_norm_sq = torch.sum(p * p, dim=-1)
    if (p_norm_sq >= radii).any():
        raise Exception("Not all your cameras are bounded by the unit sphere; please make sure the cameras are normalized properly!")
    d2 = torch.sqrt(radii - p_norm_sq) * ray_d_cos
    return d1 + d2


# nerfplusplus.py:196-205; t_rand = jt.rand_like(z_vals), injected
def perturb_samples(z_vals, t_rand):
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], dim=-1)
    lower = torch.cat([z_vals[..., 0:1], mids], dim=-1)
    return lower + (upper - lower) * t_rand


# nerfplusplus.py:207-237
def depth2pts_outside(ray_o, ray_d, depth, radii):
    d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
    p_mid = ray_o + d1.unsqueeze(-1) * ray_d
    p_mid_norm = torch.norm(p_mid, dim=-1)
    ray_d_cos = 1.0 / torch.norm(ray_d, dim=-1)
    d2 = torch.sqrt(radii * radii - p_mid_norm * p_mid_norm) * ray_d_cos
    p_sphere = ray_o + (d1 + d2).unsqueeze(-1) * ray_d
    rot_axis = torch.cross(ray_o, p_sphere, dim=-1)
    rot_axis = rot_axis / torch.norm(rot_axis, dim=-1, keepdim=True)
    phi = torch.asin(p_mid_norm / radii)
    theta = torch.asin(p_mid_norm * depth / (radii * radii))
    rot_angle = (phi - theta).unsqueeze(-1)
    p_sphere_new = p_sphere * torch.cos(rot_angle) + torch.cross(rot_axis, p_sphere, dim=-1) * torch.sin(rot_angle) + \
        rot_axis * torch.sum(rot_axis * p_sphere, dim=-1, keepdim=True) * (1.0 - torch.cos(rot_angle))
    pts = torch.cat((p_sphere_new, depth.unsqueeze(-1)), dim=-1)
    depth_real = radii / (depth + TINY_NUMBER) * torch.cos(theta) * ray_d_cos + d1
    return pts, depth_real


# nerfplusplus.py:239-269 (NerfPlusPlus.sample_ray): N samples between `near` and the sphere of radius `radii`, every one perturbed
def sample_ray_npp(sc: OracleScene, rays_o, rays_d, N_samples, t_rand):
    radii = float(sc.npp["radii"])
    fg_far_depth = intersect_sphere(rays_o, rays_d, radii * radii)
    near, far = sc.near_far
    step = (fg_far_depth - near) / (N_samples - 1)
    fg_depth = torch.stack([near + i * step for i in range(N_samples)], dim=-1)
    interpx = perturb_samples(fg_depth, _t(t_rand))
    rays_pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
    mask_outbbox = ((sc.aabb[0] > rays_pts) | (rays_pts > sc.aabb[1])).any(dim=-1)
    return rays_pts, interpx, ~mask_outbbox


# nerfplusplus.py:272-318 (NerfPlusPlus.execute).  rand_fg [N,S] and rand_bg [N,bg_samples] are the two jt.rand_like draws;
# bg_samples is the literal 512 of :284.
def execute_npp(sc: OracleScene, rays_chunk, N_samples=-1, rand_fg=None, rand_bg=None, bg_samples=512, dump=False):
    rays_chunk = _t(rays_chunk)
    N_samples = N_samples if N_samples > 0 else sc.nSamples
    P = sc.npp
    radii = float(P["radii"])
    d = execute(sc, rays_chunk, white_bg=False, N_samples=N_samples, dump=True,
                sampler=lambda o, v: sample_ray_npp(sc, o, v, N_samples, rand_fg))          # :276 super().execute(rays_chunk, False, ...)
    rgb_map, depth_map, alpha = d["rgb_map"], d["depth_map"], d["alpha"]
    T = torch.cumprod(1.0 - alpha + TINY_NUMBER, dim=-1)
    bg_lambda = T[..., -1]
    ray_o, ray_d = rays_chunk[:, :3], rays_chunk[:, 3:6]
    ray_d_norm = torch.norm(ray_d, dim=-1, keepdim=True)
    viewdirs = ray_d / ray_d_norm
    n = ray_d.shape[0]
    bg_z_vals = torch.linspace(0.0, radii, bg_samples).view(1, bg_samples).expand(n, bg_samples)
    bg_z_vals = perturb_samples(bg_z_vals, _t(rand_bg))
    bg_ray_o = ray_o.unsqueeze(-2).expand(n, bg_samples, 3)
    bg_ray_d = ray_d.unsqueeze(-2).expand(n, bg_samples, 3)
    bg_viewdirs = viewdirs.unsqueeze(-2).expand(n, bg_samples, 3)
    bg_pts, _ = depth2pts_outside(bg_ray_o, bg_ray_d, bg_z_vals, radii)
    inp = torch.cat((embed(bg_pts, P["bg_freq"] - 1, P["bg_freq"]), embed(bg_viewdirs, P["bg_view_freq"] - 1, P["bg_view_freq"])), dim=-1)
    inp = torch.flip(inp, dims=[-2])
    bg_z_vals = torch.flip(bg_z_vals, dims=[-1])
    bg_dists = bg_z_vals[..., :-1] - bg_z_vals[..., 1:]
    bg_dists = torch.cat((bg_dists, HUGE_NUMBER * torch.ones_like(bg_dists[..., 0:1])), dim=-1)
    ch_p, ch_v = 4 + 4 * 2 * P["bg_freq"], 3 + 3 * 2 * P["bg_view_freq"]
    bg_rgb, bg_sigma = mlpnet(P["net"], P["bg_D"], [int(P["bg_D"] / 2)], inp, ch_p, ch_v)
    bg_alpha = 1.0 - torch.exp(-bg_sigma * bg_dists)
    T = torch.cumprod(1.0 - bg_alpha + TINY_NUMBER, dim=-1)[..., :-1]
    T = torch.cat((torch.ones_like(T[..., 0:1]), T), dim=-1)
    bg_weights = bg_alpha * T
    bg_rgb_map = torch.sum(bg_weights.unsqueeze(-1) * bg_rgb, dim=-2)
    bg_depth_map = torch.sum(bg_weights * bg_z_vals, dim=-1)
    bg_lambda = torch.where(bg_lambda > 0.1, bg_lambda, torch.zeros_like(bg_lambda))
    bg_rgb_map = bg_lambda.unsqueeze(-1) * bg_rgb_map
    bg_depth_map = bg_lambda * bg_depth_map
    rgb_map = rgb_map + bg_rgb_map
    if dump:
        return dict(rgb_map=rgb_map, depth_map=depth_map, fg_rgb_map=d["rgb_map"], bg_rgb_map=bg_rgb_map, bg_lambda=bg_lambda, z_vals=d["z_vals"],
                    valid=d["valid"], app_mask=d["app_mask"], weight=d["weight"], bg_pts=bg_pts)
    return rgb_map, depth_map


# tensorf-myc/renderer.py:12-27
def OctreeRender_trilinear_fast(rays, sc: OracleScene, chunk=4096, N_samples=-1, white_bg=True, jitter=None):
    rays = _t(rays)
    rgbs, depths = [], []
    n = rays.shape[0]
    for ci in range(n // chunk + int(n % chunk > 0)):
        sl = slice(ci * chunk, (ci + 1) * chunk)
        r, d = execute(sc, rays[sl], white_bg=white_bg, N_samples=N_samples,
                       jitter=None if jitter is None else _t(jitter)[sl])
        rgbs.append(r)
        depths.append(d)
    return torch.cat(rgbs), None, torch.cat(depths), None, None


# tensorf-myc/models/tensorBase.py:451-473
def compute_alpha(sc: OracleScene, xyz_locs, length=1.0):
    if sc.alpha_volume is not None:
        alpha_mask = alpha_sample(sc, xyz_locs) > 0
    else:
        alpha_mask = torch.ones_like(xyz_locs[:, 0]).bool()
    sigma = torch.zeros(xyz_locs.shape[:-1])
    if alpha_mask.any():
        sigma[alpha_mask] = feature2density(sc, compute_densityfeature(sc, normalize_coord(sc, xyz_locs[alpha_mask])))
    return (1 - torch.exp(-sigma * length)).view(xyz_locs.shape[:-1])


# tensorf-myc/models/tensorBase.py:366-383
def getDenseAlpha(sc: OracleScene, gridSize):
    samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, gridSize[0]), torch.linspace(0, 1, gridSize[1]),
                                         torch.linspace(0, 1, gridSize[2]), indexing="ij"), -1)
    dense_xyz = sc.aabb[0] * (1 - samples) + sc.aabb[1] * samples
    alpha = torch.zeros_like(dense_xyz[..., 0])
    for i in range(gridSize[0]):
        alpha[i] = compute_alpha(sc, dense_xyz[i].view(-1, 3), float(sc.stepSize)).view((gridSize[1], gridSize[2]))
    return alpha, dense_xyz


# tensorf-myc/models/tensorBase.py:385-409 — returns (binary alpha volume (gz,gy,gx), new_aabb)
def updateAlphaMask(sc: OracleScene, gridSize, alphaMask_thres):
    alpha, dense_xyz = getDenseAlpha(sc, list(gridSize))
    dense_xyz = dense_xyz.transpose(0, 2).contiguous()
    alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
    alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(list(gridSize)[::-1])
    alpha = (alpha >= alphaMask_thres).float()
    valid_xyz = dense_xyz[alpha > 0.5]
    return alpha, torch.stack((valid_xyz.amin(0), valid_xyz.amax(0)))


# tensorf-myc/models/tensorBase.py:411-441 — the mask filtering_rays applies (True = the ray is kept); both filters, chunked as the reference chunks
def filtering_rays_mask(sc: OracleScene, all_rays, N_samples=256, chunk=10240 * 5, bbox_only=False):
    all_rays = _t(all_rays)
    flat = all_rays.reshape(-1, all_rays.shape[-1])
    masks = []
    for idx_chunk in torch.split(torch.arange(flat.shape[0]), chunk):
        rays_chunk = flat[idx_chunk]
        rays_o, rays_d = rays_chunk[..., :3], rays_chunk[..., 3:6]
        if bbox_only:
            vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
            rate_a = (sc.aabb[1] - rays_o) / vec
            rate_b = (sc.aabb[0] - rays_o) / vec
            t_min = torch.minimum(rate_a, rate_b).max(-1).values          # no clamp to [near, far] here (:425-426)
            t_max = torch.maximum(rate_a, rate_b).min(-1).values
            mask_inbbox = t_max > t_min
        else:
            xyz_sampled, _, _, _ = sample_ray(sc, rays_o, rays_d, N_samples=N_samples)      # is_train=False: no jitter
            mask_inbbox = (alpha_sample(sc, xyz_sampled.reshape(-1, 3)).view(xyz_sampled.shape[:-1]) > 0).any(-1)   # ALL samples, no bbox mask (:430-431)
        masks.append(mask_inbbox)
    return torch.cat(masks).view(all_rays.shape[:-1])


def scene_from_arrays(arrs: Dict[str, np.ndarray], **hyper) -> OracleScene:
    """Build an OracleScene from the flat array dict jittor_myc_nerfs_amd.synthetic produces."""
    mlp = {k: arrs[k] for k in ("W1", "b1", "W2", "b2", "W3", "b3")}
    ref = None
    if "normal_W" in arrs:
        ref = {f"{n}_{s}": arrs[f"{n}_{s}"] for n in ("normal", "diffuse", "specular", "rho") for s in ("W", "b")}
    npp = None
    if "bg.radii" in arrs:
        npp = dict(radii=float(arrs["bg.radii"]), bg_freq=int(arrs["bg.bg_freq"]), bg_view_freq=int(arrs["bg.bg_view_freq"]), bg_D=int(arrs["bg.bg_D"]),
                   net={k[len("bg_net."):]: v for k, v in arrs.items() if k.startswith("bg_net.")})
    return OracleScene(ref=ref, npp=npp,
        aabb=arrs["aabb"], gridSize=arrs["gridSize"],
        density_plane=[arrs[f"density_plane.{i}"] for i in range(3)],
        density_line=[arrs[f"density_line.{i}"] for i in range(3)],
        app_plane=[arrs[f"app_plane.{i}"] for i in range(3)],
        app_line=[arrs[f"app_line.{i}"] for i in range(3)],
        basis_mat=arrs["basis_mat"], mlp=mlp,
        alpha_volume=arrs.get("alpha_volume"), alpha_aabb=arrs.get("alpha_aabb"), **hyper)
