"""The reference's model variants that override the render path, on the same kernels as TensorVMSplit:

  * MLPRender_Fea_Ref, REFTensoRF      tensorf-myc/models/REFTensoRF.py:5-28, 64-256   (what configs/Scar.txt:28 trains)
  * Embedder, MLPNet, NerfPlusPlus     tensorf-myc/models/nerfplusplus.py:7-56, 66-140, 143-318
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .autograd_ops import _AppHFn, _FusedStepFn, _MarchFn, _RefMlpTrainFn, _f32c, _linear, _mlp3, _mlp_input, _stream_ptr
from .field import TensorVMSplit


class MLPRender_Fea_Ref(torch.nn.Module):
    """REFTensoRF.py:5-28: [dot_product, features, viewdirs, PE(features), PE(viewdirs)] -> Linear-ReLU-Linear-ReLU-Linear -> sigmoid.
    Parameters only; the arithmetic runs in the shade kernel of the owning REFTensoRF (k, the 1/rho argument, is unused there too)."""

    def __init__(self, inChanel, viewpe=6, feape=6, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * viewpe * 3 + 2 * feape * inChanel + 1 + 3 + inChanel
        self.viewpe, self.feape = viewpe, feape
        layer1 = torch.nn.Linear(self.in_mlpC, featureC)
        layer2 = torch.nn.Linear(featureC, featureC)
        layer3 = torch.nn.Linear(featureC, 3)
        self.mlp = torch.nn.Sequential(layer1, torch.nn.ReLU(), layer2, torch.nn.ReLU(), layer3)
        torch.nn.init.constant_(self.mlp[-1].bias, 0)
        self._owner = None

    def forward(self, pts, viewdirs, features, dot_product, k=None):
        if self._owner is None:
            raise L.TvrError("MLPRender_Fea_Ref is not attached to a REFTensoRF field (no packed weights on the device)")
        if torch.is_grad_enabled() and (features.requires_grad or viewdirs.requires_grad or dot_product.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            return self.forward_autograd(viewdirs, features, dot_product)
        return self._owner()._mlp_render_ref(viewdirs, features, dot_product)

    def forward_autograd(self, viewdirs, features, dot_product):
        return torch.sigmoid(_mlp3(self.mlp, _mlp_input(features, viewdirs, self.feape, self.viewpe, dot_product)))


class REFTensoRF(TensorVMSplit):
    """models/REFTensoRF.py:64-256 — the Ref-NeRF-style variant configs/Scar.txt trains: four extra Linears on the 144-wide plane*line
    product give a normal, a diffuse colour, a specular tint and a roughness; the MLP sees the reflection direction and -dot.
    Inference runs in the same fused HIP kernels (tvr_scene_desc.variant = 1); training as TensorVMSplit's (HIP march / gather
    kernels forward + backward, the small dense algebra under torch autograd)."""

    _variant = 1

    def __init__(self, aabb, gridSize, device, **kargs):
        super().__init__(aabb, gridSize, device, **kargs)
        self.norm_n_comp = self.density_n_comp                                                # :67
        self.penalty = torch.zeros((), device=self.device)                                    # :68

    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):      # :70-77
        if shadingMode != 'MLP_Fea':
            raise NotImplementedError(f"shadingMode {shadingMode!r}: only 'MLP_Fea' (MLPRender_Fea_Ref; what configs/Scar.txt uses) "
                                      "is on the accelerated render path")
        self.renderModule = MLPRender_Fea_Ref(self.app_dim, view_pe, fea_pe, featureC)
        import weakref
        self.renderModule._owner = weakref.ref(self)

    def init_svd_volume(self, res, device):                                                   # :80-96
        super().init_svd_volume(res, device)
        k = sum(self.app_n_comp)
        self.normal_linear = torch.nn.Linear(k, 3)
        self.diffuse_linear = torch.nn.Linear(k, 3)
        self.specular_linear = torch.nn.Linear(k, 1)
        self.rho_linear = torch.nn.Linear(k, 1)

    def _extra_linears(self):
        return [self.normal_linear, self.diffuse_linear, self.specular_linear, self.rho_linear]

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):            # :99-106
        grad_vars = super().get_optparam_groups(lr_init_spatialxyz, lr_init_network)
        grad_vars += [{'params': self.normal_linear.parameters(), 'lr': lr_init_network},
                      {'params': self.diffuse_linear.parameters(), 'lr': lr_init_network},
                      {'params': self.rho_linear.parameters(), 'lr': lr_init_network},
                      {'params': self.specular_linear.parameters(), 'lr': lr_init_network}]
        return grad_vars

    def _heads(self, h):                                                                      # :125-133
        return (_linear(self.basis_mat, h), _linear(self.diffuse_linear, h), torch.relu(_linear(self.specular_linear, h)),
                _linear(self.normal_linear, h), torch.relu(_linear(self.rho_linear, h)))

    def compute_appfeature(self, xyz_sampled):                                                # :107-133
        """-> (appfeatures [M,27], rgb_d [M,3], specular_tint [M,1], normal_vector [M,3], rho [M,1])"""
        sc = self._ensure_scene()
        x = _f32c(xyz_sampled, self.device).view(-1, 3)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._heads(_AppHFn.apply(self, x, *self.app_plane, *self.app_line))
        feats = torch.empty((x.shape[0], self.app_dim), dtype=torch.float32, device=self.device)
        extra = torch.empty((x.shape[0], 8), dtype=torch.float32, device=self.device)
        L.check(L.lib().tvr_app_feature_ref(sc, x.data_ptr(), x.shape[0], feats.data_ptr(), L.nbytes(feats), extra.data_ptr(), L.nbytes(extra),
                                             _stream_ptr(self.device)),
                "tvr_app_feature_ref")
        return feats, extra[:, 3:6], extra[:, 6:7], extra[:, 0:3], extra[:, 7:8]

    def _mlp_render(self, viewdirs, features):
        raise L.TvrError("REFTensoRF shades with MLPRender_Fea_Ref: call renderModule(pts, reflection, features, dot_product, k)")

    def _mlp_render_ref(self, viewdirs, features, dot_product):
        sc = self._ensure_scene()
        v = _f32c(viewdirs, self.device).view(-1, 3)
        f = _f32c(features, self.device).view(-1, self.app_dim)
        d = _f32c(dot_product, self.device).view(-1)
        if d.shape[0] != v.shape[0] or f.shape[0] != v.shape[0]:
            raise ValueError("viewdirs, features and dot_product must describe the same samples")
        out = torch.empty((v.shape[0], 3), dtype=torch.float32, device=self.device)
        L.check(L.lib().tvr_mlp_render_ref(sc, v.data_ptr(), f.data_ptr(), d.data_ptr(), v.shape[0], out.data_ptr(), L.nbytes(out), _stream_ptr(self.device)),
                "tvr_mlp_render_ref")
        return out

    @staticmethod
    def _normalize(x):
        """jt.normalize(x, dim=-1) (Jittor misc.py: x / sqrt(max(sum x^2, eps)), eps = 1e-30)"""
        return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-30))

    def render_rays_autograd(self, rays_chunk, white_bg=True, N_samples=-1, jitter=None):
        """REFTensoRF.execute with gradients (:174-256); also sets self.penalty (:240-243) for train.py:253-257."""
        rays = _f32c(rays_chunk, self.device)
        S = int(N_samples) if N_samples > 0 else self.nSamples
        eps_T = self.eps_T if self.eps_T is not None else float(self.rayMarch_weight_thres)
        if self._fused_step_ok() and not self._fused_step_outstanding():   # two C-ABI calls, no host read, fixed launch sequence (autograd_ops._FusedStepFn)
            mlp = self.renderModule.mlp
            rgb_map, depth, pen_ray = _FusedStepFn.apply(
                self, rays, jitter, S, eps_T, white_bg, *self.density_plane, *self.density_line, *self.app_plane, *self.app_line, self.basis_mat.weight,
                mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight, mlp[4].bias, self.normal_linear.weight, self.normal_linear.bias,
                self.diffuse_linear.weight, self.diffuse_linear.bias, self.specular_linear.weight, self.specular_linear.bias, self.rho_linear.weight, self.rho_linear.bias)
            self.penalty = pen_ray.sum()                                                      # :236-239: sum_i w_i relu(-dot_i)^2, summed per ray on the device
            return rgb_map, depth
        w, acc, xyz, ray_id, depth, _ = _MarchFn.apply(self, rays, jitter, S, eps_T, None, *self.density_plane, *self.density_line)
        h = _AppHFn.apply(self, xyz, *self.app_plane, *self.app_line)
        rm = self.renderModule
        if (self.fused_mlp_training and h.shape[0] > 0 and list(self.app_n_comp) == [48, 48, 48] and rm.feape == 2 and rm.viewpe == 2
                and rm.mlp[0].out_features == 128 and h.shape[0] * 576 < (1 << 32)):
            # heads + normalisation + reflection + MLPRender_Fea_Ref + the colour mix as one fused forward / backward pair (no library GEMM)
            mlp = rm.mlp
            rgb, in0 = _RefMlpTrainFn.apply(self, h, rays[ray_id, 3:6], self.basis_mat.weight, self.normal_linear.weight, self.normal_linear.bias,
                                            self.diffuse_linear.weight, self.diffuse_linear.bias, self.specular_linear.weight, self.specular_linear.bias,
                                            self.rho_linear.weight, self.rho_linear.bias, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias,
                                            mlp[4].weight, mlp[4].bias)
            penalty = torch.relu(in0).square()                                                # :237-238 (in0 = -dot_product)
        else:
            app_features, rgb_d, specular_tint, normal_vector, rho = self._heads(h)
            normal_vector = self._normalize(normal_vector)                                    # :217
            d = -rays[ray_id, 3:6]                                                            # :219
            dot_product = (d * normal_vector).sum(dim=1, keepdim=True)                        # :221-223
            reflection = 2 * dot_product * normal_vector - d                                  # :225
            rgb_s = self.renderModule.forward_autograd(reflection, app_features, -dot_product)    # :229
            rgb = specular_tint * rgb_s.clamp(min=0) + rgb_d                                  # :232
            penalty = torch.relu(-dot_product).square().squeeze(-1)                           # :237-238
        self.penalty = torch.sum(w * penalty, -1)                                             # :239
        rgb_map = torch.zeros((rays.shape[0], 3), device=self.device).index_add_(0, ray_id, w[:, None] * rgb)
        if white_bg:
            rgb_map = rgb_map + (1.0 - acc[:, None])
        return rgb_map.clamp(0, 1), depth

    def load_arrays(self, arrs):
        super().load_arrays(arrs)
        with torch.no_grad():
            for name, lin in zip(("normal", "diffuse", "specular", "rho"), self._extra_linears()):
                lin.weight.copy_(torch.as_tensor(arrs[f"{name}_W"]))
                lin.bias.copy_(torch.as_tensor(arrs[f"{name}_b"]))
        return self


class Embedder(torch.nn.Module):
    """nerfplusplus.py:7-56 with its defaults (log_sampling, include_input, (sin, cos))."""

    def __init__(self, input_dim, max_freq_log2, N_freqs):
        super().__init__()
        self.input_dim = input_dim
        self.out_dim = input_dim + input_dim * N_freqs * 2
        self.freq_bands = (2.0 ** torch.linspace(0.0, float(max_freq_log2), N_freqs)).tolist()

    def forward(self, input):
        out = [input]
        for freq in self.freq_bands:
            out += [torch.sin(input * freq), torch.cos(input * freq)]
        return torch.cat(out, dim=-1)


class MLPNet(torch.nn.Module):
    """nerfplusplus.py:66-140: the NeRF++ background network as torch modules.  On the HIP device NerfPlusPlus evaluates and trains it through its own kernels
    (tvr_mlpnet_forward / autograd_ops._BgNetFn); this `forward` is what remains for shapes those are not built for and for CPU-side host-logic checks."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_viewdirs=3, skips=(4,), use_viewdirs=False):
        super().__init__()
        self.input_ch, self.input_ch_viewdirs, self.use_viewdirs, self.skips = input_ch, input_ch_viewdirs, use_viewdirs, list(skips)
        layers, dim = [], input_ch
        for i in range(D):
            layers.append(torch.nn.Sequential(torch.nn.Linear(dim, W), torch.nn.ReLU()))
            dim = W
            if i in self.skips and i != (D - 1):
                dim += input_ch
        self.base_layers = torch.nn.ModuleList(layers)
        self.sigma_layers = torch.nn.Sequential(torch.nn.Linear(dim, 1))
        self.base_remap_layers = torch.nn.Sequential(torch.nn.Linear(dim, 256))
        self.rgb_layers = torch.nn.Sequential(torch.nn.Linear(256 + input_ch_viewdirs, W // 2), torch.nn.ReLU(), torch.nn.Linear(W // 2, 3),
                                              torch.nn.Sigmoid())

    def forward(self, input):
        # Linear layers through _LinearFn: library GEMMs forward / dX, weight and bias gradients by the deterministic tall-skinny
        # tvr_gemm_tn (M = rays x 512 rows; torch's column reductions and transposed GEMMs were 60 % of a NerfPlusPlus training step)
        lead = input.shape[:-1]
        x = input.reshape(-1, input.shape[-1])
        input_pts = x[:, :self.input_ch]
        base = torch.relu(_linear(self.base_layers[0][0], input_pts))
        for i in range(len(self.base_layers) - 1):
            if i in self.skips:
                base = torch.cat((input_pts, base), dim=-1)
            base = torch.relu(_linear(self.base_layers[i + 1][0], base))
        sigma = torch.abs(_linear(self.sigma_layers[0], base))
        base_remap = _linear(self.base_remap_layers[0], base)
        input_viewdirs = x[:, -self.input_ch_viewdirs:]
        h = torch.relu(_linear(self.rgb_layers[0], torch.cat((base_remap, input_viewdirs), dim=-1)))
        rgb = torch.sigmoid(_linear(self.rgb_layers[2], h))
        return {'rgb': rgb.reshape(*lead, 3), 'sigma': sigma.reshape(*lead)}


class NerfPlusPlus(TensorVMSplit):
    """models/nerfplusplus.py:143-318 — TensorVMSplit foreground inside a bounding sphere + a NeRF++ inverted-sphere background MLP.
    The foreground is the fused HIP path with EXPLICIT sample depths (tvr_render_z / tvr_march_*_z): NerfPlusPlus.sample_ray spaces the
    samples between `near` and the sphere and perturbs every one of them (also at evaluation: the reference's own behaviour).  The
    background (Embedder + MLPNet over 512 samples per ray) is plain torch.  `rand_fg` / `rand_bg` inject the two jt.rand_like draws."""

    HUGE_NUMBER, TINY_NUMBER, BG_SAMPLES = 1e10, 1e-6, 512                                    # :4-5, :284
    max_render_chunk = 655360     # rays per merged inference call (renderer): the background holds [rays, 512, ~20] fp32 temporaries — 24 GB for a whole 800x800 frame of the 288 GB
                                  # this card has (round 6: 103.3 -> 98.7 ms per frame against 65 536-ray calls, which also stayed below tvr_render's pieces; lower it on smaller cards)

    render_rays_is_the_frame = False          # forward() adds the background; render_rays alone is the foreground field (render.FrameStream refuses this model)

    def scene_settled(self) -> bool:
        """Never: the model keeps one set of [rays, 512, *] background temporaries and composes its picture in forward(), so two frames of this model are not put
        in flight at once (render.FrameStream then renders them one after the other).  (The background kernel's ticket word is per stream since round 6: _bg_work.)"""
        return False

    def __init__(self, aabb, gridSize, device, **kargs):
        super().__init__(aabb, gridSize, device, **kargs)
        self.bg_net = None

    def set_nerfplusplus(self, bg_freq=4, bg_view_freq=2, bg_D=4, radii=20):                  # :147-163
        self.bg_freq, self.bg_view_freq, self.radii, self.bg_D = bg_freq, bg_view_freq, radii, bg_D
        self.bg_embedder_position = Embedder(input_dim=4, max_freq_log2=bg_freq - 1, N_freqs=bg_freq)
        self.bg_embedder_viewdir = Embedder(input_dim=3, max_freq_log2=bg_view_freq - 1, N_freqs=bg_view_freq)
        self.bg_net = MLPNet(D=bg_D, W=128, skips=[int(bg_D / 2)], input_ch=self.bg_embedder_position.out_dim,
                             input_ch_viewdirs=self.bg_embedder_viewdir.out_dim, use_viewdirs=True).to(self.device)

    def get_kwargs(self):                                                                     # :165-171
        kwargs = super().get_kwargs()
        kwargs.update({'bg_freq': self.bg_freq, 'bg_view_freq': self.bg_view_freq, 'bg_D': self.bg_D, 'radii': self.radii})
        return kwargs

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):            # :173-176
        grad_vars = super().get_optparam_groups(lr_init_spatialxyz, lr_init_network)
        grad_vars += [{'params': self.bg_net.parameters(), 'lr': lr_init_network}]
        return grad_vars

    def _param_list(self):                      # the bg network is not part of the packed scene
        return super()._param_list()

    def intersect_sphere(self, ray_o, ray_d, radii):                                          # :178-194
        d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
        p = ray_o + d1.unsqueeze(-1) * ray_d
        ray_d_cos = 1. / torch.norm(ray_d, dim=-1)
        p_norm_sq = torch.sum(p * p, dim=-1)
        if (p_norm_sq >= radii).any():
            raise Exception('Not all your cameras are bounded by the unit sphere; please make sure the cameras are normalized properly!')
        d2 = torch.sqrt(radii - p_norm_sq) * ray_d_cos
        return d1 + d2

    def perturb_samples(self, z_vals, t_rand=None):                                           # :196-205
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], dim=-1)
        lower = torch.cat([z_vals[..., 0:1], mids], dim=-1)
        t_rand = torch.rand_like(z_vals) if t_rand is None else t_rand.to(z_vals)
        return lower + (upper - lower) * t_rand

    def depth2pts_outside(self, ray_o, ray_d, depth, radii):                                  # :207-237
        d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
        p_mid = ray_o + d1.unsqueeze(-1) * ray_d
        p_mid_norm = torch.norm(p_mid, dim=-1)
        ray_d_cos = 1. / torch.norm(ray_d, dim=-1)
        d2 = torch.sqrt(radii * radii - p_mid_norm * p_mid_norm) * ray_d_cos
        p_sphere = ray_o + (d1 + d2).unsqueeze(-1) * ray_d
        rot_axis = torch.cross(ray_o, p_sphere, dim=-1)
        rot_axis = rot_axis / torch.norm(rot_axis, dim=-1, keepdim=True)
        phi = torch.asin(p_mid_norm / radii)
        theta = torch.asin(p_mid_norm * depth / (radii * radii))
        rot_angle = (phi - theta).unsqueeze(-1)
        p_sphere_new = p_sphere * torch.cos(rot_angle) + torch.cross(rot_axis, p_sphere, dim=-1) * torch.sin(rot_angle) + \
            rot_axis * torch.sum(rot_axis * p_sphere, dim=-1, keepdim=True) * (1. - torch.cos(rot_angle))
        pts = torch.cat((p_sphere_new, depth.unsqueeze(-1)), dim=-1)
        depth_real = radii / (depth + self.TINY_NUMBER) * torch.cos(theta) * ray_d_cos + d1
        return pts, depth_real

    def _fg_depths(self, rays_o, rays_d, N_samples, t_rand=None):                             # :239-256
        fg_far_depth = self.intersect_sphere(rays_o, rays_d, radii=self.radii * self.radii)
        near, far = self.near_far
        step = (fg_far_depth - near) / (N_samples - 1)
        # `jt.stack([near + i * step for i in range(N_samples)], -1)` (:247): the same fp32 multiply and add per element, in one pass
        i = torch.arange(N_samples, dtype=torch.float32, device=step.device)
        fg_depth = near + i * step.unsqueeze(-1)
        return self.perturb_samples(fg_depth, t_rand).contiguous()

    def sample_ray(self, rays_o, rays_d, is_train=True, N_samples=-1, t_rand=None):           # :239-269 (host form)
        N_samples = N_samples if N_samples > 0 else self.nSamples
        interpx = self._fg_depths(rays_o, rays_d, N_samples, t_rand)
        rays_pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
        aabb = self.aabb.to(rays_o.device)
        mask_outbbox = ((aabb[0] > rays_pts) | (rays_pts > aabb[1])).any(dim=-1)
        return rays_pts, interpx, ~mask_outbbox

    def _render_z(self, rays, z_vals, S, eps_T):
        sc, lib = self._ensure_scene(), L.lib()
        self._settle_range_check()
        self._settle_arith(rays, S, False, eps_T)         # the foreground's reduced arithmetic, measured on these rays (uniform sampling: the network's arithmetic is what is probed)
        n = rays.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        depth = torch.empty((n,), dtype=torch.float32, device=self.device)
        lam = torch.empty((n,), dtype=torch.float32, device=self.device)
        scratch = self._get_scratch(lib.tvr_render_scratch_bytes(sc, n, S))
        L.check(lib.tvr_render_z(sc, rays.data_ptr(), n, S, 0, z_vals.data_ptr(), float(eps_T), rgb.data_ptr(), depth.data_ptr(), lam.data_ptr(),
                                 scratch.data_ptr(), scratch.numel(), None, None, _stream_ptr(self.device)), "tvr_render_z")
        return rgb, depth, lam

    def _render_z_autograd(self, rays, z_vals, S, eps_T):
        w, acc, xyz, ray_id, depth, lam = _MarchFn.apply(self, rays, None, S, eps_T, z_vals, *self.density_plane, *self.density_line)
        h = self._app_h_autograd(xyz)
        rgb = self._shade_autograd(h, rays[ray_id, 3:6])
        rgb_map = torch.zeros((rays.shape[0], 3), device=self.device).index_add_(0, ray_id, w[:, None] * rgb)
        return rgb_map.clamp(0, 1), depth, lam                                                # white_bg=False (:276), tensorBase.py:527

    # -- the background network on the HIP kernel (tvr_mlpnet_*): inference only; training keeps the torch modules under autograd
    def _bg_kernel_desc(self):
        net = self.bg_net
        ok = (net is not None and isinstance(net, MLPNet) and net.use_viewdirs and len(net.skips) == 1 and 2 <= len(net.base_layers) <= 4
              and net.base_layers[0][0].out_features == 128 and self.bg_view_freq == 2 and 1 <= self.bg_freq <= 4)
        if not ok:
            return None
        # (arith: the inference call follows the model's `mlp_arith` once _settle_bg_arith has measured it on this network — until then, and if refused, fp32-class;
        #  the training forward computes fp32-class whatever it says)
        return L.MlpnetDesc(len(net.base_layers), 128, int(net.skips[0]), int(self.bg_freq), int(self.bg_view_freq), self.BG_SAMPLES, self._ARITH[getattr(self, "bg_arith_in_effect", "f32")])

    bg_arith_in_effect = "f32"
    bg_arith_max_diff = None

    def _settle_bg_arith(self, pts, vd, n_pts, z=None):
        """The gate of field.py::_settle_arith for the background network (tvr_mlpnet_desc.arith is a field of a stateless descriptor: the library cannot hold a
        validation state for it).  Once per parameter state: up to 128 of this call's own rays (65 536 samples) through the kernel in "f32" and in the requested
        mode.  With the samples' depths at hand (the inference path, _background_fused) what is compared is what the picture gets — the COMPOSITED background colour
        per ray; a bare _mlpnet call compares the raw outputs (rgb absolutely, sigma relative to max(|sigma|, 1)), which is stricter.  <= `mlp_arith_tol`: the mode
        is in effect; otherwise the kernel computes in "f32" and a RuntimeWarning says so."""
        if self.mlp_arith == "f32":
            self.bg_arith_in_effect = "f32"
            return
        sig = (tuple((p.data_ptr(), p._version) for p in self.bg_net.parameters()), self.mlp_arith, float(self.mlp_arith_tol))
        if getattr(self, "_bg_arith_sig", None) == sig:
            return
        N = self.BG_SAMPLES
        k = max(min(int(n_pts), 65536) // N, 1) * N                # whole rays: the kernel takes the view direction of sample i from ray i // N
        out = {}
        for mode in ("f32", self.mlp_arith):
            self.bg_arith_in_effect = mode
            desc = self._bg_kernel_desc()
            img = self._bg_packed(desc)
            rgb = torch.empty(k, 3, device=self.device)
            sigma = torch.empty(k, device=self.device)
            wk = self._bg_work()
            L.check(L.lib().tvr_mlpnet_forward(C.byref(desc), img.data_ptr(), img.numel(), pts.data_ptr(), vd.data_ptr(), k, rgb.data_ptr(), sigma.data_ptr(),
                                               wk.data_ptr(), wk.numel(), _stream_ptr(self.device)), "tvr_mlpnet_forward")
            if z is not None:
                col = torch.empty(k // N, 3, device=self.device)
                L.check(L.lib().tvr_npp_bg_composite(rgb.data_ptr(), sigma.data_ptr(), z.data_ptr(), k // N, N, col.data_ptr(), _stream_ptr(self.device)), "tvr_npp_bg_composite")
                out[mode] = (col,)
            else:
                out[mode] = (rgb, sigma)
        if z is not None:
            d = float((out[self.mlp_arith][0] - out["f32"][0]).abs().max())
        else:
            (r0, s0), (r1, s1) = out["f32"], out[self.mlp_arith]
            d = max(float((r1 - r0).abs().max()), float(((s1 - s0).abs() / s0.abs().clamp_min(1.0)).max()))
        self.bg_arith_max_diff = d
        ok = d <= float(self.mlp_arith_tol)                      # (a NaN compares false)
        self.bg_arith_in_effect = self.mlp_arith if ok else "f32"
        self._bg_arith_sig = sig
        if not ok:
            import warnings
            warnings.warn(f"mlp_arith={self.mlp_arith!r} REFUSED for the background network: {d:.3g} off the fp32-class arithmetic on {k} probe samples "
                          f"(tolerance {self.mlp_arith_tol:g}); its kernel computes in 'f32'", RuntimeWarning, stacklevel=3)

    def _bg_work(self):
        """The forward kernel's work buffer (its ticket word; include/tvr.h tvr_mlpnet_work_bytes): one per stream, since one launch at a time may use it."""
        key = int(_stream_ptr(self.device) or 0)
        pool = self.__dict__.setdefault("_bg_work_pool", {})
        if key not in pool:
            pool[key] = L.dev_bytes(L.lib().tvr_mlpnet_work_bytes(), self.device, zero=True, what="tvr_mlpnet work")
        return pool[key]

    def _bg_params_changed(self):
        """The background network's parameters were (or are about to be) rewritten without their version counters moving (fused optimizers, graph replays):
        the packed image is stale AND so is a reduced arithmetic's validation, which was measured on the old weights."""
        self._bg_sig = None
        self._bg_arith_sig = None
        self.bg_arith_in_effect = "f32"

    def _bg_packed(self, desc):
        """Fragment image of the background network.  `base_remap_layers` (Linear 128->256, no activation) is folded into the first rgb
        layer here, in fp64: W_eff = W_rgb0[:, :256] @ W_remap, b_eff = W_rgb0[:, :256] @ b_remap + b_rgb0."""
        net = self.bg_net
        ps = list(net.parameters())
        sig = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_bg_sig", None) == sig and self._bg_image is not None:
            return self._bg_image
        f = lambda t: t.detach().to(self.device, torch.float32).contiguous()
        W0, b0 = net.rgb_layers[0].weight.detach().double(), net.rgb_layers[0].bias.detach().double()
        Wr, br = net.base_remap_layers[0].weight.detach().double(), net.base_remap_layers[0].bias.detach().double()
        keep = {"rgbh_W_base": f(W0[:, :256] @ Wr), "rgbh_W_view": f(W0[:, 256:]), "rgbh_b": f(W0[:, :256] @ br + b0),
                "sigma_W": f(net.sigma_layers[0].weight), "sigma_b": f(net.sigma_layers[0].bias), "rgbo_W": f(net.rgb_layers[2].weight),
                "rgbo_b": f(net.rgb_layers[2].bias)}
        p = L.MlpnetParams()
        for i, layer in enumerate(net.base_layers):
            keep[f"W{i}"], keep[f"b{i}"] = f(layer[0].weight), f(layer[0].bias)
            p.base_W[i], p.base_b[i] = keep[f"W{i}"].data_ptr(), keep[f"b{i}"].data_ptr()
        for k in ("sigma_W", "sigma_b", "rgbh_W_base", "rgbh_W_view", "rgbh_b", "rgbo_W", "rgbo_b"):
            setattr(p, k, keep[k].data_ptr())
        nbytes = L.lib().tvr_mlpnet_packed_bytes(C.byref(desc))
        if nbytes == 0:
            raise L.TvrError("tvr_mlpnet_packed_bytes: " + L.lib().tvr_last_error().decode(errors="replace"))
        img = L.dev_bytes(nbytes, self.device, what="tvr_mlpnet packed (inference)")
        L.check(L.lib().tvr_mlpnet_pack(C.byref(desc), C.byref(p), img.data_ptr(), nbytes, _stream_ptr(self.device)), "tvr_mlpnet_pack")
        self._bg_image, self._bg_sig = img, sig
        return img

    # ---- training through the HIP kernels (autograd_ops._BgNetFn) ----
    fused_bg_training = True
    # max |incoming gradient| * scale ~ this (a power of two on the device): the background backward's products then run on the fp16-split MFMAs.  16 leaves
    # 2^12 of growth through the layers below fp16's 65 504; a saturated step raises the model's flag (check_training_faults lowers grad_scale_target — the
    # foreground's — and this one alike).  None: the fp32-input MFMAs, no scale.
    bg_grad_scale_target = 16.0

    @staticmethod
    def _bg_layer_inputs(desc):
        """(reads the previous activations, reads the point embedding) per base layer — MLPNet.__init__'s `if i in skips and i != D-1: dim += input_ch`."""
        return [(l > 0, l == 0 or (l - 1 == desc.skip and l - 1 != desc.D - 1)) for l in range(desc.D)]

    def _bg_net_params(self):
        net = self.bg_net
        ps = []
        for layer in net.base_layers:
            ps += [layer[0].weight, layer[0].bias]
        return ps + [net.sigma_layers[0].weight, net.sigma_layers[0].bias, net.base_remap_layers[0].weight, net.base_remap_layers[0].bias,
                     net.rgb_layers[0].weight, net.rgb_layers[0].bias, net.rgb_layers[2].weight, net.rgb_layers[2].bias]

    def _bg_train_state(self, desc, P):
        """Persistent buffers of the training path: the folded first rgb layer, [W_eff; w_sigma] for the backward, and the fragment image whose block table
        points at the parameter tensors themselves (optimizers update those in place: the image is re-packed every step without a host sync)."""
        key = tuple(p.data_ptr() for p in P) + (desc.D, desc.skip, desc.pos_freqs)
        st = getattr(self, "_bg_tstate", None)
        if st is None or st["key"] != key:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)
            nbytes = L.lib().tvr_mlpnet_packed_bytes(C.byref(desc))
            if nbytes == 0:
                raise L.TvrError("tvr_mlpnet_packed_bytes: " + L.lib().tvr_last_error().decode(errors="replace"))
            st = {"key": key, "W_eff": z(64, 128), "b_eff": z(64), "W_view": z(64, 15), "W_cat": z(72, 128),
                  "image": L.dev_bytes(nbytes, self.device, what="tvr_mlpnet packed (training)"), "packed": False, "P": P}
            p = L.MlpnetParams()
            for l in range(desc.D):
                p.base_W[l], p.base_b[l] = P[2 * l].data_ptr(), P[2 * l + 1].data_ptr()
            D = desc.D
            p.sigma_W, p.sigma_b = P[2 * D].data_ptr(), P[2 * D + 1].data_ptr()
            p.rgbh_W_base, p.rgbh_W_view, p.rgbh_b = st["W_eff"].data_ptr(), st["W_view"].data_ptr(), st["b_eff"].data_ptr()
            p.rgbo_W, p.rgbo_b = P[2 * D + 6].data_ptr(), P[2 * D + 7].data_ptr()
            st["params"] = p
            self._bg_tstate = st
        return st

    def _bg_train_pack(self, desc, st):
        img = st["image"]
        fn = L.lib().tvr_mlpnet_repack if st["packed"] else L.lib().tvr_mlpnet_pack
        L.check(fn(C.byref(desc), C.byref(st["params"]), img.data_ptr(), img.numel(), _stream_ptr(self.device)), "tvr_mlpnet_(re)pack")
        st["packed"] = True

    def _mlpnet(self, bg_pts, viewdirs):
        """`self.bg_net(cat(embed(pts), embed(viewdirs)))` for pts [n, N, 4] and per-ray viewdirs [n, 3]: dict(rgb [n,N,3], sigma [n,N])."""
        n, N = bg_pts.shape[:2]
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.bg_net.parameters())
        if training:
            self._bg_params_changed()  # an optimizer step follows; fused optimizers do not bump the version counters `_bg_packed` watches
        # torch modules: under autograd, for shapes the kernel is not built for, and for host-logic checks of this class on a CPU device
        # (the foreground has no such path: it raises without the GPU)
        desc = None if bg_pts.device.type != "cuda" else self._bg_kernel_desc()
        if training and desc is not None and N == self.BG_SAMPLES and self.fused_bg_training and all(
                p.is_contiguous() and p.dtype == torch.float32 for p in self._bg_net_params()):
            from .autograd_ops import _BgNetFn
            rgb, sigma = _BgNetFn.apply(self, desc, bg_pts, viewdirs, *self._bg_net_params())
            return {'rgb': rgb, 'sigma': sigma}
        if training:
            desc = None
        if desc is None or N != self.BG_SAMPLES:
            inp = torch.cat((self.bg_embedder_position(bg_pts), self.bg_embedder_viewdir(viewdirs.unsqueeze(-2).expand(n, N, 3))), dim=-1)
            return self.bg_net(inp)
        pts = bg_pts.detach().to(torch.float32).contiguous()
        vd = viewdirs.detach().to(torch.float32).contiguous()
        self._settle_bg_arith(pts, vd, n * N)
        desc = self._bg_kernel_desc()
        img = self._bg_packed(desc)
        rgb = torch.empty(n, N, 3, device=self.device)
        sigma = torch.empty(n, N, device=self.device)
        wk = self._bg_work()
        L.check(L.lib().tvr_mlpnet_forward(C.byref(desc), img.data_ptr(), img.numel(), pts.data_ptr(), vd.data_ptr(), n * N, rgb.data_ptr(), sigma.data_ptr(),
                                           wk.data_ptr(), wk.numel(), _stream_ptr(self.device)), "tvr_mlpnet_forward")
        return {'rgb': rgb, 'sigma': sigma}

    def _background_fused(self, ray_o, ray_d, rand_bg, desc):
        """`_background` in three launches: points (perturbed depths, inverted-sphere geometry, flip), network, compositing."""
        n, N, lib = ray_d.shape[0], self.BG_SAMPLES, L.lib()
        o, d = _f32c(ray_o, self.device), _f32c(ray_d, self.device)
        z_lin = torch.linspace(0., self.radii, N, device=self.device)
        t_rand = torch.rand(n, N, device=self.device) if rand_bg is None else _f32c(torch.as_tensor(rand_bg), self.device)
        pts = torch.empty(n, N, 4, device=self.device)
        z = torch.empty(n, N, device=self.device)
        st = _stream_ptr(self.device)
        L.check(lib.tvr_npp_bg_points(o.data_ptr(), d.data_ptr(), n, z_lin.data_ptr(), N, t_rand.data_ptr(), float(self.radii), pts.data_ptr(), z.data_ptr(), st),
                "tvr_npp_bg_points")
        vdn = (d / torch.norm(d, dim=-1, keepdim=True)).contiguous()
        if desc is not None and self.mlp_arith != "f32":
            self._settle_bg_arith(pts, vdn, n * N, z=z)           # the gate, on what the picture gets: the composited background colour of this call's first rays
        raw = self._mlpnet(pts, vdn)
        out = torch.empty(n, 3, device=self.device)
        L.check(lib.tvr_npp_bg_composite(raw['rgb'].data_ptr(), raw['sigma'].data_ptr(), z.data_ptr(), n, N, out.data_ptr(), st), "tvr_npp_bg_composite")
        return out

    def _background(self, ray_o, ray_d, rand_bg=None):                                        # :280-308
        n, N = ray_d.shape[0], self.BG_SAMPLES
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.bg_net.parameters())
        if training:
            self._bg_params_changed()
        if not training and ray_d.device.type == "cuda":
            desc = self._bg_kernel_desc()
            if desc is not None:
                return self._background_fused(ray_o, ray_d, rand_bg, desc)
        viewdirs = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
        bg_z_vals = torch.linspace(0., self.radii, N, device=self.device).view(1, N).expand(n, N)
        bg_z_vals = self.perturb_samples(bg_z_vals, rand_bg)
        bg_ray_o = ray_o.unsqueeze(-2).expand(n, N, 3)
        bg_ray_d = ray_d.unsqueeze(-2).expand(n, N, 3)
        bg_pts, _ = self.depth2pts_outside(bg_ray_o, bg_ray_d, bg_z_vals, radii=self.radii)
        # the reference flips the network INPUT along the sample axis (:296); flipping the points first is the same thing
        bg_pts = torch.flip(bg_pts, dims=[-2])
        bg_z_vals = torch.flip(bg_z_vals, dims=[-1])
        bg_dists = bg_z_vals[..., :-1] - bg_z_vals[..., 1:]
        bg_dists = torch.cat((bg_dists, self.HUGE_NUMBER * torch.ones_like(bg_dists[..., 0:1])), dim=-1)
        bg_raw = self._mlpnet(bg_pts, viewdirs)
        bg_alpha = 1. - torch.exp(-bg_raw['sigma'] * bg_dists)
        T = torch.cumprod(1. - bg_alpha + self.TINY_NUMBER, dim=-1)[..., :-1]
        T = torch.cat((torch.ones_like(T[..., 0:1]), T), dim=-1)
        bg_weights = bg_alpha * T
        return torch.sum(bg_weights.unsqueeze(-1) * bg_raw['rgb'], dim=-2)

    def forward(self, rays_chunk, white_bg=False, is_train=False, ndc_ray=False, N_samples=-1, additional_output=True,
                rand_fg=None, rand_bg=None):                                                  # :272-318
        if self.bg_net is None:
            raise L.TvrError("call set_nerfplusplus() first (train.py:45-54 does so right after constructing the model)")
        if ndc_ray:
            raise NotImplementedError("ndc_ray=True is outside the accelerated path")
        rays = _f32c(rays_chunk, self.device)
        S = int(N_samples) if N_samples > 0 else self.nSamples
        eps_T = self.eps_T if self.eps_T is not None else float(self.rayMarch_weight_thres)
        with torch.no_grad():
            z_vals = self._fg_depths(rays[:, :3], rays[:, 3:6], S, rand_fg)
        if is_train and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # an optimizer step follows.  Invalidate the packed background image here, not only where the background is evaluated: a batch
            # whose rays all have bg_lambda <= 0.1 skips it, and a fused Adam still moves bg_net by momentum without bumping `_version`
            self._bg_params_changed()
            rgb_map, depth_map, bg_lambda = self._render_z_autograd(rays, z_vals, S, eps_T)
        else:
            rgb_map, depth_map, bg_lambda = self._render_z(rays, z_vals, S, eps_T)
        bg_lambda = torch.where(bg_lambda > 0.1, bg_lambda, torch.zeros_like(bg_lambda))     # :311
        # rgb_map + bg_lambda * bg_rgb_map (:312-314).  Rays whose foreground transmittance is <= 0.1 get 0 * background in the reference: their
        # 512 background samples contribute neither to the picture nor to any gradient (the `where` passes a constant zero), so they are
        # not evaluated — at inference and in training alike.
        idx = torch.nonzero(bg_lambda > 0).squeeze(-1)
        if idx.numel():
            bg = self._background(rays[idx, :3], rays[idx, 3:6], None if rand_bg is None else torch.as_tensor(rand_bg, device=self.device)[idx])
            rgb_map = rgb_map.index_add(0, idx, bg_lambda[idx].unsqueeze(-1) * bg)
        return rgb_map, depth_map

    execute = forward

    def load_arrays(self, arrs):
        super().load_arrays(arrs)
        self.set_nerfplusplus(int(arrs["bg.bg_freq"]), int(arrs["bg.bg_view_freq"]), int(arrs["bg.bg_D"]), float(arrs["bg.radii"]))
        sd = {k[len("bg_net."):]: torch.as_tensor(v) for k, v in arrs.items() if k.startswith("bg_net.")}
        self.bg_net.load_state_dict(sd)
        return self
