"""ctypes binding of libtvr.so (include/tvr.h).  The product path has NO CPU fallback: if the HIP
library is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TVR_LIB_PATH") or os.path.join(_HERE, "lib", "libtvr.so")   # env override: A/B builds of the same ABI

_FP = C.POINTER(C.c_float)


class SceneDesc(C.Structure):
    _fields_ = [("aabb", C.c_float * 6), ("grid", C.c_int32 * 3), ("density_n_comp", C.c_int32 * 3),
                ("app_n_comp", C.c_int32 * 3), ("app_dim", C.c_int32), ("featureC", C.c_int32),
                ("view_pe", C.c_int32), ("fea_pe", C.c_int32), ("near_", C.c_float), ("far_", C.c_float),
                ("step_size", C.c_float), ("inv_aabb_size", C.c_float * 3), ("density_shift", C.c_float),
                ("distance_scale", C.c_float), ("weight_thres", C.c_float), ("fea2dense_act", C.c_int32),
                ("variant", C.c_int32)]


class SceneParams(C.Structure):
    _fields_ = [("density_plane", C.c_void_p * 3), ("density_line", C.c_void_p * 3),
                ("app_plane", C.c_void_p * 3), ("app_line", C.c_void_p * 3), ("basis_mat", C.c_void_p),
                ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("W3", C.c_void_p), ("b3", C.c_void_p), ("ref_W", C.c_void_p * 4), ("ref_b", C.c_void_p * 4)]


class ScratchLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("counter", "ray_off", "ray_cnt", "acc", "q_pos", "q_out", "q_ray", "q_j", "total")]


class TrainWorkLayout(C.Structure):     # tvr_train_work_layout
    _fields_ = [(n, C.c_size_t) for n in ("h", "feats32", "h1", "h2", "rgb", "grgb", "d_out4", "dh2", "dh1", "dfeats32", "dh", "X", "total")] + [("x_blocks", C.c_int32), ("x_block_cols", C.c_int32)]


class VmGrads(C.Structure):
    _fields_ = [("density_plane", C.c_void_p * 3), ("density_line", C.c_void_p * 3), ("app_plane", C.c_void_p * 3), ("app_line", C.c_void_p * 3)]


class TrainWeights(C.Structure):         # tvr_train_weights
    _fields_ = [("W1", C.c_void_p), ("W2", C.c_void_p), ("W3", C.c_void_p), ("basis", C.c_void_p), ("heads_W", C.c_void_p * 4)]


class TrainMlpGrads(C.Structure):        # tvr_train_mlp_grads
    _fields_ = [("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("W3", C.c_void_p), ("b3", C.c_void_p), ("basis", C.c_void_p),
                ("heads_W", C.c_void_p * 4), ("heads_b", C.c_void_p * 4)]


class DenseOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("z", "valid", "bbox_valid", "cell", "sigma_feature", "sigma", "alpha",
                                          "weight", "rgb", "bg_weight", "acc", "t_min")]


class MlpnetDesc(C.Structure):           # tvr_mlpnet_desc
    _fields_ = [(n, C.c_int32) for n in ("D", "W", "skip", "pos_freqs", "view_freqs", "samples_per_ray", "arith")]


class MlpnetParams(C.Structure):         # tvr_mlpnet_params
    _fields_ = [("base_W", C.c_void_p * 4), ("base_b", C.c_void_p * 4), ("sigma_W", C.c_void_p), ("sigma_b", C.c_void_p), ("rgbh_W_base", C.c_void_p),
                ("rgbh_W_view", C.c_void_p), ("rgbh_b", C.c_void_p), ("rgbo_W", C.c_void_p), ("rgbo_b", C.c_void_p)]


class MlpnetSaved(C.Structure):          # tvr_mlpnet_saved
    _fields_ = [("act", C.c_void_p * 4), ("act_bytes", C.c_size_t), ("rgb_hidden", C.c_void_p), ("rgb_hidden_bytes", C.c_size_t),
                ("sigma_pre", C.c_void_p), ("sigma_pre_bytes", C.c_size_t), ("embed_pos", C.c_void_p), ("embed_pos_bytes", C.c_size_t),
                ("embed_view", C.c_void_p), ("embed_view_bytes", C.c_size_t), ("act_mask", C.c_void_p * 4), ("rgb_hidden_mask", C.c_void_p),
                ("mask_bytes", C.c_size_t)]


class MlpnetLayout(C.Structure):         # tvr_mlpnet_layout
    _fields_ = [(n, C.c_size_t) for n in ("fragments", "biases", "block_table", "block_table_bytes", "total")]


class NgpMarchCfg(C.Structure):          # tvr_ngp_march_cfg (include/tvr_ngp.h)
    _fields_ = [("aabb_lo", C.c_float * 3), ("aabb_hi", C.c_float * 3), ("near_distance", C.c_float), ("cone_angle", C.c_float),
                ("const_dt", C.c_int32), ("rng_state", C.c_uint64), ("rng_inc", C.c_uint64), ("slab_rays", C.c_uint32)]


class NgpGridCfg(C.Structure):           # tvr_ngp_grid_cfg
    _fields_ = [("offsets", C.c_uint32 * 17), ("scale", C.c_float * 16)]


class NgpNetParams(C.Structure):         # tvr_ngp_net_params
    _fields_ = [(n, C.c_void_p) for n in ("density0", "density1", "rgb0", "rgb1", "rgb2")]


# name -> (restype, argtypes); every symbol include/tvr.h and include/tvr_ngp.h declare
SYMBOLS = {
    "tvr_version": (C.c_int, []),
    "tvr_last_error": (C.c_char_p, []),
    "tvr_scene_packed_bytes": (C.c_size_t, [C.POINTER(SceneDesc)]),
    "tvr_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tvr_scene_update": (C.c_int, [C.c_void_p, C.POINTER(SceneParams), C.c_void_p]),
    "tvr_alpha_bits_bytes": (C.c_size_t, [C.POINTER(C.c_int32 * 3)]),
    "tvr_scene_set_alpha": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_float * 6),
                                      C.POINTER(C.c_float * 3), C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_scene_set_range_check": (C.c_int, [C.c_void_p, C.c_int32]),
    "tvr_scene_set_arith": (C.c_int, [C.c_void_p, C.c_int32]),
    "tvr_scene_set_render_pieces": (C.c_int, [C.c_void_p, C.c_int32]),
    "tvr_scene_get_render_pieces": (C.c_int, [C.c_void_p]),
    "tvr_scene_touch": (C.c_int, [C.c_void_p]),
    "tvr_scene_get_arith": (C.c_int, [C.c_void_p]),
    "tvr_scene_get_arith_requested": (C.c_int, [C.c_void_p]),
    "tvr_scene_validate_arith": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_void_p]),
    "tvr_scene_destroy": (C.c_int, [C.c_void_p]),
    "tvr_render_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "tvr_render_scratch_bytes_min": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "tvr_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_float,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(DenseOut), C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "tvr_render_z": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(DenseOut), C.c_void_p, C.c_void_p]),
    "tvr_march_forward_z": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "tvr_march_backward_z": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(VmGrads), C.c_void_p]),
    "tvr_density_feature": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_app_feature": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_mlp_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_app_feature_ref": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_mlp_render_ref": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_alpha_sample": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_float * 6), C.POINTER(C.c_float * 3),
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_scratch_describe": (C.c_int, [C.c_int64, C.c_int32, C.POINTER(ScratchLayout)]),
    "tvr_march_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "tvr_grad_scratch_bytes": (C.c_size_t, [C.c_void_p]),
    "tvr_march_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(VmGrads), C.c_void_p]),
    "tvr_app_h_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_app_h_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(VmGrads), C.c_void_p]),
    "tvr_mlp_train_image_bytes": (C.c_size_t, []),
    "tvr_mlp_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p, C.c_size_t] * 4 + [C.c_void_p]),
    "tvr_mlp_train_backward": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_void_p] + [C.c_void_p, C.c_size_t] * 5 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_mlp_train_forward_ref": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p, C.c_size_t] * 6 + [C.c_void_p]),
    "tvr_mlp_train_backward_ref": (C.c_int, [C.c_void_p] * 4 + [C.POINTER(C.c_void_p * 4)] + [C.c_void_p] * 8 + [C.c_int64, C.c_void_p] + [C.c_void_p, C.c_size_t] * 6 +
                                   [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_train_work_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64]),
    "tvr_train_work_describe": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.POINTER(TrainWorkLayout)]),
    "tvr_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvr_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int64,
                                     C.POINTER(TrainWeights), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.POINTER(VmGrads), C.POINTER(TrainMlpGrads),
                                     C.c_void_p, C.c_void_p]),
    "tvr_gemm_tn_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int64]),
    "tvr_gemm_tn": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                              C.c_void_p]),
    "tvr_gemm_tn_bias": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "tvr_pe_concat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_pe_concat_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvr_tv_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_filter_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_l1_mean_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "tvr_l1_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_l1_mean_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "tvr_line_ortho": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_line_ortho_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "tvr_profile_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "tvr_profile_reset": (C.c_int, [C.c_void_p]),
    "tvr_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 3)]),
    "tvr_profile_destroy": (C.c_int, [C.c_void_p]),
    "tvr_mlpnet_packed_bytes": (C.c_size_t, [C.POINTER(MlpnetDesc)]),
    "tvr_mlpnet_pack": (C.c_int, [C.POINTER(MlpnetDesc), C.POINTER(MlpnetParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_mlpnet_work_bytes": (C.c_size_t, []),
    "tvr_mlpnet_describe": (C.c_int, [C.POINTER(MlpnetDesc), C.POINTER(MlpnetLayout)]),
    "tvr_mlpnet_forward": (C.c_int, [C.POINTER(MlpnetDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "tvr_mlpnet_train_forward": (C.c_int, [C.POINTER(MlpnetDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.POINTER(MlpnetSaved), C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_mlpnet_repack": (C.c_int, [C.POINTER(MlpnetDesc), C.POINTER(MlpnetParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_linear_dx": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_int32, C.c_size_t, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvr_gemm_tn_scaled": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "tvr_colsum_scratch_bytes": (C.c_size_t, []),
    "tvr_colsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_npp_bg_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvr_npp_bg_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    # include/tvr_ngp.h
    "tvr_ngp_update_bitfield": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_ngp_sample_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "tvr_ngp_sample": (C.c_int, [C.POINTER(NgpMarchCfg), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_ngp_hash_encode": (C.c_int, [C.POINTER(NgpGridCfg), C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "tvr_ngp_sh_encode": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "tvr_ngp_net_packed_bytes": (C.c_size_t, []),
    "tvr_ngp_net_pack": (C.c_int, [C.POINTER(NgpNetParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_ngp_network": (C.c_int, [C.POINTER(NgpGridCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "tvr_ngp_render_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "tvr_ngp_render": (C.c_int, [C.POINTER(NgpMarchCfg), C.POINTER(NgpGridCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.POINTER(C.c_float * 3), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tvr_ngp_render_profiled": (C.c_int, [C.POINTER(NgpMarchCfg), C.POINTER(NgpGridCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.POINTER(C.c_float * 3), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_float * 2)]),
    "tvr_ngp_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float * 3), C.c_void_p, C.c_void_p]),
}

ARITH_MIN_PROBE_SAMPLES = 2048          # include/tvr.h TVR_ARITH_MIN_PROBE_SAMPLES
STAT_SAMPLES_EVAL, STAT_SAMPLES_BBOX, STAT_APP, STAT_RAYS_TERMINATED, STAT_COUNT = 0, 1, 2, 3, 8


class TvrError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libtvr.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TvrError(f"HIP extension not built: {LIB_PATH} is missing (run `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` or `make -C jittor-myc-nerfs_amd/csrc`). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)      # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


# ---- caller-owned device buffers (include/tvr.h: "ALL device memory is caller-owned ... sizes come from the *_bytes() queries") ----------------------------------
# Every scratch / work / packed / output buffer this host hands to the library is allocated here.  Normally that is a plain torch allocation of exactly the size the
# library's query returned.  With GUARD_BYTES > 0 (tests/test_gpu_canaries.py) each allocation carries that many 0xA5 bytes BEHIND its last byte, and check_guards()
# tells whether any launch wrote past the size it was given — the class of bug behind round 5's abort (a kernel-owned word placed where another
# kernel's table lived) and round 2's (an output matrix narrower than the kernel's rows).
GUARD_BYTES = 0
_guarded = []          # (base tensor, payload bytes, what)


def dev_bytes(n: int, device, zero: bool = False, what: str = ""):
    """uint8 device tensor of exactly `n` bytes, 256-byte aligned; guarded when GUARD_BYTES is set."""
    import torch
    n = int(n)
    if GUARD_BYTES <= 0:
        return (torch.zeros if zero else torch.empty)(n, dtype=torch.uint8, device=device)
    base = torch.full((n + GUARD_BYTES,), 0xA5, dtype=torch.uint8, device=device)
    if zero:
        base[:n].zero_()
    _guarded.append((base, n, what))
    return base[:n]


def dev_empty(shape, dtype, device, what: str = ""):
    """torch.empty(shape, dtype) on the device through dev_bytes (outputs whose byte count the library checks)."""
    import torch
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    if GUARD_BYTES <= 0:
        return torch.empty(shape, dtype=dtype, device=device)
    numel = 1
    for x in shape:
        numel *= x
    return dev_bytes(numel * torch.empty((), dtype=dtype).element_size(), device, what=what).view(dtype).view(shape)


def check_guards(clear: bool = False):
    """[(what, payload bytes, first damaged offset behind the payload)] for every guarded buffer whose guard bytes are no longer 0xA5."""
    bad = []
    for base, n, what in _guarded:
        g = base[n:]
        hit = (g != 0xA5).nonzero()
        if hit.numel():
            bad.append((what, n, int(hit[0])))
    if clear:
        _guarded.clear()
    return bad


def nbytes(t) -> int:
    """Size in bytes of a tensor's storage from its first element on: what the `*_bytes` arguments of include/tvr.h take."""
    return t.numel() * t.element_size()


def check(rc: int, what: str = "") -> None:
    if rc < 0:
        raise TvrError(f"{what} failed ({rc}): {lib().tvr_last_error().decode(errors='replace')}")
