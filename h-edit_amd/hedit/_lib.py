"""ctypes binding of libhedit_hip.so (include/hedit.h).  Fails loudly if the library is absent."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEDIT_STORAGE=f16 selects the half-storage build of the same kernels (`python h-edit_amd/build.py --f16`, csrc/common.h): same
# MFMA rate, the eps error of the SD UNet against fp32 1.7e-3 instead of bfloat16's 1.2e-2.  One storage format per process,
# decided before the library is first used.  Default and every benchmark figure: bfloat16 (BASELINE configs[1]).
STORAGE = os.environ.get("HEDIT_STORAGE", "bf16").lower()
if STORAGE not in ("bf16", "f16"):
    raise ValueError(f"HEDIT_STORAGE must be bf16 or f16, not {STORAGE!r}")
LIB_PATH = os.path.join(_HERE, "libhedit_hip_f16.so" if STORAGE == "f16" else "libhedit_hip.so")
HEDIT_MAX_LEVELS = 8


class UnetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("sample_size", C.c_int),
                ("n_levels", C.c_int),
                ("block_out_channels", C.c_int * HEDIT_MAX_LEVELS),
                ("down_has_attn", C.c_int * HEDIT_MAX_LEVELS),
                ("up_has_attn", C.c_int * HEDIT_MAX_LEVELS),
                ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int),
                ("heads", C.c_int), ("norm_num_groups", C.c_int)]


class VaeCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("latent_channels", C.c_int), ("n_levels", C.c_int),
                ("block_out_channels", C.c_int * 4), ("layers_per_block", C.c_int),
                ("norm_num_groups", C.c_int)]


class DdpmCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_ch", C.c_int), ("ch", C.c_int), ("n_levels", C.c_int),
                ("ch_mult", C.c_int * 8), ("attn_level", C.c_int * 8), ("num_res_blocks", C.c_int),
                ("image_size", C.c_int)]


class VitCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "layers", "heads", "patch_size", "input_resolution")]


class P2PPlan(C.Structure):
    _fields_ = [("mode", C.c_int), ("n_pairs", C.c_int),
                ("pair_src", C.c_void_p), ("pair_tar", C.c_void_p),
                ("singles", C.c_void_p), ("n_single", C.c_int),
                ("qk_src", C.c_void_p), ("mixT", C.c_void_p), ("bvec", C.c_void_p),
                ("h_store", C.POINTER(C.c_void_p)), ("n_store", C.c_int),
                ("kv_src", C.c_void_p), ("kv_first_block", C.c_int),
                ("qk_first_block", C.c_int), ("qk_max_tokens", C.c_int), ("feat_src", C.c_void_p), ("feat_resblock", C.c_int),
                ("h_store_self", C.POINTER(C.c_void_p)), ("n_store_self", C.c_int)]


class StepCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("sqrt_ab_t", "sqrt_1m_ab_t", "sqrt_ab_prev", "dir_coef",
                                         "noise_coef", "w_src", "w_hat", "w_tar", "coeff", "w_rec")]


_SIGS = {
    "hedit_last_error": (C.c_char_p, []),
    "hedit_version": (C.c_int, []),
    "hedit_unet_create": (C.c_int, [C.POINTER(UnetCfg), C.POINTER(C.c_void_p)]),
    "hedit_unet_destroy": (None, [C.c_void_p]),
    "hedit_unet_num_params": (C.c_int, [C.c_void_p]),
    "hedit_unet_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "hedit_unet_param_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hedit_unet_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_unet_param_bf16_exact": (C.c_int, [C.c_void_p, C.c_int]),
    "hedit_unet_missing": (C.c_int, [C.c_void_p]),
    "hedit_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "hedit_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.POINTER(P2PPlan), C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "hedit_unet_set_attn_hook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hedit_unet_num_store_layers": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hedit_unet_store_layer_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]),
    "hedit_prof_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hedit_prof_reset": (C.c_int, [C.c_void_p]),
    "hedit_prof_collect": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int64)]),
    "hedit_prof_collect_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "hedit_prof_records": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    "hedit_step_base": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(StepCoef), C.c_void_p]),
    "hedit_step_invert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.POINTER(StepCoef), C.c_void_p]),
    "hedit_step_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(StepCoef),
                                    C.c_void_p]),
    "hedit_step_tweedie": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hedit_step_style": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hedit_local_blend": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "hedit_local_blend_sub": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "hedit_ddpm_create": (C.c_int, [C.POINTER(DdpmCfg), C.POINTER(C.c_void_p)]),
    "hedit_ddpm_destroy": (None, [C.c_void_p]),
    "hedit_ddpm_num_params": (C.c_int, [C.c_void_p]),
    "hedit_ddpm_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "hedit_ddpm_param_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hedit_ddpm_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_ddpm_missing": (C.c_int, [C.c_void_p]),
    "hedit_ddpm_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "hedit_ddpm_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "hedit_irse50_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "hedit_irse50_destroy": (None, [C.c_void_p]),
    "hedit_irse50_num_params": (C.c_int, [C.c_void_p]),
    "hedit_irse50_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "hedit_irse50_param_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hedit_irse50_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_irse50_missing": (C.c_int, [C.c_void_p]),
    "hedit_irse50_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hedit_irse50_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "hedit_irse50_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_irse50_cos_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_lpips_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "hedit_lpips_destroy": (None, [C.c_void_p]),
    "hedit_lpips_num_params": (C.c_int, [C.c_void_p]),
    "hedit_lpips_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "hedit_lpips_param_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hedit_lpips_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_lpips_missing": (C.c_int, [C.c_void_p]),
    "hedit_lpips_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hedit_lpips_feature_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "hedit_lpips_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "hedit_lpips_source": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "hedit_lpips_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_vit_create": (C.c_int, [C.POINTER(VitCfg), C.POINTER(C.c_void_p)]),
    "hedit_vit_destroy": (None, [C.c_void_p]),
    "hedit_vit_num_params": (C.c_int, [C.c_void_p]),
    "hedit_vit_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "hedit_vit_param_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hedit_vit_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_vit_missing": (C.c_int, [C.c_void_p]),
    "hedit_vit_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hedit_vit_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "hedit_vit_gram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_vit_gram_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_vae_create": (C.c_int, [C.POINTER(VaeCfg), C.POINTER(C.c_void_p)]),
    "hedit_vae_destroy": (None, [C.c_void_p]),
    "hedit_vae_num_params": (C.c_int, [C.c_void_p]),
    "hedit_vae_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "hedit_vae_param_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hedit_vae_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_vae_missing": (C.c_int, [C.c_void_p]),
    "hedit_vae_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hedit_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p]),
    "hedit_vae_decode_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hedit_vae_decode_keep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "hedit_vae_decode_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hedit_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p]),
    "hedit_axis_mix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
    "hedit_k_gemm_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "hedit_k_gemm_canonical_chunk": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "hedit_k_gemm_plan_splits": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "hedit_k_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 13 +
                     [C.c_void_p, C.c_void_p]),
    "hedit_k_conv_gn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 12 +
                        [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hedit_k_groupnorm_from_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hedit_k_pack_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_gemm_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "hedit_test_set_flags": (C.c_int, [C.c_int]),
    "hedit_storage_is_f16": (C.c_int, []),
    "hedit_k_lin_chain_stream_bytes": (C.c_size_t, [C.c_int]),
    "hedit_k_lin_chain_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "hedit_k_lin_chain": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_lin_chain_sched": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_groupnorm_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "hedit_k_ffn_channels": (C.c_int, []),
    "hedit_k_ffn_stream_bytes": (C.c_size_t, [C.c_int]),
    "hedit_k_ffn_bias_bytes": (C.c_size_t, []),
    "hedit_k_ffn_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hedit_k_ffn_chain": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                    C.c_void_p]),
    "hedit_k_ffn_fused": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_groupnorm_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hedit_k_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "hedit_k_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_float, C.c_void_p]),
    "hedit_k_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "hedit_k_self_attn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hedit_k_cross_attn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(P2PPlan), C.c_void_p,
                                     C.c_void_p]),
    "hedit_k_attn_probs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_attn_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_pack_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hedit_k_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}

EXPORTS = tuple(_SIGS.keys())


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library (loads on first use).  There is deliberately no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python h-edit_amd/build.py{' --f16' if STORAGE == 'f16' else ''}` "
                "(hipcc --offload-arch=gfx950); hedit has no CPU/eager fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)     # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if bool(l.hedit_storage_is_f16()) != (STORAGE == "f16"):
            raise HipLibraryMissing(f"{LIB_PATH} was not built for HEDIT_STORAGE={STORAGE}")
        _lib = l
    return _lib


def storage_dtype():
    """torch dtype of the 16-bit tensors the kernel-level entries (hedit_k_*) and the P2P mix tables exchange with the library"""
    import torch
    return torch.float16 if STORAGE == "f16" else torch.bfloat16


class HipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise HipError(f"libhedit_hip error {rc}: {lib().hedit_last_error().decode()}")


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
