"""ctypes binding of ``libipoke_hip.so`` (the C ABI declared in ``include/ipoke_hip.h``).

The product path has no CPU fallback: if the shared library is missing or a call
fails, a ``RuntimeError`` is raised.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C ipoke_amd/csrc``.
"""
import ctypes
import os
from ctypes import CFUNCTYPE, POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPOKE_LIB_PATH") or os.path.join(_HERE, "libipoke_hip.so")     # IPOKE_LIB_PATH: developer A/B of two builds in one GPU call

F32, BF16 = 0, 1
ACT_NONE, ACT_ELU, ACT_RELU, ACT_LRELU02, ACT_TANH, ACT_SIGMOID = range(6)

DTYPES = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16}


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("NB", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "kd", "kh", "kw", "sd", "sh", "sw",
                                       "pd", "ph", "pw", "transposed")] + [
        ("A", c_void_p), ("a_f32", c_int32),
        ("a_sn", c_int64), ("a_sd", c_int64), ("a_sh", c_int64), ("a_sw", c_int64), ("a_sc", c_int64),
        ("a_coff", c_int32), ("Kc_real", c_int32), ("Kc", c_int32),
        ("W", c_void_p), ("ldw", c_int32), ("Nout", c_int32),
        ("bias", c_void_p), ("act", c_int32), ("dact", c_void_p), ("ld_dact", c_int32), ("dact_act", c_int32),
        ("C", c_void_p), ("c_f32", c_int32), ("c_accumulate", c_int32), ("ldc", c_int64),
        ("c_coff", c_int32), ("c_cstride", c_int32), ("splitk", c_int32),
        ("c_scatter", c_int32), ("c_sn", c_int64), ("c_sh", c_int64), ("c_sw", c_int64), ("c_row0", c_int64),
        ("row_scale", c_void_p), ("rs_images", c_int32), ("rs_stride", c_int32), ("w_kmajor", c_int32), ("c_sd", c_int64),
        ("acc_scratch", c_void_p), ("acc_scratch_bytes", c_int64)]


class WgradDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("NB", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "kd", "kh", "kw", "sd", "sh", "sw",
                                       "pd", "ph", "pw", "transposed")] + [
        ("A", c_void_p), ("a_f32", c_int32),
        ("a_sn", c_int64), ("a_sd", c_int64), ("a_sh", c_int64), ("a_sw", c_int64), ("a_sc", c_int64),
        ("a_coff", c_int32), ("Kc_real", c_int32), ("Kc", c_int32),
        ("dY", c_void_p), ("ldy", c_int32), ("y_coff", c_int32), ("Nout", c_int32),
        ("dW", c_void_p), ("w_sn", c_int64), ("w_sc", c_int64), ("w_st", c_int64),
        ("accumulate", c_int32), ("splitm", c_int32), ("Kc_store", c_int32), ("split_stride", c_int64), ("max_workgroups", c_int32),
        ("adam", c_void_p)]


class WgradAdam(Structure):
    _fields_ = [("params", c_void_p), ("m", c_void_p), ("v", c_void_p), ("vmax", c_void_p), ("operand", c_void_p),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float), ("grad_scale", c_float),
                ("step", c_int32), ("keep_grad", c_int32)]


class AffineDesc(Structure):
    _fields_ = [("raw", c_void_p), ("nsplit", c_int32), ("split_stride", c_int64), ("ldraw", c_int32),
                ("bias", c_void_p), ("Cp", c_int32), ("t_off", c_int32), ("t_stride", c_int32),
                ("P", c_int32), ("ld", c_int32)]


class CouplingEpi(Structure):
    """ipoke_coupling_epi (include/ipoke_hip.h): outputs of the fused conv3 + coupling launch"""
    _fields_ = [("mode", c_int32), ("inp", c_void_p), ("out", c_void_p), ("out2", c_void_p), ("scale_out", c_void_p),
                ("logdet_slot", c_void_p), ("slot_stride", c_int32), ("an_c0", c_int32), ("an_C", c_int32),
                ("an_log_scale", c_void_p), ("an_bias", c_void_p), ("an_idx", c_void_p), ("ext", c_void_p), ("ext_ld", c_int32),
                ("xchg", c_void_p)]


class NormBwdDesc(Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_int32), ("y", c_void_p), ("ldy", c_int32), ("dy", c_void_p), ("lddy", c_int32),
                ("dx", c_void_p), ("lddx", c_int32), ("dres", c_void_p), ("lddres", c_int32),
                ("dmod_gamma", c_void_p), ("dmod_beta", c_void_p), ("ld_dmod", c_int32),
                ("N", c_int32), ("S", c_int32), ("C", c_int32), ("G", c_int32), ("eps", c_float),
                ("gamma", c_void_p), ("beta", c_void_p), ("mod_gamma", c_void_p), ("ld_mod", c_int32),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("act", c_int32), ("workspace", c_void_p), ("stats", c_void_p),
                ("mod_samples", c_int32),
                ("rs_scale", c_void_p), ("rs_scale_stride", c_int32), ("rs_rows_per_group", c_int64), ("rs_bias", c_void_p),
                ("rs_dots", c_void_p), ("rs_dbias", c_void_p), ("rs_workspace", c_void_p), ("dmod_summed", c_int32),
                ("act_from_pre", c_int32)]


class GruDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "T", "L", "Cx", "Ch", "H", "W")]


class RowScaleBwdDesc(Structure):
    _fields_ = [("dy", c_void_p), ("lddy", c_int32), ("y", c_void_p), ("ldy", c_int32), ("M", c_int64), ("C", c_int32), ("Cpad", c_int32),
                ("act", c_int32), ("bias", c_void_p), ("scale", c_void_p), ("scale_stride", c_int32), ("rows_per_group", c_int64),
                ("gs", c_void_p), ("ldgs", c_int32), ("dots", c_void_p), ("dbias", c_void_p), ("workspace", c_void_p)]


class McfDesc(Structure):
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("ld", c_int32), ("C", c_int32), ("B", c_int32),
                ("cond", c_void_p), ("Cc", c_int32),
                ("W1", c_void_p), ("W2", c_void_p), ("bias2", c_void_p), ("order", c_int32),
                ("rows_per_block", c_int32),
                ("a2_save", c_void_p), ("scale_save", c_void_p), ("logdet_slot", c_void_p),
                ("W2T", c_void_p), ("W1T", c_void_p), ("dy", c_void_p), ("dld", c_void_p), ("dx", c_void_p),
                ("dparams_save", c_void_p), ("dc_save", c_void_p), ("dbias_part", c_void_p),
                ("post_log_scale", c_void_p), ("post_bias", c_void_p), ("y_post", c_void_p), ("post_part", c_void_p),
                ("x_op_save", c_void_p),
                ("zc_out", c_void_p), ("zc_off", c_int32), ("zc_stride", c_int32), ("zc_cin", c_int32), ("zc_ld", c_int32),
                ("split", c_int32), ("xchg", c_void_p), ("pair", c_void_p)]


class UnitPairDesc(Structure):
    """ipoke_unit_pair_desc (include/ipoke_hip.h)"""
    _fields_ = [("an_log_scale", c_void_p), ("an_idx", c_void_p), ("an_x", c_void_p), ("an_part", c_void_p),
                ("Cp", c_int32), ("t_off", c_int32), ("t_stride", c_int32), ("x0", c_void_p), ("scale", c_void_p),
                ("dparams", c_void_p), ("ldp", c_int32), ("dbias_part", c_void_p), ("dx", c_void_p)]


class NormDesc(Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_int32), ("y", c_void_p), ("ldy", c_int32), ("y_f32", c_int32),
                ("N", c_int32), ("S", c_int32), ("C", c_int32), ("G", c_int32), ("eps", c_float),
                ("gamma", c_void_p), ("beta", c_void_p),
                ("mod_gamma", c_void_p), ("mod_beta", c_void_p), ("ld_mod", c_int32),
                ("res", c_void_p), ("ld_res", c_int32), ("act", c_int32), ("workspace", c_void_p), ("mod_samples", c_int32),
                ("res_post", c_int32), ("next_part", c_void_p), ("next_G", c_int32), ("part_chunks", c_int32)]


class SnJob(Structure):
    _fields_ = [("w", c_void_p), ("cout", c_int32), ("cin", c_int32), ("taps", c_int32), ("transposed", c_int32),
                ("u", c_void_p), ("v", c_void_p), ("out", c_void_p), ("out_stride", c_int64), ("snap", c_void_p), ("snap_stride", c_int64),
                ("workspace", c_void_p)]


class FlowConfig(Structure):
    _fields_ = [("z_channels", c_int32), ("hidden", c_int32), ("cond_channels", c_int32), ("factor", c_int32),
                ("n_levels", c_int32), ("num_steps", c_int32 * 32), ("kernel_h", c_int32), ("kernel_w", c_int32),
                ("dtype", c_int32), ("max_batch", c_int32), ("use1x1", c_int32), ("condition_nice", c_int32)]


# name -> (restype, argtypes); every symbol declared in include/ipoke_hip.h
_P = c_void_p
GRAD_READY_FN = CFUNCTYPE(None, c_void_p, c_int, c_int64, c_int64)     # ipoke_grad_ready_fn(user, piece, begin, end)
SIGNATURES = {
    "ipoke_last_error": (c_char_p, []),
    "ipoke_version": (c_int, []),
    "ipoke_dtype_size": (c_int, [c_int]),
    "ipoke_conv_forward": (c_int, [POINTER(ConvDesc), c_int, _P]),
    "ipoke_conv_forward_repeat": (c_int, [POINTER(ConvDesc), c_int, c_int, _P]),
    "ipoke_set_dispatch_override": (c_int, [c_char_p, c_int]),
    "ipoke_last_conv_kernel": (c_int, []),
    "ipoke_gru_set_fused": (c_int, [c_int]),
    "ipoke_gru_workspace_bytes": (c_int64, [POINTER(GruDesc), c_int]),
    "ipoke_gru_unroll_forward": (c_int, [POINTER(GruDesc), _P, c_int, _P, c_int, POINTER(c_void_p), _P, _P, c_int, c_int, _P]),
    "ipoke_gru_unroll_backward": (c_int, [POINTER(GruDesc), _P, c_int, _P, POINTER(c_void_p), _P, _P, c_int, _P]),
    "ipoke_rowscale_bwd_workspace_floats": (c_int64, [c_int64, c_int, c_int64]),
    "ipoke_rowscale_bwd": (c_int, [POINTER(RowScaleBwdDesc), c_int, _P]),
    "ipoke_spectral_bwd_frames": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int64, _P, c_int64, _P, c_int, _P]),
    "ipoke_sum_frames": (c_int, [_P, _P, c_int, c_int64, c_int, _P]),
    "ipoke_desc_sizes": (c_int, [POINTER(c_int32), c_int]),
    "ipoke_conv3x3_skinny_splitk": (c_int, [c_int, c_int, c_int]),
    "ipoke_conv3x3_coupling_splitk": (c_int, [c_int, c_int, c_int]),
    "ipoke_conv3x3_coupling_xchg_bytes": (c_int64, []),
    "ipoke_conv3x3_coupling_xchg_init": (c_int, [_P, _P]),
    "ipoke_conv3x3_coupling": (c_int, [POINTER(ConvDesc), POINTER(AffineDesc), POINTER(CouplingEpi), c_int, c_int, _P]),
    "ipoke_conv_wgrad": (c_int, [POINTER(WgradDesc), c_int, _P]),
    "ipoke_wgrad_batch_entry_size": (c_int, []),
    "ipoke_conv_wgrad_batched": (c_int, [POINTER(WgradDesc), _P, c_int, _P, _P, _P, c_int, _P]),
    "ipoke_reduce_entry_size": (c_int, []),
    "ipoke_reduce_rows_multi": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "ipoke_nchw_to_state": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ipoke_state_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ipoke_extract_cols": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int64, c_int, _P]),
    "ipoke_copy_cols": (c_int, [_P, c_int, _P, c_int, c_int, c_int64, c_int, _P]),
    "ipoke_cond_prepare": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ipoke_actnorm_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "ipoke_actnorm_inv": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "ipoke_actnorm_inv_ext": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ipoke_actnorm_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P]),
    "ipoke_actnorm_init": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "ipoke_affine_fwd": (c_int, [POINTER(AffineDesc), _P, _P, _P, _P, c_int, c_int, _P]),
    "ipoke_affine_inv": (c_int, [POINTER(AffineDesc), _P, _P, c_int, _P]),
    "ipoke_affine_fwd_ext": (c_int, [POINTER(AffineDesc), _P, _P, _P, _P, c_int, c_int, _P, c_int, c_int, _P]),
    "ipoke_affine_inv_ext": (c_int, [POINTER(AffineDesc), _P, _P, c_int, _P, c_int, c_int, _P]),
    "ipoke_affine_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, _P]),
    "ipoke_affine_actnorm_fwd": (c_int, [POINTER(AffineDesc), _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "ipoke_actnorm_affine_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P,
                                         c_int, c_int, _P]),
    "ipoke_reduce_rows": (c_int, [_P, _P, c_int, c_int, _P]),
    "ipoke_spin_delay": (c_int, [c_int, _P]),
    "ipoke_logdet_finalize": (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "ipoke_actnorm_logdet": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "ipoke_flow_nll": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "ipoke_adam_amsgrad_step": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int,
                                        c_float, _P]),
    "ipoke_adam_amsgrad_step_grid": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int,
                                             c_float, c_int, _P]),
    "ipoke_mcf_shadow_dims": (c_int, [c_int, c_int, c_int, POINTER(c_int32)]),
    "ipoke_mcf_fwd": (c_int, [POINTER(McfDesc), c_int, _P]),
    "ipoke_mcf_inv": (c_int, [POINTER(McfDesc), c_int, _P]),
    "ipoke_mcf_bwd": (c_int, [POINTER(McfDesc), c_int, _P]),
    "ipoke_macow_unit_supported": (c_int, [c_int, c_int, c_int]),
    "ipoke_macow_unit_xchg_bytes": (c_int64, [c_int, c_int]),
    "ipoke_macow_unit_fwd": (c_int, [POINTER(McfDesc), c_int, _P]),
    "ipoke_macow_unit_bwd": (c_int, [POINTER(McfDesc), c_int, _P]),
    "ipoke_macow_unit_inv": (c_int, [POINTER(McfDesc), c_int, _P]),
    "ipoke_relayout_job_size": (c_int, []),
    "ipoke_wn_job_size": (c_int, []),
    "ipoke_relayout_multi": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, c_int, _P]),
    "ipoke_wn_scale_multi": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "ipoke_wn_bwd_multi": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "ipoke_groupnorm_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "ipoke_groupnorm": (c_int, [POINTER(NormDesc), c_int, _P]),
    "ipoke_add_act": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int64, c_int, c_int, c_int, _P]),
    "ipoke_gru_gates": (c_int, [_P, _P, c_int, _P, c_int, _P, c_int64, c_int, c_int, _P]),
    "ipoke_gru_update": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int64, c_int, c_int, _P]),
    "ipoke_reparameterize": (c_int, [_P, c_int, _P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "ipoke_bilinear_cl": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "ipoke_cl_to_nchw": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "ipoke_nchw_to_cl": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ipoke_image_metrics_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "ipoke_psnr_ssim": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "ipoke_clip_to_cl4": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "ipoke_groupnorm_stats": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int, _P]),
    "ipoke_groupnorm_stats_offset": (c_int64, [c_int, c_int, c_int]),
    "ipoke_groupnorm_bwd_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "ipoke_groupnorm_bwd": (c_int, [POINTER(NormBwdDesc), c_int, _P]),
    "ipoke_groupnorm_bwd_rs_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "ipoke_act_bwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int64, c_int, c_int, c_int, c_int, _P]),
    "ipoke_colsum_workspace_floats": (c_int64, [c_int64, c_int]),
    "ipoke_colsum": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, c_int, _P, c_int, _P]),
    "ipoke_gru_update_bwd": (c_int, [_P, _P, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int64, c_int, c_int, _P]),
    "ipoke_gru_gates_bwd": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int64, c_int, c_int, _P]),
    "ipoke_conv_weight_operand": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P]),
    "ipoke_conv_weight_operand_multi": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), c_int, c_int, _P]),
    "ipoke_spectral_workspace_floats": (ctypes.c_long, [c_int, c_int, c_int]),
    "ipoke_spectral_sigma": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, ctypes.c_float, _P, _P, _P, _P]),
    "ipoke_spectral_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "ipoke_spectral_bwd_workspace_floats": (ctypes.c_long, []),
    "ipoke_sn_job_size": (c_int, []),
    "ipoke_sn_jobs_upload": (c_int, [_P, c_int, _P, _P]),
    "ipoke_spectral_sigma_multi": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, _P]),
    "ipoke_adam_multi": (c_int, [_P, _P, _P, _P, _P, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, c_int, ctypes.c_float, _P]),
    "ipoke_maxpool3d_fwd": (c_int, [_P, _P, c_int, _P, c_int, _P, c_int, _P]),
    "ipoke_maxpool3d_bwd": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, _P]),
    "ipoke_avgpool_rows": (c_int, [_P, c_int, _P, c_int, c_int64, c_int, c_int, c_int, _P]),
    "ipoke_avgpool_rows_bwd": (c_int, [_P, c_int, _P, c_int, c_int64, c_int, c_int, c_int, _P]),
    "ipoke_groupnorm_jvp": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float,
                                    _P, c_int, _P]),
    "ipoke_groupnorm_jvp_bwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P,
                                        c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int, _P]),
    "ipoke_groupnorm_jvp_workspace_floats": (ctypes.c_long, [c_int, c_int]),
    "ipoke_gather_rows": (c_int, [_P, c_int, _P, _P, c_int, c_int64, c_int, c_int, _P]),
    "ipoke_l1_pair": (c_int, [_P, c_int, _P, c_int, c_int64, c_int, c_float, _P, _P, c_int, c_int, _P]),
    "ipoke_kl_loss": (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, _P]),
    "ipoke_reparam_bwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, _P]),
    "ipoke_l1_loss": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int64, c_float, _P, _P, c_int, _P, _P]),
    "ipoke_l1_loss_partials": (ctypes.c_long, []),
    "ipoke_relayout_multi_range": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "ipoke_wn_scale_multi_range": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ipoke_flow_prepare_weights_range": (c_int, [_P, _P, _P, c_int64, c_int64, _P]),
    "ipoke_flow_set_native_adam": (c_int, [_P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_int]),
    "ipoke_flow_adam_range": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float, c_int,
                                      c_float, c_int, _P]),
    "ipoke_adam_tile_job_size": (c_int, []),
    "ipoke_adam_seg_size": (c_int, []),
    "ipoke_adam_amsgrad_shadow_tiles": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                                                c_int, c_float, c_int, c_int, _P]),
    "ipoke_adam_amsgrad_cast_tiles": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                                              c_int, c_float, c_int, c_int, _P]),
    "ipoke_adam_amsgrad_segments": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int64, c_int64, c_float, c_float, c_float, c_float,
                                            c_float, c_int, c_float, c_int, _P]),
    "ipoke_wn_bwd_multi_range": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ipoke_flow_backward_pieces": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, _P, GRAD_READY_FN, _P, _P]),
    "ipoke_video_to_cl": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int,
                                  c_int, _P, _P]),
    "ipoke_min_reset": (c_int, [_P, _P]),
    "ipoke_denorm_if_negative": (c_int, [_P, c_int64, c_int, c_int, c_int, c_int, _P, _P]),
    "ipoke_pool3d_same": (c_int, [_P, _P, c_int, _P, c_int, c_int, _P]),
    "ipoke_pool_rows_weighted": (c_int, [_P, c_int, _P, c_int, c_int64, c_int, c_int, _P, c_int, _P]),
    "ipoke_activation_moments": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "ipoke_flow_resize": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "ipoke_poke_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "ipoke_poke_simulate": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ipoke_timing_start": (c_int, []),
    "ipoke_timing_start_all": (c_int, []),
    "ipoke_timing_stop": (c_int, [POINTER(c_int), c_int, POINTER(c_int), POINTER(ctypes.c_double)]),
    "ipoke_timing_stop_ex": (c_int, [POINTER(c_int), c_int, POINTER(c_int), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                     POINTER(ctypes.c_double)]),
    "ipoke_flow_create": (c_int, [POINTER(FlowConfig), POINTER(c_void_p)]),
    "ipoke_flow_destroy": (None, [_P]),
    "ipoke_flow_piece_ranges": (c_int, [_P, c_int, POINTER(c_int64), c_int]),
    "ipoke_flow_handoff_timeouts": (c_int, [_P, POINTER(c_uint32)]),
    "ipoke_flow_side_stream": (c_void_p, [_P]),
    "ipoke_flow_test_inject_timeout": (c_int, [_P, c_int, _P]),
    "ipoke_conv_wgrad_splitm": (c_int, [_P, c_int, c_int]),
    "ipoke_conv_acc_scratch_bytes": (c_int64, [c_int, c_int, c_int]),
    "ipoke_conv_acc_scratch_init": (c_int, [_P, _P]),
    "ipoke_flow_param_count": (c_int64, [_P]),
    "ipoke_flow_index_count": (c_int64, [_P]),
    "ipoke_flow_tensor_count": (c_int32, [_P]),
    "ipoke_flow_op_count": (c_int32, [_P]),
    "ipoke_flow_tensor_info": (c_int, [_P, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int32), POINTER(c_int64),
                                       POINTER(c_int32)]),
    "ipoke_flow_shadow_bytes": (c_int64, [_P]),
    "ipoke_flow_set_graph": (c_int, [_P, c_int]),
    "ipoke_flow_float_buffer_count": (c_int64, [_P]),
    "ipoke_flow_set_float_buffers": (c_int, [_P, _P]),
    "ipoke_lu_job_size": (c_int, []),
    "ipoke_lu_prepare": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "ipoke_lu_apply": (c_int, [_P, _P, c_int64, c_int, c_int, _P, c_int, _P]),
    "ipoke_lu_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "ipoke_flow_op_info": (c_int, [_P, c_int, POINTER(c_int64)]),
    "ipoke_flow_shadow_base": (c_int64, [_P]),
    "ipoke_flow_workspace_bytes": (c_int64, [_P, c_int, c_int]),
    "ipoke_flow_prepare_weights": (c_int, [_P, _P, _P, _P]),
    "ipoke_flow_forward": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, _P]),
    "ipoke_flow_init_forward": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "ipoke_flow_reverse": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "ipoke_flow_backward": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. ipoke_amd has no CPU fallback; "
                "run `make -C ipoke_amd/csrc` (or __graft_entry__.build()) first.")
        # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  Load it FIRST so that this library binds to the same
        # HIP runtime instance (same SONAME): with the order reversed the process holds two runtimes and the second one
        # finds no device ("no ROCm-capable device is detected" from the first kernel-attribute call).
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().ipoke_last_error()
        raise RuntimeError(f"libipoke_hip call failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device (or host) pointer of a torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("ipoke_amd needs an MI355X (gfx950) GPU: no HIP device is visible and there is no CPU path")


import contextlib


KERNEL_NONE, KERNEL_IGEMM, KERNEL_S8, KERNEL_HALO, KERNEL_HALO16, KERNEL_C64, KERNEL_K8 = range(7)       # ipoke_last_conv_kernel


@contextlib.contextmanager
def dispatch_override(name, value):
    """Test hook: run a block with the kernel-dispatch switch ``name`` ("c64" | "halo16") at ``value`` (0 off, 1 the measured default
    rule, 2 wherever the kernel can run), then return it to the process default (ipoke_set_dispatch_override)."""
    check(lib().ipoke_set_dispatch_override(name.encode(), int(value)))
    try:
        yield
    finally:
        check(lib().ipoke_set_dispatch_override(name.encode(), -1))
