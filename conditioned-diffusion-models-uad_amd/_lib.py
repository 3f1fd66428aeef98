"""ctypes binding of csrc/libcddpm_hip.so (C ABI in include/cddpm.h).

There is no CPU fallback: if the library is missing or cannot be loaded the import of the HIP path
fails loudly with the reason (build it with `python __graft_entry__.py` or `build.build_lib()`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CDDPM_LIB") or os.path.join(_HERE, "csrc", "libcddpm_hip.so")   # CDDPM_LIB: A/B builds (tools/)

CDDPM_MAX_LEVELS = 8


class UnetDesc(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("model_channels", C.c_int32),
        ("num_levels", C.c_int32),
        ("channel_mult", C.c_int32 * CDDPM_MAX_LEVELS),
        ("num_res_blocks", C.c_int32),
        ("num_attention_resolutions", C.c_int32),
        ("attention_resolutions", C.c_int32 * CDDPM_MAX_LEVELS),
        ("head_channels", C.c_int32),
        ("cond_dim", C.c_int32),
        ("timesteps", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_h", C.c_int32),
        ("max_w", C.c_int32),
    ]


# every symbol include/cddpm.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _u64, _u32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_size_t
_fp = C.c_void_p   # raw device / host float pointers are passed as integers
SYMBOLS = {
    "cddpm_create": (_i, [C.POINTER(_vp), C.POINTER(UnetDesc), _i]),
    "cddpm_destroy": (None, [_vp]),
    "cddpm_last_error": (C.c_char_p, [_vp]),
    "cddpm_workspace_bytes": (_sz, [C.POINTER(UnetDesc)]),
    "cddpm_num_weights": (_i, [_vp]),
    "cddpm_weight_name": (C.c_char_p, [_vp, _i]),
    "cddpm_weight_numel": (_i64, [_vp, _i]),
    "cddpm_load_weights": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(_fp), C.POINTER(_i64), _i]),
    "cddpm_set_schedule": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _i, _i]),
    "cddpm_prepare_cond": (_i, [_vp, _fp, _i, _vp]),
    "cddpm_unet_forward": (_i, [_vp, _fp, _fp, _i, _fp, _i, _i, _i, _vp]),
    "cddpm_reverse": (_i, [_vp, _fp, _fp, _u64, _u64, _i, _i, _i, _i, _vp]),
    "cddpm_reverse_range": (_i, [_vp, _fp, _fp, _u64, _u64, _i, _i, _i, _i, _i, _vp]),
    "cddpm_p_sample": (_i, [_vp, _fp, _fp, _u64, _u64, _i, _i, _i, _i, _vp]),
    "cddpm_ddim_step": (_i, [_vp, _fp, _fp, _u64, _u64, _i, C.c_float, C.c_float, C.c_float, _i, _i, _i, _i, _i, _vp]),
    "cddpm_noise_fill": (_i, [_vp, _fp, _u64, _u32, _i, _u64, _i, _i, _i, _vp]),
    "cddpm_residual_postprocess": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "cddpm_simplex_fill": (_i, [_vp, _fp, _i64, _i, _i, _i, _i, C.c_double, C.c_double, _vp]),
    "cddpm_q_sample": (_i, [_vp, _fp, _fp, _fp, _i, _fp, _fp, _i, _fp, _i, _i, _i, _vp]),
    "cddpm_set_clip_denoised": (_i, [_vp, _i]),
    "cddpm_set_accumulation_switch": (_i, [_vp, _i]),
    "cddpm_set_profiling": (_i, [_vp, _i]),
    "cddpm_get_profile": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(_i64)]),
    "cddpm_num_blocks": (_i, [_vp]),
    "cddpm_block_name": (C.c_char_p, [_vp, _i]),
    "cddpm_set_tap": (_i, [_vp, _i, _fp]),
    "cddpm_block_shape": (_i, [_vp, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "cddpm_op_conv": (_i, [_vp, _fp, _i, _fp, _i, _fp, _i, _i, _fp, _fp, _i, _i, _fp, _i, _fp, _i, _i, _i, _vp]),
    "cddpm_op_conv_skip": (_i, [_vp, _fp, _i, _fp, _i, _fp, _fp, _i, _fp, _i, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_conv_gn": (_i, [_vp, _fp, _i, _fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_conv_bench": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_double),
                                 C.POINTER(_u64)]),
    "cddpm_op_gn_coef": (_i, [_vp, _fp, _i, _fp, _i, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "cddpm_op_attention": (_i, [_vp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_encoder_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "cddpm_encoder_destroy": (None, [_vp]),
    "cddpm_encoder_last_error": (C.c_char_p, [_vp]),
    "cddpm_encoder_num_weights": (_i, []),
    "cddpm_encoder_load_weights": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(_fp), C.POINTER(_i64), _i]),
    "cddpm_encoder_forward": (_i, [_vp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_conv_dgrad": (_i, [_vp, _fp, _i, _fp, _i, _i, _fp, _i, _i, _i, _vp]),
    "cddpm_op_conv_wgrad": (_i, [_vp, _fp, _i, _fp, _i, _fp, _i, _i, _fp, _i, _i, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_bias_grad": (_i, [_vp, _fp, _i64, _i, _fp, _vp]),
    "cddpm_op_attention_backward": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_linear_backward": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _vp]),
    "cddpm_op_linear": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _vp]),
    "cddpm_op_conv_in1": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "cddpm_op_head": (_i, [_vp, _fp, _fp, _fp, C.c_float, _fp, _fp, _i, _i, _i, _i, _vp]),
    "cddpm_op_pool_act": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "cddpm_op_unpool2": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "cddpm_op_sumpool2": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "cddpm_op_add_inplace": (_i, [_vp, _fp, _fp, _i64, _vp]),
    "cddpm_op_chan_image_corr": (_i, [_vp, _fp, _fp, _i, _fp, _i, _fp, _i, _i, _i, _i, _vp]),
    "cddpm_op_head_dgrad": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "cddpm_op_loss": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, C.c_float, _fp, _fp, _vp]),
    "cddpm_op_adam": (_i, [_vp, _fp, _fp, _fp, _fp, _i64, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _vp]),
    "cddpm_set_train_precision": (_i, [_i]),
    "cddpm_get_train_precision": (_i, []),
    "cddpm_op_grad_check": (_i, [_vp, _fp, _i64, _vp, _vp]),
    "cddpm_op_guard_commit": (_i, [_vp, _vp, C.c_float, C.c_float, _vp]),
    "cddpm_op_adam_guarded": (_i, [_vp, _fp, _fp, _fp, _fp, _i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp]),
    "cddpm_op_gn_silu_backward": (_i, [_vp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _vp]),
    "cddpm_op_enc_pack_w": (_i, [_vp, _fp, _i, _i, _i, _fp, _fp, _vp]),
    "cddpm_op_enc_conv": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cddpm_op_enc_conv_wgrad": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cddpm_op_enc_stem": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_enc_stem_wgrad": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_enc_bn_forward": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _i, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _i64, _i, _i, _vp]),
    "cddpm_op_enc_bn_backward": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _i64, _i, _i, _vp]),
    "cddpm_op_enc_maxpool": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "cddpm_op_enc_avgpool": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "cddpm_op_set_scratch": (_i, [_vp, _sz]),
    "cddpm_op_absmax": (_i, [_vp, _fp, _i64, _fp, _vp]),
    "cddpm_op_pack_conv": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cddpm_op_pack_conv_batch": (_i, [_vp, _vp, _i, _i64, _vp]),
    "cddpm_op_conv_packed": (_i, [_vp, _fp, _i, _fp, _i, _fp, _i, _i, _vp, _i, _fp, _i, _i, _fp, _i, _fp, _i, _fp, _i, _vp, _fp, _fp, _i, _i, _i, _vp]),
    "cddpm_op_gn_coef_rec": (_i, [_vp, _fp, _i, _i, _fp, _i, _i, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "cddpm_stat_records": (_i, [_i, _i, _i]),
    "cddpm_packed_conv_bytes": (_sz, [_i, _i, _i]),
    "cddpm_pack_conv_weights": (_i, [_fp, _i, _i, _i, _vp, C.POINTER(_i)]),
}

_lib = None


class CddpmLibraryError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and bind every entry point. Raises CddpmLibraryError -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise CddpmLibraryError(
            f"{path} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    # torch must come first: it ships its own libamdhip64 / libhsa-runtime64, and the process must end up with
    # ONE HIP runtime. Loaded first, ours would pull /opt/rocm's copy in and torch would then see no device.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise CddpmLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CddpmLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
