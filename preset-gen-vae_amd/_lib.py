"""ctypes binding of the C-ABI library declared in ``include/pgv_hip.h``.

This is the reference-side stub a maintainer would add: the reference has no FFI — its hot
path is torch ops called from ``model/*.py`` — so the binding is a plain ``ctypes.CDLL`` of ``libpgv_hip.so``.
There is no CPU fallback: if the library is missing or a call fails, a ``RuntimeError`` is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgv_hip.so")

PGV_ACT_NONE, PGV_ACT_LEAKY_RELU, PGV_ACT_HARDTANH = 0, 1, 2
PGV_PREZEROED = 1


class ConvDesc(Structure):
    """Mirror of ``pgv_conv_desc`` (include/pgv_hip.h)."""
    _fields_ = [(n, c_int32) for n in ("B", "Cb", "Hb", "Wb", "Cs", "Hs", "Ws", "kh", "kw", "stride", "pad",
                                        "flags")] + [("w_shadow", c_void_p)]


_P = c_void_p  # device pointers travel as integers
_DESC = POINTER(ConvDesc)


class BwdFuse(Structure):
    """Mirror of ``pgv_bwd_fuse``: BatchNorm + activation backward of the next-lower block fused into an
    input-gradient call."""
    _fields_ = [("a", c_void_p), ("coef", c_void_p), ("gbias", c_void_p), ("act", c_int32), ("slope", c_float),
                ("cls", c_void_p), ("gbias_copies", c_int32)]


_FUSE = POINTER(BwdFuse)


class BnSrc(Structure):
    """Mirror of ``pgv_bn_src``: the arguments of ``pgv_bn_finalize`` as a value, handed to the kernel that consumes the
    BatchNorm so that it finalizes it in its own prologue."""
    _fields_ = [("stats", c_void_p), ("n", c_int64), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
                ("momentum", c_float), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("num_batches_tracked", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("mean", c_void_p),
                ("rstd", c_void_p), ("stats_copies", c_int32)]


_BN = POINTER(BnSrc)


class CoefReq(Structure):
    """Mirror of ``pgv_coef_req``: the BatchNorm-backward coefficients of the block below, asked of a weight-gradient
    call."""
    _fields_ = [("lower_is_big", c_int32), ("cls", c_void_p), ("w", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("mean", c_void_p), ("rstd", c_void_p), ("n", c_int64), ("coef", c_void_p), ("ggamma", c_void_p),
                ("gbeta", c_void_p), ("scratch", c_void_p), ("cls_copies", c_int32)]


_COEF = POINTER(CoefReq)


class BiasReq(Structure):
    """Mirror of ``pgv_bias_req``: the block's bias gradient from its per-XCD partial copies, asked of its weight-gradient
    call."""
    _fields_ = [("copies", c_void_p), ("gbias", c_void_p), ("C", c_int32), ("accumulate", c_int32)]


_BIAS = POINTER(BiasReq)


class ParamsTables(Structure):
    """Mirror of ``pgv_params_tables``: the index tables of a PresetIndexesHelper (device pointers)."""
    _fields_ = [("n_num", c_int32), ("num_idx", c_void_p), ("num_rules", c_void_p), ("n_groups", c_int32), ("K", c_int32),
                ("cat_idx", c_void_p), ("cat_rules", c_void_p), ("n_rules", c_int32), ("rule_trig", c_void_p)]


_PT = POINTER(ParamsTables)
PGV_PARAMS_CCE, PGV_PARAMS_CCE_SOFTMAX, PGV_PARAMS_BCE = 0, 1, 2
PGV_PARAMS_COL_QUANTIZED, PGV_PARAMS_COL_ONEHOT_VALUE, PGV_PARAMS_COL_CLASS, PGV_PARAMS_COL_ONEHOT_CLASS = 0, 1, 2, 3

# name -> (restype, argtypes); must list every function include/pgv_hip.h declares (tests/test_abi.py checks).
SIGNATURES = {
    "pgv_abi_version": (c_int, []),
    "pgv_last_error": (c_char_p, []),
    "pgv_set_kernel_policy": (c_int, [c_int]),
    "pgv_conv_down": (c_int, [_DESC, _P, _P, _P, _P, _P, c_int, c_float, _P, _P, _P]),
    "pgv_conv_up": (c_int, [_DESC, _P, _P, _P, _P, _P, c_int, c_float, _P, _P, _P]),
    "pgv_conv_down_fused": (c_int, [_DESC, _P, _P, _P, _P, _P, c_int, c_float, _P, _P, _FUSE, _P]),
    "pgv_conv_up_fused": (c_int, [_DESC, _P, _P, _P, _P, _P, c_int, c_float, _P, _P, _FUSE, _P]),
    "pgv_conv_down_bn": (c_int, [_DESC, _P, _BN, _P, _P, c_int, c_float, _P, _P, _P]),
    "pgv_conv_up_bn": (c_int, [_DESC, _P, _BN, _P, _P, c_int, c_float, _P, _P, _P]),
    "pgv_conv_up_sqerr": (c_int, [_DESC, _P, _BN, _P, _P, _P, _P, c_int, c_float, _P, _P, c_float, _P, _P, _P, _P,
                                  ctypes.POINTER(c_int), _P]),
    "pgv_conv_weight_shadow_bytes": (c_int64, [_DESC]),
    "pgv_conv_weight_shadow": (c_int, [_DESC, _P, _P, _P]),
    "pgv_conv_weight_shadows": (c_int, [c_int, POINTER(_DESC), POINTER(c_void_p), POINTER(c_void_p), _P]),
    "pgv_conv_wgrad_workspace": (c_int64, [_DESC]),
    "pgv_conv_wgrad": (c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P]),
    "pgv_conv_wgrad_coef": (c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _COEF, _P]),
    "pgv_conv_wgrad_ex": (c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _COEF, _BIAS, _P]),
    "pgv_bn_stats": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "pgv_bn_finalize": (c_int, [_P, c_int, c_int64, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pgv_bn_finalize_src": (c_int, [_BN, c_int, _P]),
    "pgv_bn_eval_affine": (c_int, [_P, _P, _P, _P, c_float, c_int, _P, _P, _P]),
    "pgv_affine_nchw": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "pgv_bn1d_fwd": (c_int, [_P, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pgv_bn1d_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "pgv_bn_bwd_coef": (c_int, [_DESC, c_int, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "pgv_bn_bwd_coef_from_gy": (c_int, [_DESC, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, c_int,
                                        _P]),
    "pgv_conv_tap_sums": (c_int, [_DESC, c_int, _P, _P, _P, c_int, _P]),
    "pgv_conv_class_sums": (c_int, [_DESC, c_int, _P, _P, c_int, _P]),
    "pgv_act_bwd_coef": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, c_int, _P]),
    "pgv_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "pgv_bn_act_bwd_fusable": (c_int, [c_int, c_int, c_int]),
    "pgv_bn_act_bwd_fused": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, c_int, _P]),
    "pgv_act_bn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, c_int,
                               _P]),
    "pgv_gemm_workspace": (c_int64, [c_int, c_int, c_int]),
    "pgv_gemm": (c_int, [c_int, c_int, c_int, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, _P, c_int,
                         _P, c_int64, _P]),
    "pgv_sqerr_act_bwd": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_int, _P]),
    "pgv_sqerr_act_bwd_cls": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P,
                                      c_int, _P]),
    "pgv_colsum": (c_int, [_P, c_int, c_int, c_int64, _P, c_int, _P]),
    "pgv_dropout_mask": (c_int, [_P, c_uint64, c_float, c_int64, _P, _P]),
    "pgv_dropout_apply": (c_int, [_P, c_uint64, c_float, c_int64, _P, _P, _P, _P]),
    "pgv_dropout_fwd": (c_int, [_P, c_uint64, c_float, _P, c_int64, c_int, c_int64, _P, _P, _P, _P, _P]),
    "pgv_dropout_fwd_bn": (c_int, [_P, c_uint64, c_float, _P, c_int64, c_int, c_int64, _BN, _P, _P, _P]),
    "pgv_dropout_bwd": (c_int, [_P, c_uint64, c_float, c_int64, _P, _P, _P]),
    "pgv_dropout_bwd_bn_reduce": (c_int, [_P, c_uint64, c_float, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "pgv_dropout_bwd_colsum": (c_int, [_P, c_uint64, c_float, c_int, c_int, _P, _P, _P, c_int, _P]),
    "pgv_normal": (c_int, [_P, c_uint64, c_int64, _P, _P]),
    "pgv_rng_advance": (c_int, [_P, c_uint64, _P]),
    "pgv_mul": (c_int, [_P, _P, c_int64, _P, _P]),
    "pgv_reparam_kl_fwd": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "pgv_reparam_kl_fwd_rng": (c_int, [_P, _P, c_uint64, c_int, c_int, c_float, _P, _P, _P, c_int, _P]),
    "pgv_bn1d_reparam_fwd": (c_int, [_P, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P,
                                     c_uint64, c_float, _P, _P, _P, c_int, _P]),
    "pgv_bn1d_reparam_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P, c_int,
                                     _P]),
    "pgv_reparam_kl_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P, _P]),
    "pgv_sqerr_fwd": (c_int, [_P, _P, c_int64, c_float, _P, _P]),
    "pgv_sqerr_bwd": (c_int, [_P, _P, _P, c_int64, c_float, c_int, _P, _P]),
    "pgv_adam_step": (c_int, [_P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, _P]),
    "pgv_adam_tick": (c_int, [_P, _P, c_float, c_float, _P]),
    "pgv_step_tick": (c_int, [_P, _P, c_float, c_float, _P, c_uint64, _P, _P, _P, _P, _P, _P, _P]),
    "pgv_stft_mel": (c_int, [_P, c_int, c_int64, c_int, c_int, c_int, _P, c_float, _P, _P, _P, c_int, c_float,
                             c_float, c_float, _P, _P]),
    "pgv_stft": (c_int, [_P, c_int, c_int64, c_int, c_int, c_int, _P, c_float, _P, _P, _P, c_int, c_int, c_float,
                         c_float, c_float, _P, _P]),
    "pgv_fill": (c_int, [_P, c_int64, c_float, _P]),
    "pgv_params_loss": (c_int, [_P, _P, c_int, c_int, _PT, c_int, c_float, c_int, c_float, _P, _P, _P, c_int64, _P]),
    "pgv_params_columns": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pgv_axpy": (c_int, [c_int64, c_float, _P, _P, _P]),
    "pgv_copy": (c_int, [_P, _P, c_int64, _P]),
    "pgv_probe_mfma": (c_int, [c_int, c_int, _P, POINTER(c_int64), _P]),
    "pgv_probe_read": (c_int, [_P, c_int64, _P, _P]),
    # tuning knobs of the timing scripts (pgv_hip.h, last section)
    "pgv_dbg_set_gemm_tiles": (c_int, [c_int]),
    "pgv_dbg_set_gemm_variant": (c_int, [c_int]),
    "pgv_dbg_set_v2_down_variant": (c_int, [c_int]),
    "pgv_dbg_set_wgrad_bf16_variant": (c_int, [c_int]),
    "pgv_dbg_set_deep_bf16_variant": (c_int, [c_int]),
    "pgv_dbg_set_deep_bf16_stamps": (None, [_P]),
}

_lib = None


def load():
    """Load ``libpgv_hip.so`` (built by ``__graft_entry__.build()`` / ``build_ext.build()``); raise if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # torch first: libpgv_hip.so depends on libamdhip64 and must bind to the HIP runtime torch ships and initialises
    # (streams and device pointers are shared with it).  Loaded before torch, the dependency would resolve to a second
    # runtime (/opt/rocm) that never sees the device ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    """Turn a negative PGV_E_* code into a Python exception (error convention of the boundary)."""
    if rc != 0:
        msg = load().pgv_last_error()
        raise RuntimeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
