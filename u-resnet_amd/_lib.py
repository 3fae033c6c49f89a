"""ctypes binding of include/uresnet_hip.h.  No fallback: a missing library is an ImportError."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liburesnet_hip.so")


class ursn_config(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("spatial", C.c_int32 * 3), ("cin", C.c_int32),
                ("base_filters", C.c_int32), ("num_class", C.c_int32), ("num_strides", C.c_int32),
                ("max_batch", C.c_int32), ("trainable", C.c_int32), ("use_weight", C.c_int32),
                ("bn_eps", C.c_float), ("act_dtype", C.c_int32)]


class ursn_sizes(C.Structure):
    _fields_ = [("n_params", C.c_int64), ("n_tensors", C.c_int64), ("n_layers", C.c_int64),
                ("workspace_bytes", C.c_int64)]


class ursn_param_info(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("offset", C.c_int64), ("nelem", C.c_int64),
                ("rank", C.c_int32), ("shape", C.c_int32 * 5)]


class ursn_layer_info(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("transposed", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("relu", C.c_int32), ("w_offset", C.c_int64),
                ("beta_offset", C.c_int64)]


class ursn_conv_desc(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("n", C.c_int32), ("in_sp", C.c_int32 * 3), ("cin", C.c_int32),
                ("cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("transposed", C.c_int32),
                ("in_cstride", C.c_int32), ("out_cstride", C.c_int32), ("algo", C.c_int32),
                ("in_split", C.c_int32), ("in2_cstride", C.c_int32), ("x2", C.c_void_p), ("dx2", C.c_void_p),
                ("pw_dy", C.c_void_p), ("pw_w", C.c_void_p), ("pw_dy_cstride", C.c_int32), ("dtype", C.c_int32),
                ("in_mean", C.c_void_p), ("in_rstd", C.c_void_p), ("in_beta", C.c_void_p),
                ("bs_z", C.c_void_p), ("bs_mean", C.c_void_p), ("bs_rstd", C.c_void_p), ("bs_beta", C.c_void_p),
                ("bs_z2", C.c_void_p), ("bs_mean2", C.c_void_p), ("bs_rstd2", C.c_void_p), ("bs_mask", C.c_void_p),
                ("bs_partial", C.c_void_p), ("bs_z_cstride", C.c_int32), ("bs_z2_cstride", C.c_int32),
                ("bs_relu", C.c_int32), ("in_relu", C.c_int32),
                ("vdz_z", C.c_void_p), ("vdz_coef", C.c_void_p), ("vdz_out", C.c_void_p), ("vdz_relu", C.c_int32)]


class ursn_bn_bf16_desc(C.Structure):
    _fields_ = [("voxels", C.c_int64), ("channels", C.c_int32), ("relu", C.c_int32),
                ("z", C.c_void_p), ("z_cstride", C.c_int32), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("beta", C.c_void_p),
                ("z2", C.c_void_p), ("z2_cstride", C.c_int32), ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("beta2", C.c_void_p),
                ("res", C.c_void_p), ("res_cstride", C.c_int32),
                ("y", C.c_void_p), ("y_cstride", C.c_int32), ("mask_out", C.c_void_p), ("cat", C.c_int32),
                ("dy", C.c_void_p), ("dy_cstride", C.c_int32), ("dy2", C.c_void_p), ("dy2_cstride", C.c_int32),
                ("mask", C.c_void_p), ("dz", C.c_void_p), ("dz_cstride", C.c_int32), ("dz2", C.c_void_p), ("dz2_cstride", C.c_int32),
                ("dbeta", C.c_void_p), ("dbeta2", C.c_void_p), ("dres", C.c_void_p), ("dres_cstride", C.c_int32),
                ("dres_accumulate", C.c_int32)]


class ursn_prof_rec(C.Structure):
    _fields_ = [("kernel", C.c_char * 48), ("layer", C.c_char * 96), ("pass_", C.c_int32), ("ms", C.c_float),
                ("flops", C.c_double), ("bytes", C.c_double), ("launches", C.c_int32), ("reserved_", C.c_int32)]


_P = C.c_void_p
_SIGS = {
    "ursn_abi_version": (C.c_int, []),
    "ursn_last_error": (C.c_char_p, []),
    "ursn_query": (C.c_int, [C.POINTER(ursn_config), C.POINTER(ursn_sizes)]),
    "ursn_query_layer": (C.c_int, [C.POINTER(ursn_config), C.c_int64, C.POINTER(ursn_layer_info)]),
    "ursn_query_concat": (C.c_int, [C.POINTER(ursn_config), C.c_int32, C.c_char_p, C.c_char_p, C.c_size_t]),
    "ursn_create": (C.c_int, [C.POINTER(ursn_config), _P, _P, _P, _P, _P, C.c_size_t, C.POINTER(_P)]),
    "ursn_destroy": (C.c_int, [_P]),
    "ursn_get_sizes": (C.c_int, [_P, C.POINTER(ursn_sizes)]),
    "ursn_param": (C.c_int, [_P, C.c_int64, C.POINTER(ursn_param_info)]),
    "ursn_zero_grad": (C.c_int, [_P, _P]),
    "ursn_accum_step": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(C.c_float), _P]),
    "ursn_apply_adam": (C.c_int, [_P, C.c_float, _P]),
    "ursn_eval": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(C.c_float), _P]),
    "ursn_infer": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.POINTER(C.c_float), _P]),
    "ursn_infer_labels": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.POINTER(C.c_float), _P]),
    "ursn_read_metrics": (C.c_int, [_P, C.POINTER(C.c_float), _P]),
    "ursn_get_adam_step": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ursn_set_adam_step": (C.c_int, [_P, C.c_int64]),
    "ursn_tensor": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                              C.POINTER(C.c_int32)]),
    "ursn_set_wgrad_overlap": (C.c_int, [_P, C.c_int32]),
    "ursn_profile_enable": (C.c_int, [_P, C.c_int32]),
    "ursn_profile_read": (C.c_int, [_P, C.POINTER(ursn_prof_rec), C.c_int64, C.POINTER(C.c_int64)]),
    "ursn_conv_forward": (C.c_int, [C.POINTER(ursn_conv_desc), _P, _P, _P, _P]),
    "ursn_conv_forward_stats": (C.c_int, [C.POINTER(ursn_conv_desc), _P, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P]),
    "ursn_conv_backward_data": (C.c_int, [C.POINTER(ursn_conv_desc), _P, _P, _P, C.c_int32, _P]),
    "ursn_conv_backward_weight": (C.c_int, [C.POINTER(ursn_conv_desc), _P, _P, _P, _P, C.c_size_t, _P]),
    "ursn_conv_wgrad_scratch_bytes": (C.c_size_t, [C.POINTER(ursn_conv_desc)]),
    "ursn_conv_bs_blocks": (C.c_int32, [C.POINTER(ursn_conv_desc)]),
    "ursn_conv_plan": (C.c_int, [C.POINTER(ursn_conv_desc), C.c_int32, C.c_char_p, C.c_size_t]),
    "ursn_last_kernel_name": (C.c_char_p, []),
    "ursn_bn_forward": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, C.c_int32, _P, _P,
                                  C.c_size_t, _P]),
    "ursn_bn_backward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, C.c_int32, _P,
                                   C.c_size_t, _P]),
    "ursn_bn_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "ursn_bn_bf16_forward": (C.c_int, [C.POINTER(ursn_bn_bf16_desc), _P]),
    "ursn_bn_bf16_backward": (C.c_int, [C.POINTER(ursn_bn_bf16_desc), _P, C.c_size_t, _P]),
    "ursn_bn_bf16_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "ursn_softmax_ce": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P, _P,
                                  C.POINTER(C.c_float), _P, C.c_size_t, _P]),
    "ursn_adam": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                            C.c_int64, _P]),
    "ursn_mfma_probe": (C.c_int, [C.c_int32, _P, _P]),
}
EXPORTS = tuple(_SIGS.keys())

ABI_VERSION = 7
_lib = None


def load():
    """Loads liburesnet_hip.so (built in-tree by __graft_entry__.build() / u-resnet_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- first, so that ONE HIP runtime (torch's bundled libamdhip64.so.7) serves both
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "liburesnet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback for the U-ResNet hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.ursn_abi_version() != ABI_VERSION:
        raise ImportError("liburesnet_hip.so ABI version mismatch")
    _lib = lib
    return lib


class UrsnError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise UrsnError((load().ursn_last_error() or b"").decode("utf-8", "replace") or ("error %d" % rc))
