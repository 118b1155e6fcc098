"""ctypes binding of libmpgan_hip.so (the C ABI declared in include/mpgan_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPGAN_LIB_PATH") or os.path.join(_HERE, "libmpgan_hip.so")   # override: kernel experiments
_lib = None


class ConvGeomC(C.Structure):
    _fields_ = [("n", C.c_int32), ("in_dhw", C.c_int32 * 3), ("out_dhw", C.c_int32 * 3),
                ("cin", C.c_int32), ("cout", C.c_int32), ("k", C.c_int32 * 3),
                ("stride", C.c_int32 * 3), ("pad", C.c_int32 * 3), ("transposed", C.c_int32),
                ("flags", C.c_int32), ("min_blocks", C.c_int32)]


class PeerTapsC(C.Structure):
    _fields_ = [("z_peer", C.c_void_p), ("ld_peer", C.c_int32), ("scale_peer", C.c_void_p),
                ("shift_peer", C.c_void_p), ("coef", C.c_void_p)]


class NormFoldC(C.Structure):
    _fields_ = [("acc", C.c_void_p), ("replicas", C.c_int32), ("cstride", C.c_int32), ("count", C.c_int64),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p)]


class PrologueC(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("n_stride", C.c_int32),
                ("act", C.c_int32), ("slope", C.c_float), ("slope_ptr", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float
_G = C.POINTER(ConvGeomC)
_PR = C.POINTER(PrologueC)
_I3 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); must list EVERY symbol include/mpgan_hip.h declares
SIGNATURES = {
    "mpgan_last_error": (C.c_char_p, []),
    "mpgan_abi_version": (_I, []),
    "mpgan_conv_stats_rows": (_I, [_G, _I]),
    "mpgan_conv_forward": (_I, [_G, _P, _I, _P, _P, _PR, _P, _I, _I, _P, _P, _I, _P]),
    "mpgan_conv_backward_data": (_I, [_G, _P, _I, _P, _P, _I, _P, _I, _P]),
    "mpgan_conv_variant": (_I, [_G, _I, _I]),
    "mpgan_conv_bwd_stats_rows": (_I, [_G]),
    "mpgan_conv_variant_bf16": (_I, [_G, _I]),
    "mpgan_conv_bwd_stats_rows_bf16": (_I, [_G]),
    "mpgan_conv_backward_data_stats_bf16": (_I, [_G, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _F, _P, _P]),
    "mpgan_conv_backward_data_stats": (_I, [_G, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _F, _P, _P]),
    "mpgan_conv_wgrad_workspace": (_L, [_G]),
    "mpgan_conv_backward_weight": (_I, [_G, _P, _I, _PR, _P, _I, _P, _P, _F, _P, _L, _P]),
    "mpgan_pack_weights": (_I, [_P, _P, _P, _I, _L, _P]),
    "mpgan_stats_chunks": (_I, [_L, _I]),
    "mpgan_channel_stats": (_I, [_P, _I, _I, _L, _I, _P, _P]),
    "mpgan_norm_finalize": (_I, [_P, _I, _I, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mpgan_norm_finalize_strided": (_I, [_P, _I, _I, _I, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mpgan_norm_from_running": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P]),
    "mpgan_norm_from_running_multi": (_I, [_P, _I, _P]),
    "mpgan_epi_vectors_multi": (_I, [_P, _I, _P]),
    "mpgan_conv_forward_act": (_I, [_G, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "mpgan_norm_act_add": (_I, [_P, _I, _PR, _P, _I, _PR, _I, _L, _I, _I, _P, _I, _P]),
    "mpgan_norm_bwd_reduce": (_I, [_P, _I, _P, _I, _PR, _P, _P, C.POINTER(PeerTapsC), _I, _L, _I, _P, _P]),
    "mpgan_tap_l1_partials": (_I, []),
    "mpgan_tap_l1": (_I, [_P, _I, _PR, _P, _I, _PR, _L, _I, _P, _P, _P]),
    "mpgan_conv_splitk_workspace": (_L, [_G]),
    "mpgan_conv_forward_splitk": (_I, [_G, _P, _I, _P, _P, _PR, _P, _L, _P, _I, _P]),
    "mpgan_sigmoid_forward": (_I, [_P, _I, _P, _P]),
    "mpgan_norm_bwd_finalize": (_I, [_P, _I, _I, _I, _L, _I, _P, _P, _P, _P, _P, _P]),
    "mpgan_norm_bwd_apply": (_I, [_P, _I, _P, _I, _PR, _P, _P, _P, _P, C.POINTER(PeerTapsC), _I, _L, _I, _P, _I, _P]),
    "mpgan_reduce_partials": (_I, [_P, _I, _I, _I, _P, _F, _P]),
    "mpgan_add_tanh": (_I, [_P, _P, _L, _I, _P, _P]),
    "mpgan_tanh_backward": (_I, [_P, _P, _L, _P, _P]),
    "mpgan_axpby": (_I, [_P, _F, _P, _F, _L, _P, _P]),
    "mpgan_copy_slice": (_I, [_P, _I, _P, _I, _L, _I, _I, _P]),
    "mpgan_linear1_partials": (_I, [_I]),
    "mpgan_linear1_forward": (_I, [_P, _PR, _I, _L, _I, _P, _P, _P, _P, _P, _P]),
    "mpgan_bce_forward": (_I, [_P, _P, _I, _P, _P]),
    "mpgan_bce_backward": (_I, [_P, _P, _I, _P, _P, _P]),
    "mpgan_sigmoid_backward": (_I, [_P, _P, _I, _P, _P]),
    "mpgan_scale_by_device_scalar": (_I, [_P, _P, _L, _P, _P]),
    "mpgan_weighted_sum": (_I, [_P, _P, _I, _P, _P]),
    "mpgan_debug_stamps": (_I, [_P, _L, _L]),
    "mpgan_debug_stamps_used": (_L, []),
    "mpgan_debug_clock_khz": (_I, []),
    "mpgan_linear1_backward": (_I, [_P, _PR, _I, _L, _I, _P, _P, _P, _P, _P, _F, _P]),
    "mpgan_sigmoid_bce": (_I, [_P, _I, _F, _F, _P, _P, _P, _P]),
    "mpgan_l1_partials": (_I, []),
    "mpgan_l1_loss": (_I, [_P, _P, _L, _F, _P, _P, _P, _P]),
    "mpgan_metric_partials": (_I, []),
    "mpgan_rescale_minmax": (_I, [_P, _L, _F, _F, _I, _P, _P, _P, _P]),
    "mpgan_image_errors": (_I, [_P, _P, _L, _F, _P, _P, _P]),
    "mpgan_percentile_workspace": (_L, []),
    "mpgan_percentiles": (_I, [_P, _L, C.POINTER(C.c_double), _I, _P, _L, _P, _P]),
    "mpgan_scale_intensity_range": (_I, [_P, _L, _P, _F, _F, _I, _P, _P]),
    "mpgan_resample_to_identity_grid": (_I, [_P, _I3, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             _I3, C.c_double, _P, _P]),
    "mpgan_ssim_workspace": (_L, [_I3]),
    "mpgan_ssim": (_I, [_P, _P, _I3, _F, _P, _L, _P, _P]),
    "mpgan_adam_step": (_I, [_P, _P, _P, _P, _L, C.c_double, C.c_double, C.c_double, C.c_double, _I, _F, _P]),
    "mpgan_patch_gather": (_I, [_P, _I, _I3, _P, _I, _I3, _P, _P]),
    "mpgan_patch_scatter_add": (_I, [_P, _I, _I3, _P, _I, _I3, _P, _P]),
    "mpgan_zero_bytes": (_I, [_P, _L, _P]),
    "mpgan_conv_acc_supported": (_I, [_G, _I]),
    "mpgan_conv_fold_supported": (_I, [_G]),
    "mpgan_conv_forward_fold": (_I, [_G, _P, _I, _P, _P, _PR, C.POINTER(NormFoldC), _P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "mpgan_norm_act_add_fold": (_I, [_P, _I, _PR, C.POINTER(NormFoldC), _P, _I, _PR, _I, _L, _I, _I, _P, _I, _P]),
    # bf16 storage path (config C5)
    "mpgan_conv_stats_rows_bf16": (_I, [_G]),
    "mpgan_conv_forward_bf16": (_I, [_G, _P, _I, _P, _P, _P, _P, _I, _P]),
    "mpgan_conv_backward_data_bf16": (_I, [_G, _P, _I, _P, _P, _I, _P]),
    "mpgan_conv_wgrad_workspace_bf16": (_L, [_G]),
    "mpgan_conv_wgrad_variant_bf16": (_I, [_G]),
    "mpgan_conv_backward_weight_bf16": (_I, [_G, _P, _I, _P, _I, _P, _F, _P, _L, _P]),
    "mpgan_conv_forward_f32_to_bf16": (_I, [_G, _P, _I, _P, _P, _P, _P, _I, _P]),
    "mpgan_conv_backward_data_bf16_to_f32": (_I, [_G, _P, _I, _P, _P, _I, _P]),
    "mpgan_conv_wgrad_workspace_bf16dy": (_L, [_G]),
    "mpgan_conv_backward_weight_bf16dy": (_I, [_G, _P, _I, _P, _I, _P, _P, _F, _P, _L, _P]),
    "mpgan_pack_weights_bf16": (_I, [_P, _P, _P, _I, _L, _P]),
    "mpgan_norm_act_bf16": (_I, [_P, _I, _P, _P, _F, _L, _I, _P, _I, _I, _P]),
    "mpgan_norm_bwd_rows_bf16": (_I, [_L, _I]),
    "mpgan_norm_bwd_reduce_bf16": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _F, _L, _I, _P, _P]),
    "mpgan_norm_bwd_apply_bf16": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _F, _L, _I, _P, _I, _P, _P]),
}


def lib():
    """Return the loaded library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so): import it FIRST so this
        # library's libamdhip64 dependency binds to the runtime torch initialises.  Loaded the
        # other way round the process holds two runtimes and our launches see "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP hot path has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mpgan_last_error().decode()
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
