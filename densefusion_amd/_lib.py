"""ctypes binding of libdfusion_hip.so (the C ABI declared in include/dfusion.h).

This is the only way the Python host layer reaches the GPU for the hot path.  There is no CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DF_DEV_LIB=1 (tests that compare kernel variants): the -DDF_DEV build, the only one that reads development switches from the environment
LIB_PATH = os.path.join(_HERE, "libdfusion_hip_dev.so" if os.environ.get("DF_DEV_LIB") else "libdfusion_hip.so")
_lib = None

_vp, _i, _i64, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t

class ConvDesc(ctypes.Structure):
    """Mirror of df_conv_desc (include/dfusion.h)."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("in_", "wgt", "bias", "res", "prelu", "out")] +
                [(n, ctypes.c_int32) for n in ("B", "H", "W", "Cin", "in_ld", "in_coff", "OH", "OW", "Cout", "out_ld",
                                               "out_coff", "res_ld", "res_coff", "KH", "KW", "stride", "pad", "dil",
                                               "act")] +
                [("splitk_ws", ctypes.c_void_p), ("splitk_ws_bytes", ctypes.c_size_t)])


# symbol -> (restype, argtypes); must list every function include/dfusion.h declares
SIGNATURES = {
    "df_last_error": (ctypes.c_char_p, []),
    "df_version": (_i, []),
    "df_knn_device": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "df_knn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "knn_device": (None, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "df_shader_clock_mhz": (_i, [ctypes.POINTER(ctypes.c_double), _vp]),
    "df_posenet_create": (_vp, [_i, _i]),
    "df_refiner_create": (_vp, [_i, _i]),
    "df_net_destroy": (None, [_vp]),
    "df_net_num_params": (_i, [_vp]),
    "df_net_param_info": (_i, [_vp, _i, ctypes.c_char_p, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i)]),
    "df_net_load_param": (_i, [_vp, ctypes.c_char_p, _vp, _i64]),
    "df_posenet_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "df_posenet_forward": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "df_posenet_multi_workspace_bytes": (_sz, [_vp, _i, _vp, _vp, _vp]),
    "df_posenet_forward_multi": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "df_refiner_workspace_bytes": (_sz, [_vp, _i]),
    "df_refiner_forward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "df_estimate_workspace_bytes": (_sz, [_vp, _vp, _i, _i, _i]),
    "df_estimate_poses": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "df_estimate_multi_workspace_bytes": (_sz, [_vp, _vp, _i, _vp, _vp, _vp]),
    "df_estimate_poses_multi": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "df_loss_forward": (_i, [_vp] * 6 + [_i, _i, _f, _i] + [_vp] * 7),
    "df_loss_forward_frames": (_i, [_i] + [_vp] * 7 + [_i, _i, _f] + [_vp] * 7),
    "df_loss_refine_forward": (_i, [_vp] * 5 + [_i, _i, _i] + [_vp] * 5),
    "df_loss_backward": (_i, [_vp] * 8 + [_i, _i, _f, _f] + [_vp] * 4),
    "df_loss_refine_backward": (_i, [_vp] * 5 + [_i, _f] + [_vp] * 3),
    "df_add_metric": (_i, [_vp] * 4 + [_i, _i, _vp, _vp]),
    "df_ycb_distances": (_i, [_vp] * 3 + [_i, _i, _vp, _vp, _vp]),
    "df_act_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "df_maxpool3s2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "df_maxpool2x2_idx": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "df_maxunpool2x2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "df_maxpool3s2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "df_adaptive_avgpool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "df_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "df_logsoftmax": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "df_dropout2d_mask": (_i, [_vp, _i64, ctypes.c_uint, _f, _vp]),
    "df_channel_scale": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp]),
    "df_gather_rows": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _i, _vp]),
    "df_colmean": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "df_sigmoid": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "df_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp]),
    "df_preprocess_objects": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "df_conv2d_nhwc": (_i, [ctypes.POINTER(ConvDesc), _vp]),
    "df_conv2d_nhwc_multi": (_i, [ctypes.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp]),
    "df_conv2d_wgrad_multi_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc), _i, _vp, _vp, _vp]),
    "df_conv2d_wgrad_nhwc_multi": (_i, [ctypes.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "df_conv_last_splitk": (_i, []),
    "df_conv3x3_winograd_scratch_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "df_conv3x3_winograd_nhwc": (_i, [ctypes.POINTER(ConvDesc), _vp, ctypes.c_size_t, _vp]),
    "df_wino_route": (_i, [_i, _i, _i, _i, _i]),
    "df_trainer_create": (_vp, [_i, _i, _i]),
    "df_trainer_destroy": (None, [_vp]),
    "df_trainer_flat_numel": (_i64, [_vp]),
    "df_trainer_num_params": (_i, [_vp]),
    "df_trainer_param_info": (_i, [_vp, _i, ctypes.c_char_p, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i)]),
    "df_trainer_pack_param": (_i, [_vp, ctypes.c_char_p, _vp, _vp, _vp]),
    "df_trainer_unpack_param": (_i, [_vp, ctypes.c_char_p, _vp, _vp, _vp]),
    "df_posenet_train_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "df_posenet_train_step": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i] + [_vp] * 6 + [_i, _vp, _f, _i, ctypes.c_uint] + [_vp] * 9 + [_sz, _vp]),
    "df_posenet_train_multi_workspace_bytes": (_sz, [_vp, _i, _vp, _vp, _vp, _i]),
    "df_posenet_train_step_multi": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp] + [_vp] * 5 + [_i, _vp, _f, _i, ctypes.c_uint] + [_vp] * 9 + [_sz, _vp]),
    "df_trainer_set_splitk": (_i, [_vp, _i]),
    "df_trainer_profile": (_i, [_vp, _i]),
    "df_trainer_profile_read": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i)]),
    "df_refiner_train_workspace_bytes": (_sz, [_vp, _i, _i]),
    "df_refiner_train_step": (_i, [_vp, _vp, _vp, _i64, _i] + [_vp] * 5 + [_i, _vp] + [_vp] * 4 + [_sz, _vp]),
    "df_conv3x3_winograd_tile_scratch_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc), _i]),
    "df_conv3x3_winograd_tile_nhwc": (_i, [ctypes.POINTER(ConvDesc), _i, _vp, ctypes.c_size_t, _vp]),
    "df_conv2d_dgrad_nhwc": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp]),
    "df_conv2d_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "df_conv2d_wgrad_nhwc": (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "df_net_debug_taps": (_i, [_vp, _i]),
    "df_net_debug_tap_read": (_i, [_vp, ctypes.c_char_p, _vp, _i64, ctypes.POINTER(_i64)]),
    "df_net_profile": (_i, [_vp, _i]),
    "df_net_profile_read": (_i, [_vp] + [ctypes.POINTER(ctypes.c_double)] * 4 + [ctypes.POINTER(_i)]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` at the repo root (needs hipcc). "
                "densefusion_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().df_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdfusion_hip {what} failed ({status}): {msg}")


def current_stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:                      # older / newer torch without the private accessor
        return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """``torch.cuda.device(device)`` only when `device` is not already current (the common single-GPU-per-process case
    costs nothing)."""
    import torch
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def dptr(t) -> int:
    """Device pointer of a contiguous CUDA(HIP) tensor."""
    if not t.is_cuda:
        raise RuntimeError("densefusion_amd ops need device tensors (no CPU path); got a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError("densefusion_amd ops need contiguous tensors")
    return t.data_ptr()
