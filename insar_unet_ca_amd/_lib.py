"""ctypes binding of libinsar_hip.so (the C ABI declared in include/insar_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is
missing (or an entry point fails) we raise, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# INSAR_HIP_LIB points the loader at another build of the same ABI (A/B runs of kernel experiments)
LIB_PATH = os.environ.get("INSAR_HIP_LIB") or os.path.join(_HERE, "libinsar_hip.so")

F32, BF16 = 0, 1
ABI_VERSION = 6
IGEMM_OOB_ZERO = 1
IGEMM_PINGPONG = 2


class InsarAct(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C", C.c_int32), ("c_off", C.c_int32), ("c_len", C.c_int32), ("dtype", C.c_int32),
                ("_pad", C.c_int32)]


class InsarBstat(C.Structure):
    _fields_ = [("y", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


class InsarIgemm(C.Structure):
    _fields_ = [("x", InsarAct), ("y", InsarAct), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("stats", C.c_void_p), ("N", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("stride", C.c_int32), ("ntaps", C.c_int32), ("mode", C.c_int32),
                ("dy", C.c_int8 * 12), ("dx", C.c_int8 * 12), ("flags", C.c_int32), ("out_stride", C.c_int32),
                ("out_oy", C.c_int32), ("out_ox", C.c_int32), ("_pad", C.c_int32), ("add", C.c_void_p), ("bstat", InsarBstat), ("gate", C.c_void_p)]


class InsarWgrad(C.Structure):
    _fields_ = [("x", InsarAct), ("dy", InsarAct), ("tabx", C.c_void_p), ("tabdy", C.c_void_p),
                ("part", C.c_void_p), ("Mpad", C.c_int64), ("nsplit", C.c_int32), ("ntaps", C.c_int32),
                ("offx", C.c_int32 * 12), ("offdy", C.c_int32 * 12), ("tabx_tap_stride", C.c_int64)]


class InsarBnFinalize(C.Structure):
    _fields_ = [("part", C.c_void_p), ("rows", C.c_int64), ("count", C.c_int64), ("C", C.c_int32), ("training", C.c_int32),
                ("conv_bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p)]


class InsarSeFwd(C.Structure):
    _fields_ = [("part", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C", C.c_int32), ("Cr", C.c_int32), ("rows", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("pooled", C.c_void_p), ("sq", C.c_void_p), ("hid", C.c_void_p), ("gate", C.c_void_p)]


class InsarBnSeBwd(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("Cr", C.c_int32),
                ("use_se", C.c_int32),
                ("mean", C.c_void_p), ("invstd", C.c_void_p), ("pooled", C.c_void_p), ("sq", C.c_void_p),
                ("hid", C.c_void_p), ("gate", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("dw1", C.c_void_p), ("dw2", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("coefB", C.c_void_p), ("k1", C.c_void_p), ("k2", C.c_void_p),
                ("accumulate", C.c_int32), ("_pad", C.c_int32)]


class InsarCam(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("Cr", C.c_int32),
                ("rows", C.c_int32),
                ("psum", C.c_void_p), ("pmax", C.c_void_p), ("parg", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("avg", C.c_void_p), ("mx", C.c_void_p), ("arg", C.c_void_p), ("ha", C.c_void_p), ("hm", C.c_void_p),
                ("gate", C.c_void_p), ("coefB", C.c_void_p), ("dmax", C.c_void_p), ("ws", C.c_void_p),
                ("dw1", C.c_void_p), ("dw2", C.c_void_p), ("accumulate", C.c_int32), ("_pad", C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float
_AP = C.POINTER(InsarAct)

# name -> (argtypes); every function returns int except the two noted below
_SIGNATURES = {
    "insar_pack_nchw": [_P, _AP, _P],
    "insar_unpack_nchw": [_AP, _P, _P],
    "insar_weight_prep": [_P, _P, _I, _I, _I, _I, _L, _L, _L, _P],
    "insar_weight_prep_batch": [_P, _I, _L, _P],
    "insar_weight_prep_pair_batch": [_P, _I, _L, _P],
    "insar_igemm_num_mtiles": [_L, _I],
    "insar_igemm_tile_rows": [_L, _I],
    "insar_igemm_tile_cols": [_L, _I],
    "insar_igemm_tile_cols_dt": [_L, _I, _I],
    "insar_igemm": [C.POINTER(InsarIgemm), _P],
    "insar_conv3x3_flat_ok": [_AP, _I],
    "insar_conv3x3_flat_num_mtiles": [_AP],
    "insar_conv3x3_flat_rows_ok": [_AP, _I],
    "insar_conv3x3_flat2_rows_ok": [_AP, _I],
    "insar_conv3x3_flat_rows_dil_ok": [_AP, _I, _I],
    "insar_conv3x3_flat_stat_rows": [_AP, _I, _I],
    "insar_conv3x3_flat": [_AP, _AP, _P, _I, _P, _P],
    "insar_conv3x3_flat_bstat": [_AP, _AP, _P, _I, _P, C.POINTER(InsarBstat), _P],
    "insar_conv3x3_c64_ok": [_AP, _I],
    "insar_conv3x3_c64_rows": [_AP],
    "insar_conv3x3_c64_geometry": [_AP, _P],
    "insar_conv3x3_c64": [_AP, _AP, _P, _I, _P, _P],
    "insar_conv3x3_c64_bstat": [_AP, _AP, _P, _I, _P, C.POINTER(InsarBstat), _P],
    "insar_wgrad": [C.POINTER(InsarWgrad), _P],
    "insar_wgrad_tile": [_I, _I],
    "insar_wgrad_tile_pair": [_I, _I, _I],
    "insar_wgrad_conv3_tile": [_AP, _I],
    "insar_wgrad_conv3": [_AP, _AP, _P, _I, _P],
    "insar_wgrad_conv3x_tile": [_AP, _I],
    "insar_wgrad_conv3x": [_AP, _AP, _P, _I, _P],
    "insar_wgrad_conv3y_tile": [_AP, _I],
    "insar_wgrad_conv3y": [_AP, _AP, _P, _I, _P],
    "insar_wgrad_conv3k_tile": [_AP, _I],
    "insar_wgrad_conv3k_slices": [_AP, _I],
    "insar_wgrad_conv3k": [_AP, _AP, _P, _I, _P],
    "insar_wgrad_reduce": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "insar_wgrad_fold": [_P, _P, _L, _I, _I, _P],
    "insar_pixel_table": [_P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "insar_conv3x3_small_fwd": [_AP, _P, _AP, _P, _P],
    "insar_conv3x3_small_fwd_rows": [_AP, _AP],
    "insar_conv3x3_small_wgrad_blocks": [_I, _I],
    "insar_conv3x3_small_wgrad": [_AP, _AP, _P, _P],
    "insar_conv3x3_small_wgrad_fused_ok": [_P, _P],
    "insar_conv3x3_small_wgrad_fused": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "insar_colsum": [_P, _P, _I, _L, _I, _I, _P, _L, _P],
    "insar_colsum_ld": [_P, _P, _I, _L, _I, _L, _I, _P, _L, _P],
    "insar_colsum_partial": [_P, _P, _L, _I, _I, _P],
    "insar_bn_finalize": [C.POINTER(InsarBnFinalize), _P],
    "insar_bn_relu_apply": [_AP, _P, _P, _P, _AP, _I, _P],
    "insar_bn_relu_apply_pool": [_AP, _P, _P, _P, _AP, _AP, _I, _P],
    "insar_se_squeeze": [_AP, _P, _P, _P, _I, _I, _P],
    "insar_se_excite": [C.POINTER(InsarSeFwd), _P],
    "insar_bnrelu_bwd_reduce": [_AP, _AP, _P, _P, _P, _I, _I, _P],
    "insar_bnse_bwd_coef": [C.POINTER(InsarBnSeBwd), _P, _I, _P, _P, _P, _P, _I, _P],
    "insar_bnse_bwd_coef_fused": [C.POINTER(InsarBnSeBwd), _P, _I, _P, _P, _P, _P, _I, _P, _P],
    "insar_bn_bwd_coef": [C.POINTER(InsarBnSeBwd), _P, _L, _P, _P, _I, _P],
    "insar_bnrelu_bwd_apply": [_AP, _AP, _P, _P, _P, _P, _P, _P, _P, _P, _AP, _I, _P],
    "insar_bn_relu_apply_pool_arg": [_AP, _P, _P, _P, _AP, _AP, _P, _I, _P],
    "insar_bnrelu_bwd_reduce_pool": [_AP, _AP, _P, _AP, _P, _P, _P, _I, _I, _P],
    "insar_bnrelu_bwd_apply_pool": [_AP, _AP, _P, _AP, _P, _P, _P, _P, _P, _P, _P, _P, _AP, _I, _P],
    "insar_conv1x1_out_wgrad": [_AP, _P, _P, _I, _P, _P],
    "insar_bn_relu_apply_outc": [_AP, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "insar_conv1x1_out_wgrad_y": [_AP, _P, _P, _P, _P, _P, _I, _P, _P],
    "insar_bnrelu_bwd_reduce_outc": [_P, _P, _I, _AP, _P, _P, _P, _I, _I, _P, _P, _P],
    "insar_bnrelu_bwd_apply_outc": [_P, _P, _I, _AP, _P, _P, _P, _P, _P, _P, _P, _P, _AP, _I, _P],
    "insar_bnse_bwd_coef_stage": [C.POINTER(InsarBnSeBwd), _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "insar_bnrelu_bwd_apply_part": [_AP, _AP, _P, _P, _P, _P, _P, _P, _P, _P, _AP, _I, _P],
    "insar_cam_pool": [_AP, _P, _P, _P, _I, _P],
    "insar_cam_excite": [C.POINTER(InsarCam), _P],
    "insar_cam_bwd_coef": [C.POINTER(InsarCam), _P, _I, _P],
    "insar_cam_scatter_max": [_AP, _P, _P, _P],
    "insar_resize_bilinear_fwd": [_AP, _AP, _P],
    "insar_resize_bilinear_bwd": [_AP, _AP, _P],
    "insar_maxpool2_fwd": [_AP, _AP, _P],
    "insar_maxpool2_bwd": [_AP, _AP, _AP, _I, _P],
    "insar_conv1x1_out_fwd": [_AP, _P, _P, _P, _I, _P],
    "insar_conv1x1_out_bwd_blocks": [_I, _I],
    "insar_conv1x1_out_bwd": [_AP, _P, _P, _I, _AP, _P, _P],
    "insar_ce_blocks": [_L],
    "insar_cross_entropy": [_P, _P, _I, _I, _L, _L, _P, _P, _P, _P],
    "insar_dice": [_P, _P, _I, _I, _L, _L, _F, _P, _P, _P, _P],
    "insar_dice_ce": [_P, _P, _I, _I, _L, _L, _F, _F, _F, _P, _P, _P, _P],
    "insar_confusion": [_P, _P, _I, _I, _L, _L, _P, _P],
    "insar_adam_step": [_P, _P, _I, _I, _F, _F, _F, _F, _F, _F, _F, _P],
    "insar_scale_f32": [_P, _L, _F, _P],
    "insar_mul_dev_f32": [_P, _P, _L, _P, _P],
    "insar_adam_step_dev": [_P, _P, _I, _I, _F, C.c_double, C.c_double, _F, _P, _F, _P],
    "insar_pixel_table_taps": [_P, _L, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "insar_conv7x7s2_fwd_rows": [_I, _I],
    "insar_conv7x7s2_fwd": [_P, _I, _I, _P, _AP, _P, _P],
    "insar_conv7x7s2_wgrad_blocks": [_I, _I],
    "insar_conv7x7s2_wgrad": [_P, _I, _I, _AP, _P, _P],
    "insar_maxpool3s2_fwd": [_AP, _AP, _P, _P],
    "insar_maxpool3s2_bwd": [_AP, _P, _AP, _P],
    "insar_bn_add_relu": [_AP, _P, _P, _AP, _AP, _I, _P],
    "insar_relu_gate_bwd": [_AP, _AP, _AP, _P],
    "insar_sum_hw": [_AP, _AP, _F, _P],
    "insar_broadcast_hw": [_AP, _AP, _F, _I, _P],
    "insar_broadcast_hw_gate": [_AP, _AP, _AP, _F, _I, _P],
    "insar_dropout": [_AP, _AP, _P, C.c_uint64, _P, _F, _I, _P],
    "insar_bilinear_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "insar_bilinear_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "insar_tune_set": [C.c_char_p, _I],
    "insar_tune_get": [C.c_char_p],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["insar_version", "insar_last_error"])

_lib = None


class InsarError(RuntimeError):
    pass


def load():
    """Load libinsar_hip.so (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise InsarError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU / eager fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    lib.insar_version.restype = C.c_int
    lib.insar_version.argtypes = []
    lib.insar_last_error.restype = C.c_char_p
    lib.insar_last_error.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    if lib.insar_version() != ABI_VERSION:
        raise InsarError(f"libinsar_hip.so ABI {lib.insar_version()} != expected {ABI_VERSION}")
    _lib = lib
    for item in filter(None, os.environ.get("INSAR_TUNE", "").split(",")):
        k, _, v = item.partition("=")
        if lib.insar_tune_set(k.strip().encode(), int(v or 1)) != 0:
            raise InsarError(f"INSAR_TUNE: unknown knob {k!r}")
    return lib


_COUNT_ONLY = {"insar_tune_get", "insar_igemm_num_mtiles", "insar_igemm_tile_rows", "insar_igemm_tile_cols", "insar_igemm_tile_cols_dt", "insar_wgrad_tile", "insar_wgrad_tile_pair", "insar_wgrad_conv3_tile", "insar_wgrad_conv3x_tile", "insar_wgrad_conv3y_tile", "insar_wgrad_conv3k_tile", "insar_wgrad_conv3k_slices", "insar_conv3x3_flat_ok", "insar_conv3x3_flat_rows_ok", "insar_conv3x3_flat2_rows_ok", "insar_conv3x3_flat_rows_dil_ok", "insar_conv3x3_flat_num_mtiles", "insar_conv3x3_flat_stat_rows", "insar_conv3x3_c64_ok", "insar_conv3x3_c64_rows", "insar_conv3x3_c64_geometry", "insar_conv3x3_small_wgrad_blocks", "insar_conv3x3_small_wgrad_fused_ok", "insar_conv3x3_small_fwd_rows", "insar_conv1x1_out_bwd_blocks",
               "insar_ce_blocks", "insar_conv7x7s2_fwd_rows", "insar_conv7x7s2_wgrad_blocks"}


_TAPE = None      # while a launch tape is being recorded (tape.py): the list every launch is appended to


def call(name: str, *args) -> int:
    """Invoke an entry point; negative return codes raise InsarError(insar_last_error())."""
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*args)
    if name in _COUNT_ONLY:
        return rc
    if rc != 0:
        raise InsarError(f"{name} failed ({rc}): {lib.insar_last_error().decode(errors='replace')}")
    if _TAPE is not None:
        _TAPE.append((fn, args, name))
    return 0


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """The caller's HIP stream (torch's current stream on the current device): kernels are enqueued there, never synced.
    (torch.cuda.current_stream() builds a Stream object and resolves the device three times: 8 us per call, 0.6-2 ms of
    host time per training step; the raw accessors return the same handle in 0.3 us.)"""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise InsarError(f"unsupported compute dtype {dt} (float32 or bfloat16)")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def tune(name: str, value: int = None) -> int:
    """Set (or read) a kernel-variant knob of the library (include/insar_hip.h: insar_tune_set). INSAR_TUNE=name=v,name=v
    in the environment sets knobs when the library is first used (A/B runs of bench.py)."""
    if value is None:
        v = call("insar_tune_get", name.encode())
        if v == -1005:
            raise InsarError(f"unknown tuning knob {name!r}")
        return v
    call("insar_tune_set", name.encode(), int(value))
    return int(value)
