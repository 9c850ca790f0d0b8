"""ctypes binding of libdiga_hip.so (the C-ABI declared in include/diga_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdiga_hip.so")

MAX_TAPS = 18


class ConvDesc(C.Structure):
    """diga_conv_desc (include/diga_hip.h)."""
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
                ("stride", C.c_int32), ("ntaps", C.c_int32),
                ("dy", C.c_int16 * MAX_TAPS), ("dx", C.c_int16 * MAX_TAPS),
                ("cout_pad", C.c_int32), ("splitk", C.c_int32),
                ("out_f32", C.c_int32), ("tile", C.c_int32), ("nplanes", C.c_int32)]


class DigaError(RuntimeError):
    pass


_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol include/diga_hip.h declares
SIGNATURES = {
    "diga_version": (_i, []),
    "diga_last_error_string": (C.c_char_p, []),
    "diga_conv_stat_units": (_i, [C.POINTER(ConvDesc)]),
    "diga_conv_igemm_bf16": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "diga_conv_dam_ticket_words": (_i, [C.POINTER(ConvDesc)]),
    "diga_conv_igemm_bf16_dam": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "diga_conv_igemm_bf16_fused_in": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "diga_conv_igemm_bf16_apply": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "diga_stem_conv7x7s2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "diga_bn_finalize": (_i, [_vp, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "diga_bn_reduce_slab": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "diga_bn_finalize_sums": (_i, [_vp, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "diga_bn_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "diga_bn_finalize_apply": (_i, [_vp, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "diga_bn_stats_nchw_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "diga_bn_apply_nchw_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "diga_bn_relu_maxpool": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "diga_head_finish_pseudo_label": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i64, _vp]),
    "diga_head_tapsum_pseudo_label": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i64, _vp]),
    "diga_nearest_down_labels": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i64, _vp]),
    "diga_proto_pool_work_bytes": (_i64, [_i, _i64, _i, _i]),
    "diga_proto_pool": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "diga_proto_pool_frames_work_bytes": (_i64, [_i, _i, _i64, _i, _i]),
    "diga_proto_pool_frames": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "diga_proto_update": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp]),
    "diga_proto_update_frames": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp]),
    "diga_proto_classify_frames_work_bytes": (_i64, [_i, _i64, _i]),
    "diga_proto_classify_fuse_frames": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp]),
    "diga_proto_classify_work_bytes": (_i64, [_i64, _i]),
    "diga_proto_classify_fuse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp]),
    "diga_proto_classify_fuse_noalloc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp]),
    "diga_ccl_work_bytes": (_i64, [_i, _i, _i, _i]),
    "diga_ccl_label": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "diga_instance_pool_work_bytes": (_i64, [_i, _i64, _i, _i, _i]),
    "diga_instance_pool": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "diga_upsample_argmax_hist": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "diga_argmax_hist": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "diga_upsample_argmax_hist_frames": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "diga_xgmi_buffer_bytes": (_i64, [_i, _i64]),
    "diga_xgmi_create": (_i, [_i, _i, _i64, C.POINTER(_vp), _vp]),
    "diga_xgmi_connect": (_i, [_vp, _vp]),
    "diga_xgmi_destroy": (_i, [_vp]),
    "diga_xgmi_allreduce": (_i, [_vp, _vp, _i64, _i, _vp]),
    "diga_xgmi_check": (_i, [_vp]),
    "diga_xgmi_set_timeout_ms": (_i, [_vp, _i]),
    "diga_bn_exchange_finalize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "diga_preprocess_image": (_i, [_vp, _i, _vp, _i, _i, _i, _d, _d, _d, _i, _vp]),
    "diga_map_labels": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _vp]),
    "diga_stem_conv7x7s2_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "diga_bn_apply_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "diga_bn_relu_maxpool_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "diga_split_planes_f32": (_i, [_vp, _vp, _i, _i64, _vp]),
    "diga_proto_pool_f32": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "diga_proto_classify_fuse_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp]),
    "diga_nchw_f32_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "diga_nhwc_f32_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "diga_nhwc_bf16_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "diga_nchw_f32_to_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load libdiga_hip.so (once).  Raises DigaError when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DigaError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C diga_amd/csrc`). There is no CPU fallback.")
    # torch ships its own libamdhip64: import it first so that this library binds to the SAME HIP runtime
    # (two runtimes in one process -> "no ROCm-capable device is detected" on the second one).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc < 0:
        msg = load().diga_last_error_string().decode(errors="replace")
        raise DigaError(f"{what}: error {rc}: {msg}")
    return rc


def call(name: str, *args):
    """Call an int-returning entry point and raise on a negative status."""
    return check(getattr(load(), name)(*args), name)
