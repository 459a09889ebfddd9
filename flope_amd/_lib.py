"""ctypes binding of libflope_amd.so (include/flope_amd.h).  Fails loudly when the
library has not been built -- there is deliberately no fallback path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLOPE_AMD_LIB") or os.path.join(_HERE, "lib", "libflope_amd.so")   # override: A/B builds only

DT_BF16, DT_F16, DT_F32 = 0, 1, 2
IN_F32_NCHW, IN_BF16_NHWC, IN_F16_NHWC, IN_U8_NHWC = 0, 1, 2, 3
STAGE_STEM, STAGE_POOL, STAGE_FEAT, STAGE_HIDDEN = 0, 1, 10, 11


def STAGE_LAYER(li: int, bi: int) -> int:
    return 2 + (li - 1) * 2 + bi


# every symbol include/flope_amd.h declares: name -> (restype, argtypes)
_P, _I, _F, _D = C.c_void_p, C.c_int, C.c_float, C.c_double
SIGNATURES = {
    "flope_create": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "flope_destroy": (_I, [_P]),
    "flope_last_error": (C.c_char_p, [_P]),
    "flope_load_weights": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_I), C.POINTER(_P)]),
    "flope_forward": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "flope_forward_poses": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "flope_extract_features": (_I, [_P, _P, _I, _I, _P, _P]),
    "flope_procrustes": (_I, [_P, _P, _I, _P]),
    "flope_profile_timeline": (_I, [_P, _P, _P, _I]),
    "flope_nullify_yaw": (_I, [_P, _P, _I, _P]),
    "flope_compose_pose": (_I, [_P, _P, _I, _I, _P, _P]),
    "flope_crop_resize_mask": (_I, [_P, _P, _I, _I, _P, _I, _I, _I, _P, _P]),
    "flope_lanczos4_table": (_I, [_I, _I, _P, _P, _P]),
    "flope_merge_masks_resize": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P]),
    "flope_depth_lift": (_I, [_P, _I, _P, _I, _I, _F, _F, _F, _P, _I, C.POINTER(_F), _P, _P, _P, _P, _P]),
    "flope_read_stage": (_I, [_P, _I, _I, _P, C.POINTER(C.c_int64), _P]),
    "flope_set_option": (_I, [_P, C.c_char_p, _I]),
    "flope_debug_read_ws": (_I, [_P, _P, C.c_size_t, C.c_size_t]),
    "flope_debug_pk16": (_I, [_I, _I, _P, _P, _I, _P]),
    "flope_forward_flops": (_D, [_P, _I]),
    "flope_forward_launches": (_I, [_P]),
    "flope_profile_read": (_I, [_P, C.POINTER(_F), _I]),
    "flope_launch_info": (_I, [_P, _I, _I, C.c_char_p, _I, C.POINTER(_D)]),
    "flope_describe_plan": (_I, [_P, C.c_char_p, _I]),
    "flope_version": (C.c_char_p, []),
    "flope_engine_geometry": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "flope_frame_create": (_I, [_P, _I, _I, _I, _I, C.POINTER(_P)]),
    "flope_frame_destroy": (_I, [_P]),
    "flope_frame_last_error": (C.c_char_p, [_P]),
    "flope_frame_select": (_I, [_P, _I, _P, _P, _I, _P]),
    "flope_frame_enqueue": (_I, [_P, _I, _P, _P, _P, _I, _F, C.POINTER(_F), _F, _F, _P]),
    "flope_frame_finish": (_I, [_P, _I, _P, _I]),
    "flope_frame_to_poses": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _F, C.POINTER(_F), _F, _F, _P, _I, _P]),
    "flope_frame_read_boxes": (_I, [_P, _I, _P, _P, _I]),
    "flope_stream_create_cu_mask": (_I, [_I, C.POINTER(C.c_uint32), _I, C.POINTER(_P)]),
    "flope_stream_destroy": (_I, [_I, _P]),
    "flope_yolo_create": (_I, [_I, _I, _I, _I, _I, C.POINTER(_P)]),
    "flope_yolo_destroy": (_I, [_P]),
    "flope_yolo_last_error": (C.c_char_p, [_P]),
    "flope_yolo_input_size": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "flope_yolo_load_weights": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_I), C.POINTER(_P)]),
    "flope_yolo_detect": (_I, [_P, _P, _F, _F, _I, _P, _P, _P, _P]),
    "flope_yolo_forward": (_I, [_P, _P, _P]),
    "flope_yolo_read_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _P]),
    "flope_yolo_set_option": (_I, [_P, C.c_char_p, _I]),
    "flope_yolo_profile": (_I, [_P, _P, _I, C.c_char_p, _I, _P]),
    "flope_yolo_flops": (_D, [_P]),
    "flope_yolo_launches": (_I, [_P]),
    "flope_yolo_graph_cache_size": (_I, [_P]),
    "flope_tf_create": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "flope_tf_destroy": (_I, [_P]),
    "flope_tf_last_error": (C.c_char_p, [_P]),
    "flope_tf_load_weights": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_I), C.POINTER(_P)]),
    "flope_tf_forward": (_I, [_P, _P, _I, _I, _P, _P]),
    "flope_tf_set_option": (_I, [_P, C.c_char_p, _I]),
    "flope_tf_forward_flops": (_D, [_P, _I, _I]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles the HIP runtime, and the process must end up with ONE runtime instance -- loading
    # this library before torch left its hipGetDeviceCount without devices (seen with build() then smoke() in one process)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"flope_amd: {LIB_PATH} is missing -- build it with `make` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    msg = load().flope_last_error(handle)
    return msg.decode() if msg else ""


def check(rc: int, handle=None) -> None:
    if rc != 0:
        raise RuntimeError(f"flope_amd error {rc}: {last_error(handle)}")
