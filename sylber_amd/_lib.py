"""ctypes binding of libsylber_hip.so (include/sylber_hip.h).  There is no CPU fallback: if the
HIP library is missing or fails to load, importing the product path raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build and nothing else: no environment variable swaps the library (round 6; development A/Bs of
# two builds go through tools/with_lib.py, which calls use_library() before anything is loaded)
LIB_PATH = os.path.join(_HERE, "libsylber_hip.so")
_DEV_LIB = False        # set by use_library(): an older A/B build may lack the newest entry points
MAX_LAYERS = 12

c_float_p = POINTER(c_float)


class SylberLayerWeights(ctypes.Structure):
    _fields_ = [(n, c_float_p) for n in (
        "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b", "ln1_w", "ln1_b",
        "ff1_w", "ff1_b", "ff2_w", "ff2_b", "ln2_w", "ln2_b")]


class SylberMlpHidden(ctypes.Structure):
    _fields_ = [(n, c_float_p) for n in ("lin_w", "lin_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b", "ln_w", "ln_b")]


class SylberMlpWeights(ctypes.Structure):
    """mirror of SylberMlpWeights in include/sylber_hip.h"""
    _fields_ = [("input_dim", c_int32), ("output_dim", c_int32), ("num_hidden", c_int32), ("hidden_dims", c_int32 * 4),
                ("hidden", SylberMlpHidden * 4), ("out_w", c_float_p), ("out_b", c_float_p)]


class SylberWeights(ctypes.Structure):
    _fields_ = [("num_layers", c_int32), ("conv_w", c_float_p * 7), ("gn_w", c_float_p), ("gn_b", c_float_p),
                ("fp_ln_w", c_float_p), ("fp_ln_b", c_float_p), ("fp_w", c_float_p), ("fp_b", c_float_p),
                ("pos_w", c_float_p), ("pos_b", c_float_p), ("enc_ln_w", c_float_p), ("enc_ln_b", c_float_p),
                ("layers", SylberLayerWeights * MAX_LAYERS)]


EXPORTS = {
    "sylber_num_frames": (c_int32, [c_int32]),
    "sylber_padded_frames": (c_int32, [c_int32]),
    "sylber_set_option": (c_int, [c_void_p, c_int32, c_int32]),
    "sylber_create": (c_int, [POINTER(SylberWeights), c_int, c_int, POINTER(c_void_p)]),
    "sylber_destroy": (None, [c_void_p]),
    "sylber_last_error": (c_char_p, []),
    "sylber_forward": (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int32, c_int32, c_void_p, c_void_p]),
    "sylber_segment": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "sylber_set_stop_stage": (c_int, [c_void_p, c_int32]),
    "sylber_set_profiling": (c_int, [c_void_p, c_int32]),
    "sylber_set_graph_mode": (c_int, [c_void_p, c_int32]),
    "sylber_get_profile": (c_int, [c_void_p, POINTER(c_char_p), POINTER(c_float), c_int32]),
    "sylber_workspace_bytes": (c_int64, [c_void_p]),
    "sylber_get_fp16_audit": (c_int, [c_void_p, POINTER(c_char_p), POINTER(ctypes.c_uint32), POINTER(c_float), c_int32]),
    "sylber_op_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_int32, c_void_p]),
    "sylber_op_linear16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_void_p]),
    "sylber_op_conv3": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "sylber_op_linear_resln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                       c_int32, c_void_p]),
    "sylber_op_mx_quantize": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "sylber_op_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "sylber_ingest_num_frames": (c_int64, [c_int64, c_int32]),
    "sylber_ingest_workspace_bytes": (c_int64, [c_int32]),
    "sylber_ingest": (c_int, [c_void_p, c_int32, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "sylber_flac_info": (c_int, [c_void_p, c_int64, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "sylber_flac_decode": (c_int, [c_void_p, c_int64, c_void_p, c_int64, POINTER(c_int64)]),
    "sylber_km_workspace_floats": (c_int64, [c_int32, c_int32, c_int32]),
    "sylber_km_assign": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "sylber_km_decode": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "sylber_mlp_create": (c_int, [POINTER(SylberMlpWeights), c_int, POINTER(c_void_p)]),
    "sylber_mlp_destroy": (None, [c_void_p]),
    "sylber_condition_workspace_floats": (c_int64, [c_void_p, c_int32, c_int32]),
    "sylber_condition": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "sylber_condition_features": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "sylber_op_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
}
# include/sylber_hip_dev.h: development aids (tools/ only)
DEV_EXPORTS = {
    "sylber_debug_gemm_bench": (c_int, [c_int32] * 8 + [POINTER(c_float)]),
    "sylber_debug_gemm_pick": (c_int, [c_int32] * 8),
    "sylber_debug_gemm_trace": (c_int, [c_int32] * 6 + [POINTER(ctypes.c_uint64), POINTER(c_float)]),
    "sylber_debug_attention_bench": (c_int, [c_int32] * 4 + [POINTER(c_float)]),
    "sylber_debug_poison_workspace": (c_int, [c_void_p, c_int32]),
}
OPT_GEMM_TILE, OPT_ATTN_QUERIES_PER_WAVE, OPT_GEMM_PERSISTENT = 1, 2, 3

_LIB = None


class SylberHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SylberHipError(
                "libsylber_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python sylber_amd/build.py`; there is no CPU fallback on the product path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(EXPORTS.items()) + list(DEV_EXPORTS.items()):
            if _DEV_LIB and not hasattr(lib, name):
                continue                 # an OLDER build loaded for a same-box A/B (tools/with_lib.py) may lack the newest entry points
            fn = getattr(lib, name)      # AttributeError if the ABI drifted from include/sylber_hip.h
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def use_library(path: str) -> None:
    """development only (tools/with_lib.py): load another build of the library (a reference build for a same-box A/B, or the
    experiments build with its timing kernels).  Must be called before the first load()."""
    global LIB_PATH, _DEV_LIB
    if _LIB is not None:
        raise SylberHipError("use_library() after the library was loaded")
    LIB_PATH, _DEV_LIB = os.path.abspath(path), True


def check(status: int, what: str) -> None:
    if status != 0:
        raise SylberHipError("%s failed: %s" % (what, load().sylber_last_error().decode()))
