"""ctypes binding of libviditq_hip.so (C ABI in include/viditq.h).

There is NO fallback: if the library cannot be loaded the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIDITQ_LIB=<path>: bind ANOTHER build of the same C ABI (smoke() checks a library it just built from source on the GPU
# box in a child process this way); the default is the in-tree library
LIB_PATH = os.environ.get("VIDITQ_LIB") or os.path.join(_HERE, "csrc", "libviditq_hip.so")

VQ_OK = 0
VQ_ST_EPSFILL = 1
EPI_NONE, EPI_GELU, EPI_GATE_RESID, EPI_RESID = 0, 1, 2, 3

_vp, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long

# name -> (restype, argtypes); mirrors include/viditq.h one to one
SIGNATURES = {
    "vq_version": (_i, []),
    "vq_strerror": (C.c_char_p, [_i]),
    "vq_last_hip_error": (_i, []),
    "vq_gelu_rowquant": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "vq_rowquant": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                         _i, _i, _i, _i, _i, _vp, _vp]),
    "vq_ln_modulate_rowquant": (_i, [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _i, _i, _i, _i, _i, _vp, _vp]),
    "vq_smooth_reciprocal": (_i, [_vp, _vp, _i, _vp, _vp]),
    "vq_rowquant_smooth_multi": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vq_smooth_div_check": (_i, [_vp, _vp, _vp, _vp, _l, _vp]),
    "vq_fakequant_act": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "vq_epsfill_fixup": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vq_pack_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vq_weight_minmax": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vq_gemm_i8_batched": (_i, [_vp] * 10 + [_i] * 6 + [_vp]),
    "vq_gemm_i8_grouped": (_i, [_i] + [_vp] * 10 + [_i] * 6 + [_vp]),
    "vq_gemm_i8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp,
                        _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vq_gemm_i8_stamped": (_i, [_vp] * 10 + [_i] * 5 + [_vp, _l, _vp]),
    "vq_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _l, _vp, _f, _vp]),
    "vq_attn_temporal": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _l, _l, _f, _vp]),
    "vq_attn_temporal_rowquant": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _l, _i, _f,
                                       _vp]),
    "vq_adaln_table": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "vq_linear_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _vp]),
    "vq_cfg_ddim_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _f, _f, _vp]),
}

_lib = None


class VQError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    """Load the shared library and bind every symbol include/viditq.h declares."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise VQError(
            "libviditq_hip.so not found at %s - build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(hipcc --offload-arch=gfx950); there is no CPU/eager fallback for the product path" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


CALLS = [0]      # C-ABI calls checked so far (graph.StepGraph differences it around a capture: the census of bench.py)


def check(code: int, what: str = ""):
    CALLS[0] += 1
    if code != VQ_OK:
        lib = load()
        msg = lib.vq_strerror(code).decode()
        extra = ""
        if code == -3:
            extra = " (hipError %d)" % lib.vq_last_hip_error()
        raise VQError("%s failed: %s%s" % (what or "viditq call", msg, extra))
