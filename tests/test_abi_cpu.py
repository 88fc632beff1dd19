"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/viditq.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "viditq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    import __graft_entry__ as ge
    ge.build()
    import viditq_amd  # noqa: F401
    from viditq_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
        assert s in _lib.SIGNATURES, "ctypes binding missing for %s" % s
    for s in _lib.SIGNATURES:
        assert s in syms, "%s bound but not declared in include/viditq.h" % s
    assert lib.vq_version() >= 100
    assert lib.vq_strerror(0) == b"ok"
    assert b"shape" in lib.vq_strerror(-2)


def test_ctypes_signatures_have_the_headers_arity_and_kinds():
    """Every binding in _lib.SIGNATURES has exactly the parameter count of its declaration in include/viditq.h, and
    pointer / integer / float kinds agree position by position (an ABI edit that forgets the binding - or the other
    way round - would otherwise pass garbage without any error)."""
    import ctypes as C
    import viditq_amd  # noqa: F401
    from viditq_amd import _lib
    src = open(os.path.join(ROOT, "include", "viditq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(vq_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S))
    assert set(decls) == set(_lib.SIGNATURES)
    for name, params in decls.items():
        params = " ".join(params.split())
        plist = [] if params in ("", "void") else [q.strip() for q in params.split(",")]
        _, argtypes = _lib.SIGNATURES[name]
        assert len(plist) == len(argtypes), "%s: header has %d parameters, binding %d" % (name, len(plist), len(argtypes))
        for i, (decl, ct) in enumerate(zip(plist, argtypes)):
            if "*" in decl:
                kind = C.c_void_p
            elif re.match(r"^(const\s+)?float\b", decl):
                kind = C.c_float
            elif re.match(r"^(const\s+)?long\b", decl):
                kind = C.c_long
            else:
                kind = C.c_int
            assert ct is kind, "%s parameter %d (%s): bound as %s" % (name, i, decl, ct.__name__)


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (error behaviour of the ABI)."""
    import viditq_amd  # noqa: F401
    from viditq_amd import _lib
    lib = _lib.load()
    assert lib.vq_rowquant(None, None, 0, 1, None, None, None, None, None, None, None, None, None, 0,
                           1, 1, 8, 128, 8, None, None) == -1
    assert lib.vq_gemm_i8(None, None, None, None, None, None, None, None, None, None, 0, None, None,
                          0, 1, 1, 1, 128, 8, 0, 0, None) == -1
    one = ctypes.c_void_p(16)  # non-null dummy; rejected by the shape checks before any dereference
    assert lib.vq_rowquant(one, None, 0, 1, None, None, one, one, one, one, None, None, None, 0,
                           1, 1, 12, 128, 8, None, None) == -2
    assert lib.vq_rowquant(one, None, 0, 1, None, None, one, one, one, one, None, None, None, 0,
                           1, 1, 8, 128, 9, None, None) == -4
    assert lib.vq_attn_temporal(one, one, one, one, 1, 17, 4, 4, 72, 8, 8, 1.0, None) == -2


def _args(name, ptr, ints=None, longs=None):
    """Arguments for `name` from its binding: every pointer = `ptr`, every int 1 / long 8 / float 1.0 unless a position is
    given in `ints` / `longs` (position = index in the parameter list)."""
    import ctypes as C
    from viditq_amd import _lib
    out = []
    for i, ct in enumerate(_lib.SIGNATURES[name][1]):
        if ct is C.c_void_p:
            out.append(ptr)
        elif ct is C.c_float:
            out.append(1.0)
        elif ct is C.c_long:
            out.append((longs or {}).get(i, 8))
        else:
            out.append((ints or {}).get(i, 1))
    return out


def test_every_entry_point_rejects_null_and_bad_shapes_without_gpu():
    """Error behaviour of the whole ABI (include/viditq.h: 0 ok / negative code, never throws, nothing dereferenced or
    launched before the checks): for EVERY compute entry point, null pointers -> VQ_EINVAL, a non-positive size ->
    VQ_EINVAL, and - with non-null dummies that the checks must not dereference - one shape / alignment violation ->
    VQ_ESHAPE and one unsupported width / mode -> VQ_EUNSUP.  Positions are parameter indices of the header."""
    import viditq_amd  # noqa: F401
    from viditq_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    compute = [n for n in _lib.SIGNATURES if n not in ("vq_version", "vq_strerror", "vq_last_hip_error")]
    assert len(compute) >= 19
    for name in compute:
        fn = getattr(lib, name)
        assert fn(*_args(name, None)) == -1, name                       # null pointers
    # a zero extent with non-null pointers (position of one size parameter per entry point)
    zero_size = {"vq_rowquant": 15, "vq_gelu_rowquant": 8, "vq_ln_modulate_rowquant": 13, "vq_rowquant_smooth_multi": 8,
                 "vq_smooth_reciprocal": 2, "vq_fakequant_act": 9, "vq_epsfill_fixup": 7, "vq_pack_weight": 8,
                 "vq_weight_minmax": 4, "vq_gemm_i8": 14, "vq_gemm_i8_stamped": 11, "vq_gemm_i8_batched": 11, "vq_gemm_i8_grouped": 12,
                 "vq_attn_fwd": 5, "vq_attn_temporal": 6, "vq_attn_temporal_rowquant": 13, "vq_adaln_table": 4,
                 "vq_linear_f16": 4, "vq_cfg_ddim_step": 4}
    for name, pos in zero_size.items():
        assert getattr(lib, name)(*_args(name, one, ints={pos: 0})) == -1, name
    assert lib.vq_smooth_div_check(one, one, one, one, 0, None) == -1
    # shape / alignment: {position: value} on top of a consistent base (C = 64, Kp = 128, widths 8)
    shape = {
        "vq_rowquant": ({14: 1, 15: 4, 16: 60, 17: 128, 18: 8}, -2),                 # C % 8
        "vq_gelu_rowquant": ({7: 1, 8: 4, 9: 64, 10: 100, 11: 8}, -2),               # Kp % 128
        "vq_ln_modulate_rowquant": ({4: 1, 12: 1, 13: 4, 14: 64, 15: 0, 16: 8}, -2),  # Kp < C
        "vq_rowquant_smooth_multi": ({1: 1, 8: 4, 9: 62, 10: 128, 11: 8}, -2),
        "vq_fakequant_act": ({7: 1, 8: 1, 9: 4, 10: 60, 11: 8, 12: 0}, -2),
        "vq_epsfill_fixup": ({6: 1, 7: 70000, 8: 64, 9: 64, 10: 8}, -2),             # L > 65535
        "vq_pack_weight": ({8: 4, 9: 64, 10: 64, 11: 8}, -2),                        # Kp % 128
        "vq_gemm_i8": ({10: 64, 13: 1, 14: 4, 15: 62, 16: 64, 17: 128, 18: 8, 19: 0, 20: 0}, -2),       # N % 4
        "vq_gemm_i8_batched": ({10: 1, 11: 4, 12: 64, 13: 64, 14: 120, 15: 8}, -2),
        "vq_gemm_i8_stamped": ({10: 64, 11: 4, 12: 64, 13: 64, 14: 120}, -2),         # Kp % 128
        "vq_gemm_i8_grouped": ({0: 2, 11: 64, 12: 4, 13: 64, 14: 64, 15: 128, 16: 8}, -2),             # ldo < 2 N
        "vq_attn_temporal": ({4: 1, 5: 17, 6: 4, 7: 4, 8: 72}, -2),                  # T > 16
        "vq_attn_temporal_rowquant": ({11: 1, 12: 16, 13: 4, 14: 4, 15: 72, 17: 100}, -2),
        "vq_linear_f16": ({4: 4, 5: 64, 6: 60, 10: 0, 11: 0}, -2),                   # K % 8
    }
    for name, (ints, code) in shape.items():
        longs = {i: 128 for i, ct in enumerate(_lib.SIGNATURES[name][1]) if ct is ctypes.c_long}
        assert getattr(lib, name)(*_args(name, one, ints=ints, longs=longs)) == code, name
    assert lib.vq_attn_fwd(*_args("vq_attn_fwd", one, ints={4: 1, 5: 4, 6: 4, 7: 1, 8: 72}, longs={10: 12})) == -2   # row % 8
    assert lib.vq_gemm_i8_stamped(*_args("vq_gemm_i8_stamped", one, ints={10: 64, 11: 300, 12: 64, 13: 64, 14: 128},
                                         longs={16: 159})) == -2            # two tiles need 160 stamps
    unsup = {
        "vq_rowquant": {14: 1, 15: 4, 16: 64, 17: 128, 18: 9},
        "vq_gelu_rowquant": {7: 3, 8: 4, 9: 64, 10: 128, 11: 8},                    # B > 2: GEMM epilogue route
        "vq_ln_modulate_rowquant": {4: 1, 12: 1, 13: 4, 14: 64, 15: 128, 16: 1},
        "vq_rowquant_smooth_multi": {1: 1, 8: 4, 9: 64, 10: 128, 11: 12},
        "vq_fakequant_act": {7: 1, 8: 1, 9: 4, 10: 64, 11: 1, 12: 0},
        "vq_epsfill_fixup": {6: 1, 7: 4, 8: 64, 9: 64, 10: 9},
        "vq_pack_weight": {8: 4, 9: 64, 10: 128, 11: 16},
        "vq_weight_minmax": {4: 4, 5: 64, 6: 1, 7: 0},
        "vq_gemm_i8": {10: 64, 13: 1, 14: 4, 15: 64, 16: 64, 17: 128, 18: 8, 19: 7, 20: 0},           # epilogue kind
        "vq_gemm_i8_batched": {10: 1, 11: 4, 12: 64, 13: 64, 14: 128, 15: 4},        # batched form: 8-bit images only
        "vq_gemm_i8_grouped": {0: 2, 11: 128, 12: 4, 13: 64, 14: 64, 15: 128, 16: 9},
        "vq_linear_f16": {4: 4, 5: 64, 6: 64, 10: 2, 11: 2},                         # act pair the models never use
    }
    for name, ints in unsup.items():
        longs = {i: 128 for i, ct in enumerate(_lib.SIGNATURES[name][1]) if ct is ctypes.c_long}
        assert getattr(lib, name)(*_args(name, one, ints=ints, longs=longs)) == -4, name
    for code in (0, -1, -2, -3, -4, -99):
        assert lib.vq_strerror(code)


def test_product_ops_refuse_cpu_tensors():
    """No CPU / eager fallback anywhere in the operator layer: every wrapper of viditq_amd.ops raises VQError on a CPU
    tensor instead of computing something else."""
    import pytest
    import torch
    import viditq_amd  # noqa: F401
    from viditq_amd import ops

    def h(*s):
        return torch.zeros(*s, dtype=torch.float16)

    def f(*s):
        return torch.ones(*s, dtype=torch.float32)
    cases = {
        "rowquant": lambda: ops.rowquant(h(1, 4, 64)),
        "rowquant_multi": lambda: ops.rowquant_multi(h(1, 4, 64), [f(64)]),
        "gelu_rowquant": lambda: ops.gelu_rowquant(h(1, 4, 64)),
        "ln_modulate_rowquant": lambda: ops.ln_modulate_rowquant(h(1, 4, 64), f(1, 64), f(1, 64)),
        "fakequant_act": lambda: ops.fakequant_act(h(1, 4, 64)),
        "weight_minmax": lambda: ops.weight_minmax(h(8, 64), 8),
        "pack_weight": lambda: ops.pack_weight(h(8, 64), f(8, 1), f(8, 1), 8),
        "attn_fwd": lambda: ops.attn_fwd(h(1, 4, 72), h(1, 4, 72), h(1, 4, 72), h(1, 4, 72), 1, 4, 4, 1, 72, 288, 72, 288, 72,
                                         288, 72),
        "attn_temporal": lambda: ops.attn_temporal(h(4, 72), h(4, 72), h(4, 72), h(4, 72), 1, 4, 1, 1, 72, 72, 72),
        "attn_temporal_rowquant": lambda: ops.attn_temporal_rowquant(h(64, 64), h(64, 64), h(64, 64), 1, 16, 4, 4, 16, 64),
        "adaln_table": lambda: ops.adaln_table(h(6, 64), h(1, 384)),
        "linear_f16": lambda: ops.linear_f16(h(4, 64), h(8, 64)),
        "cfg_ddim_step": lambda: ops.cfg_ddim_step(f(1, 8, 4), f(1, 8, 4), f(1, 4, 4), 4.0, 1.0, 1.0, 1.0, 0.5),
        "gemm_i8_stamped": lambda: ops.gemm_i8_stamped(
            ops.QAct(torch.zeros(4, 128, dtype=torch.int8), f(4), torch.zeros(4, dtype=torch.int32),
                     torch.zeros(4, dtype=torch.int32), 64, 8),
            ops.PackedWeight(torch.zeros(8, 128, dtype=torch.int8), f(8), torch.zeros(8, dtype=torch.int32),
                             torch.zeros(8, dtype=torch.int32), 8, 64, 128, 8)),
        "smooth_rcp": lambda: ops.smooth_rcp(f(64)),
        "smooth_div_check": lambda: ops.smooth_div_check(f(64), f(64)),
        "epsfill_fixup": lambda: ops.epsfill_fixup(torch.zeros(1, dtype=torch.int32), h(1, 4, 64), None, h(8, 64), None,
                                                   h(1, 4, 8), 8),
    }
    for name, call in cases.items():
        with pytest.raises(ops.VQError):
            call()
        assert name in dir(ops)
