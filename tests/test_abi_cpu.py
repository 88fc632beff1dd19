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


def test_product_ops_refuse_cpu_tensors():
    import pytest
    import torch
    import viditq_amd  # noqa: F401
    from viditq_amd import ops
    with pytest.raises(ops.VQError):
        ops.rowquant(torch.zeros(1, 4, 64, dtype=torch.float16))
