"""ctypes binding of tools/lab/libviditq_lab.so (retired GEMM variants, ablations, probes) for the measurement
scripts in tools/.  ``gemm_i8`` has the signature of viditq_amd.ops.gemm_i8 with a mandatory ``variant``."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import viditq_amd  # noqa: E402,F401
from viditq_amd import ops  # noqa: E402

_vp, _i = C.c_void_p, C.c_int
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libviditq_lab.so")
        if not os.path.exists(path):
            import importlib.util
            spec = importlib.util.spec_from_file_location("lab_build", os.path.join(HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build()
        _lib = C.CDLL(path)
        _lib.vq_lab_gemm_i8.restype = _i
        _lib.vq_lab_gemm_i8.argtypes = [_vp] * 10 + [_i, _vp, _vp] + [_i] * 8 + [_vp]
        _lib.vq_lab_gemm_12w.restype = _i
        _lib.vq_lab_gemm_12w.argtypes = [_vp] * 10 + [_i, _vp, _vp] + [_i] * 8 + [_vp]
        _lib.vq_lab_gemm_4w.restype = _i
        _lib.vq_lab_gemm_4w.argtypes = [_vp] * 10 + [_i, _vp, _vp] + [_i] * 8 + [_vp]
        _lib.vq_lab_gemm_loader.restype = _i
        _lib.vq_lab_gemm_loader.argtypes = [_vp] * 10 + [_i, _vp, _vp] + [_i] * 8 + [_vp]
        _lib.vq_lab_gemm_persist6.restype = _i
        _lib.vq_lab_gemm_persist6.argtypes = [_vp] * 10 + [_i, _vp, _vp] + [_i] * 8 + [_vp]
        _lib.vq_lab_gemm_sp.restype = _i
        _lib.vq_lab_gemm_sp.argtypes = [_vp] * 10 + [_i, _vp, _vp] + [_i] * 8 + [_vp]
        _lib.vq_probe_mfma_i8.argtypes = [_vp, _vp, _vp, _vp]
        _lib.vq_probe_stage_rate.argtypes = [_i, _vp, _i, _i, _i, _vp, _vp]
        _lib.vq_probe_mfma_rate.argtypes = [_i, _i, _i, _vp, _vp]
    return _lib


def _p(t):
    return None if t is None else t.data_ptr()


def gemm_i8(a, w, bias=None, out=None, epilogue=0, resid=None, gate=None, rows_per_gate=0, variant=11):
    M, N = a.rows, w.N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    rc = lib().vq_lab_gemm_i8(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs), _p(bias),
                              _p(out), out.stride(0), _p(resid), _p(gate), rows_per_gate, M, N, a.K, a.Kp, w.n_bits,
                              epilogue, variant, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("vq_lab_gemm_i8 variant %d: error %d" % (variant, rc))
    return out


def gemm_4w(a, w, bias=None, out=None, epilogue=0, resid=None, gate=None, rows_per_gate=0, variant=0, waves=4):
    """tools/lab/gemm_4w.hip: the ring kernel with four waves of 128 x 144 (one per SIMD), or (waves=12) twelve of 64 x 96"""
    M, N = a.rows, w.N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    fn = lib().vq_lab_gemm_12w if waves == 12 else lib().vq_lab_gemm_4w
    rc = fn(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs), _p(bias),
                              _p(out), out.stride(0), _p(resid), _p(gate), rows_per_gate, M, N, a.K, a.Kp, w.n_bits,
                              epilogue, variant, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("vq_lab_gemm_4w variant %d: error %d" % (variant, rc))
    return out


def gemm_loader(a, w, bias=None, out=None, epilogue=0, resid=None, gate=None, rows_per_gate=0, variant=0, mode=0):
    """tools/lab/gemm_loader.hip: mode 0 = 256 x 192 tile, 8 MFMA waves of 64 x 96 + 4 dedicated loader waves; mode 1 =
    256 x 288 tile, 8 waves, waves 0-3 issue every LDS-DMA piece; mode 2 = 256 x 192 tile without loader waves (control)"""
    M, N = a.rows, w.N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    rc = lib().vq_lab_gemm_loader(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs), _p(bias),
                                  _p(out), out.stride(0), _p(resid), _p(gate), rows_per_gate, M, N, a.K, a.Kp, mode,
                                  epilogue, variant, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("vq_lab_gemm_loader mode %d variant %d: error %d" % (mode, variant, rc))
    return out


def gemm_persist6(a, w, bias=None, out=None, epilogue=0, resid=None, gate=None, rows_per_gate=0, mode=2):
    """tools/lab/gemm_persist6.hip: mode 2 = the interior GEMM as persistent workgroups with next-tile prefetch (round 6),
    mode 1 = the same header's interior form"""
    M, N = a.rows, w.N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    rc = lib().vq_lab_gemm_persist6(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs), _p(bias),
                                    _p(out), out.stride(0), _p(resid), _p(gate), rows_per_gate, M, N, a.K, a.Kp, mode,
                                    epilogue, w.n_bits, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("vq_lab_gemm_persist6 mode %d: error %d" % (mode, rc))
    return out


def gemm_sp(a, w, bias=None, out=None, epilogue=0, resid=None, gate=None, rows_per_gate=0, variant=0, grid=0):
    """tools/lab/gemm_sp.hip: persistent 256 x 192 tiles, the finished tile's stores fed into the next tile's main loop"""
    M, N = a.rows, w.N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    rc = lib().vq_lab_gemm_sp(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs), _p(bias),
                              _p(out), out.stride(0), _p(resid), _p(gate), rows_per_gate, M, N, a.K, a.Kp, grid,
                              epilogue, variant, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("vq_lab_gemm_sp variant %d: error %d" % (variant, rc))
    return out


def probe_mfma_i8(a, b):
    out = torch.empty((32, 32), dtype=torch.int32, device=a.device)
    rc = lib().vq_probe_mfma_i8(_p(a), _p(b), _p(out), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out
