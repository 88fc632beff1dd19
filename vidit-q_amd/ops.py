"""Tensor-level wrappers over the C ABI (include/viditq.h).

PyTorch is plumbing here: device memory (torch tensors), the current HIP stream and nothing
else.  Every function enqueues one or a few HIP kernels on ``torch.cuda.current_stream()`` and
never synchronises.  There is no fallback path: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import EPI_GATE_RESID, EPI_GELU, EPI_NONE, EPI_RESID, VQError, check  # noqa: F401


def _L():
    return _lib.load()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise VQError("%s must be a GPU tensor (no CPU fallback in the product path)" % name)
    if t.dtype != dtype:
        raise VQError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise VQError("%s must be contiguous" % name)
    return t


def pad128(k: int) -> int:
    return (k + 127) // 128 * 128


@dataclass
class QAct:
    """A per-token-quantized activation: int8 codes (minus cx) + per-row dequant terms."""
    xq: torch.Tensor      # [rows, Kp] int8
    sx: torch.Tensor      # [rows] fp32 delta
    zx: torch.Tensor      # [rows] int32 zp - cx
    R: torch.Tensor       # [rows] int32 rowsum - K*zx
    K: int
    n_bits: int = 8
    zpf: Optional[torch.Tensor] = None

    @property
    def rows(self) -> int:
        return self.xq.shape[0]

    @property
    def Kp(self) -> int:
        return self.xq.shape[1]


@dataclass
class PackedWeight:
    """An offline-quantized weight: packed codes + per-out-channel dequant terms."""
    wq: torch.Tensor      # [N, Kp] int8  or [N, Kp/2] uint8 (n_bits <= 4)
    sw: torch.Tensor      # [N] fp32
    zw: torch.Tensor      # [N] int32
    cs: torch.Tensor      # [N] int32
    N: int
    K: int
    Kp: int
    n_bits: int

    def tensors(self):
        return [self.wq, self.sw, self.zw, self.cs]


def new_status(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


# --------------------------------------------------------------------------- quantizers
# reciprocal of a smooth-quant channel scale, keyed by the tensor (identity + in-place version): computed on first
# sight (one device round trip for the bad-channel count, so: in the eager warm-up pass, never under graph capture),
# None when the reciprocal form of x / s is not guaranteed bit-exact for some channel (the kernels then divide)
_RCP_CACHE: "dict[int, tuple]" = {}
# bumped whenever a device tensor that a captured HIP graph may reference by address is replaced (packed weights:
# qdiff/models/quant_layer.py; reciprocal vectors: below); graph.GraphedSampler drops its graphs when it moves
PACK_EPOCH = [0]


def smooth_rcp(s: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if s is None:
        return None
    key = id(s)
    ent = _RCP_CACHE.get(key)
    if ent is not None and ent[0]() is s and ent[1] == s._version:
        return ent[2]
    _req(s, torch.float32, "s")
    if torch.cuda.is_current_stream_capturing():
        return None                       # an unseen vector under capture: exact division, no host round trip
    r = torch.empty_like(s)
    bad = torch.zeros(1, dtype=torch.int32, device=s.device)
    check(_L().vq_smooth_reciprocal(_p(s), _p(r), s.numel(), _p(bad), _stream()), "vq_smooth_reciprocal")
    out = r if int(bad.item()) == 0 else None
    import weakref
    # entries of dead vectors go on every insert; a LIVE entry is never evicted (its reciprocal may be referenced by
    # address from a captured graph) - only replaced when its vector was modified in place, which moves the epoch
    for k in [k for k, v in _RCP_CACHE.items() if v[0]() is None]:
        _RCP_CACHE.pop(k, None)
    if ent is not None and ent[0]() is s:
        PACK_EPOCH[0] += 1
    _RCP_CACHE[key] = (weakref.ref(s), s._version, out)
    return out


def smooth_div_check(a: torch.Tensor, b: torch.Tensor):
    """(reciprocal-form quotient, IEEE quotient) of a / b elementwise - test hook."""
    _req(a, torch.float32, "a")
    _req(b, torch.float32, "b")
    fast, exact = torch.empty_like(a), torch.empty_like(a)
    check(_L().vq_smooth_div_check(_p(a), _p(b), _p(fast), _p(exact), a.numel(), _stream()), "vq_smooth_div_check")
    return fast, exact


def rowquant(x: torch.Tensor, n_bits: int = 8, s: Optional[torch.Tensor] = None,
             add_rows: Optional[torch.Tensor] = None, add_div: int = 1,
             status: Optional[torch.Tensor] = None, want_zp: bool = False,
             delta: Optional[torch.Tensor] = None, zp: Optional[torch.Tensor] = None, fast_div: bool = True) -> QAct:
    """Per-token quantizer of x [B, n_tok, C] fp16 (scales shared over B): dynamic min-max, or on a
    static calibrated grid when ``delta``/``zp`` (1 or n_tok entries) are given."""
    _req(x, torch.float16, "x")
    assert x.dim() == 3
    B, n_tok, Cc = x.shape
    Kp = pad128(Cc)
    rows = B * n_tok
    dev = x.device
    xq = torch.empty((rows, Kp), dtype=torch.int8, device=dev)
    sx = torch.empty(rows, dtype=torch.float32, device=dev)
    zx = torch.empty(rows, dtype=torch.int32, device=dev)
    R = torch.empty(rows, dtype=torch.int32, device=dev)
    zpf = torch.empty(rows, dtype=torch.float32, device=dev) if want_zp else None
    if s is not None:
        _req(s, torch.float32, "s")
        assert s.numel() == Cc
    n_add = 0
    if add_rows is not None:
        _req(add_rows, torch.float16, "add_rows")
        n_add = add_rows.shape[0]
    n_param = 0
    if delta is not None:
        delta = _req(delta.reshape(-1).contiguous(), torch.float32, "delta")
        zp = _req(zp.reshape(-1).contiguous(), torch.float32, "zp")
        n_param = delta.numel()
    s_rcp = smooth_rcp(s) if fast_div else None
    check(_L().vq_rowquant(_p(x), _p(add_rows), n_add, add_div, _p(s), _p(s_rcp), _p(xq), _p(sx), _p(zx), _p(R), _p(zpf),
                           _p(delta), _p(zp), n_param, B, n_tok, Cc, Kp, n_bits, _p(status), _stream()),
          "vq_rowquant")
    return QAct(xq, sx, zx, R, Cc, n_bits, zpf)


def rowquant_multi(x: torch.Tensor, smooth: Sequence[torch.Tensor], n_bits: int = 8,
                   status: Optional[torch.Tensor] = None) -> List[QAct]:
    """One QAct of x [1, n_tok, C] per smoothing vector (the q / k / v quantizers of a plan that balances each Linear
    against its own weight) in ONE launch when the reciprocal-form kernel covers the shape, else one launch each."""
    _req(x, torch.float16, "x")
    B, n_tok, Cc = x.shape
    rcps = [smooth_rcp(sm) for sm in smooth]
    Kp = pad128(Cc)
    G = len(smooth)
    if B != 1 or G > 3 or any(r is None for r in rcps) or Kp != Cc or Cc % 128 or not (768 <= Cc <= 1280) or n_tok < 2:
        return [rowquant(x, n_bits=n_bits, s=sm, status=status) for sm in smooth]
    dev = x.device
    outs = [QAct(torch.empty((n_tok, Kp), dtype=torch.int8, device=dev), torch.empty(n_tok, dtype=torch.float32, device=dev),
                 torch.empty(n_tok, dtype=torch.int32, device=dev), torch.empty(n_tok, dtype=torch.int32, device=dev),
                 Cc, n_bits) for _ in range(G)]
    keep = [_ptr_array([_p(sm) for sm in smooth]), _ptr_array([_p(r) for r in rcps])] + \
           [_ptr_array([getattr(o, f).data_ptr() for o in outs]) for f in ("xq", "sx", "zx", "R")]
    check(_L().vq_rowquant_smooth_multi(_p(x), G, *[C.cast(k, C.c_void_p) for k in keep], n_tok, Cc, Kp, n_bits,
                                        _p(status), _stream()), "vq_rowquant_smooth_multi")
    return outs


def gelu_rowquant(x: torch.Tensor, n_bits: int = 8, s: Optional[torch.Tensor] = None,
                  status: Optional[torch.Tensor] = None, fast_div: bool = True) -> QAct:
    """GELU(tanh) then the per-token quantizer of x [B, n_tok, C] fp16 (the fc1 -> act -> fc2-quantizer hand-over);
    B = 1, or B = 2 with the grid of a token shared by its two samples (rows = sample * n_tok + token)."""
    _req(x, torch.float16, "x")
    B, n_tok, Cc = x.shape
    Kp = pad128(Cc)
    dev = x.device
    rows = B * n_tok
    xq = torch.empty((rows, Kp), dtype=torch.int8, device=dev)
    sx = torch.empty(rows, dtype=torch.float32, device=dev)
    zx = torch.empty(rows, dtype=torch.int32, device=dev)
    R = torch.empty(rows, dtype=torch.int32, device=dev)
    if s is not None:
        _req(s, torch.float32, "s")
    s_rcp = smooth_rcp(s) if fast_div else None
    check(_L().vq_gelu_rowquant(_p(x), _p(s), _p(s_rcp), _p(xq), _p(sx), _p(zx), _p(R), B, n_tok, Cc, Kp, n_bits,
                                _p(status), _stream()), "vq_gelu_rowquant")
    return QAct(xq, sx, zx, R, Cc, n_bits, None)


def ln_modulate_rowquant(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, eps: float = 1e-6,
                         smooth: Sequence[Optional[torch.Tensor]] = (None,), n_bits: int = 8,
                         status: Optional[torch.Tensor] = None, want_xm: bool = False, fast_div: bool = True):
    """LN(no affine) + (1+scale)*.+shift + per-token quant; one QAct per entry of ``smooth``."""
    _req(x, torch.float16, "x")
    B, n_tok, Cc = x.shape
    _req(shift, torch.float32, "shift")
    _req(scale, torch.float32, "scale")
    assert shift.numel() == B * Cc and scale.numel() == B * Cc
    n_out = len(smooth)
    Kp = pad128(Cc)
    rows = B * n_tok
    dev = x.device
    outs: List[QAct] = []
    for _ in range(n_out):
        outs.append(QAct(torch.empty((rows, Kp), dtype=torch.int8, device=dev),
                         torch.empty(rows, dtype=torch.float32, device=dev),
                         torch.empty(rows, dtype=torch.int32, device=dev),
                         torch.empty(rows, dtype=torch.int32, device=dev), Cc, n_bits))
    arr = C.c_void_p * 3

    def mk(vals):
        vals = list(vals) + [None] * (3 - len(vals))
        return arr(*[C.c_void_p(v) if v is not None else None for v in vals])

    for sm in smooth:
        if sm is not None:
            _req(sm, torch.float32, "smooth")
    s_arr = mk([_p(sm) for sm in smooth])
    rcps = [smooth_rcp(sm) if fast_div else None for sm in smooth]     # kept alive until the launch is enqueued
    r_arr = mk([_p(r) for r in rcps])
    xq_arr = mk([o.xq.data_ptr() for o in outs])
    sx_arr = mk([o.sx.data_ptr() for o in outs])
    zx_arr = mk([o.zx.data_ptr() for o in outs])
    R_arr = mk([o.R.data_ptr() for o in outs])
    xm = torch.empty_like(x) if want_xm else None
    check(_L().vq_ln_modulate_rowquant(_p(x), _p(shift), _p(scale), float(eps), n_out,
                                       C.cast(s_arr, C.c_void_p), C.cast(r_arr, C.c_void_p), C.cast(xq_arr, C.c_void_p),
                                       C.cast(sx_arr, C.c_void_p), C.cast(zx_arr, C.c_void_p),
                                       C.cast(R_arr, C.c_void_p), _p(xm), B, n_tok, Cc, Kp, n_bits,
                                       _p(status), _stream()), "vq_ln_modulate_rowquant")
    return (outs, xm) if want_xm else outs


def fakequant_act(x: torch.Tensor, n_bits: int = 8, delta: Optional[torch.Tensor] = None,
                  zp: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
                  want_codes: bool = False):
    """Exact fake-quant of x [B, n_tok, C] fp16.  delta/zp None -> per-token dynamic."""
    _req(x, torch.float16, "x")
    B, n_tok, Cc = x.shape
    dev = x.device
    out = torch.empty_like(x)
    codes = torch.empty((B, n_tok, Cc), dtype=torch.uint8, device=dev) if want_codes else None
    if delta is None:
        d_out = torch.empty(n_tok, dtype=torch.float32, device=dev)
        z_out = torch.empty(n_tok, dtype=torch.float32, device=dev)
        scratch = torch.empty(1, dtype=torch.float32, device=dev)
        check(_L().vq_fakequant_act(_p(x), _p(out), _p(codes), _p(d_out), _p(z_out), None, None, 0, B, n_tok, Cc,
                                    n_bits, 0, _p(scratch), _p(status), _stream()), "vq_fakequant_act")
        return out, codes, d_out, z_out
    d = _req(delta.reshape(-1).contiguous(), torch.float32, "delta")
    z = _req(zp.reshape(-1).contiguous(), torch.float32, "zp")
    assert d.numel() == z.numel() and d.numel() in (1, n_tok)
    check(_L().vq_fakequant_act(_p(x), _p(out), _p(codes), None, None, _p(d), _p(z), d.numel(), B, n_tok, Cc,
                                n_bits, 1, None, _p(status), _stream()), "vq_fakequant_act")
    return out, codes, d, z


def epsfill_fixup(flag: torch.Tensor, x: torch.Tensor, s: Optional[torch.Tensor], wdq: torch.Tensor,
                  bias: Optional[torch.Tensor], out: torch.Tensor, n_bits: int = 8) -> torch.Tensor:
    """In-place global eps-fill fix-up of ``out`` [n_batch, L, N] fp16 = integer-route Linear(s) of the shared input x
    [L, C] fp16: a no-op kernel while ``flag`` (the quantizer's private status word) is clear, the reference's fp16-mode
    result with every token on the 1e-6 grid when it is set (base_quantizer.py:219-223).  wdq [n_batch, N, C] fp16
    dequantized weights, bias [n_batch, N] fp16 or None."""
    _req(x, torch.float16, "x"); _req(wdq, torch.float16, "wdq"); _req(out, torch.float16, "out")
    nb, N, C = wdq.shape
    L = x.shape[-2]
    assert x.shape[-1] == C and x.numel() == L * C and out.numel() == nb * L * N
    assert x.is_contiguous() and wdq.is_contiguous() and out.is_contiguous()
    if s is not None:
        _req(s, torch.float32, "s")
    if bias is not None:
        _req(bias, torch.float16, "bias")
        assert bias.numel() == nb * N and bias.is_contiguous()
    check(_L().vq_epsfill_fixup(_p(flag), _p(x), _p(s), _p(wdq), _p(bias), _p(out), nb, L, C, N, n_bits, _stream()),
          "vq_epsfill_fixup")
    return out


# --------------------------------------------------------------------------- weights
def weight_minmax(W: torch.Tensor, n_bits: int, s: Optional[torch.Tensor] = None, force_eps: bool = False,
                  status: Optional[torch.Tensor] = None):
    """Per-out-channel min-max (delta, zp) of W*s, W [N,K] fp16."""
    _req(W, torch.float16, "W")
    N, K = W.shape
    delta = torch.empty(N, dtype=torch.float32, device=W.device)
    zp = torch.empty(N, dtype=torch.float32, device=W.device)
    if s is not None:
        _req(s, torch.float32, "s")
    check(_L().vq_weight_minmax(_p(W), _p(s), _p(delta), _p(zp), N, K, n_bits, int(force_eps), _p(status),
                                _stream()), "vq_weight_minmax")
    return delta, zp


def packed_shapes(N: int, K: int, n_bits: int):
    """(shape, dtype) of the four tensors of a PackedWeight [N, K] at ``n_bits``: codes, sw, zw, cs."""
    Kp = pad128(K)
    wq = ((N, Kp // 2), torch.uint8) if n_bits <= 4 else ((N, Kp), torch.int8)
    return [wq, ((N,), torch.float32), ((N,), torch.int32), ((N,), torch.int32)]


def pack_weight(W: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, n_bits: int,
                s: Optional[torch.Tensor] = None, out: Optional[Sequence[torch.Tensor]] = None) -> PackedWeight:
    """Quantize W*s on the (delta, zp) grid and pack for the int8 MFMA GEMM.  ``out`` = (wq, sw, zw, cs) pre-allocated
    with :func:`packed_shapes` (e.g. views of one broadcast arena, shard.py) instead of fresh tensors."""
    _req(W, torch.float16, "W")
    N, K = W.shape
    Kp = pad128(K)
    dev = W.device
    d = _req(delta.reshape(-1).contiguous(), torch.float32, "delta")
    z = _req(zp.reshape(-1).contiguous(), torch.float32, "zp")
    assert d.numel() == N and z.numel() == N
    if out is not None:
        wq, sw, zw, cs = out
        for t_, (shape, dt) in zip(out, packed_shapes(N, K, n_bits)):
            if tuple(t_.shape) != tuple(shape) or t_.dtype != dt or not t_.is_contiguous() or t_.device != dev:
                raise VQError("pack_weight: out tensors must match packed_shapes() on the weight's device")
    else:
        (wqs, wqd), _, _, _ = packed_shapes(N, K, n_bits)
        wq = torch.empty(wqs, dtype=wqd, device=dev)
        sw = torch.empty(N, dtype=torch.float32, device=dev)
        zw = torch.empty(N, dtype=torch.int32, device=dev)
        cs = torch.empty(N, dtype=torch.int32, device=dev)
    if s is not None:
        _req(s, torch.float32, "s")
    check(_L().vq_pack_weight(_p(W), _p(s), _p(d), _p(z), _p(wq), _p(sw), _p(zw), _p(cs), N, K, Kp, n_bits,
                              _stream()), "vq_pack_weight")
    return PackedWeight(wq, sw, zw, cs, N, K, Kp, n_bits)


# --------------------------------------------------------------------------- GEMM
# When set to a list, every GEMM launch is bracketed by two events recorded on the launch stream and
# (start, end, int8_ops, algorithmic_bytes) is appended - bench.py's live roofline measurement.
GEMM_TIMING = None
# kernel used when the caller does not choose: -1 = the library's own choice per shape (VQ_GEMM_DEFAULT);
# 11 pins the full-line (128 B of k per row) double-buffered LDS-DMA ring, 256x288 tile (csrc/gemm_wide.h);
# nibble-packed (<= 4 bit) weights take the same kernels with 64-byte packed rows
DEFAULT_GEMM_VARIANT = -1


def gemm_i8(a: QAct, w: PackedWeight, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            epilogue: int = EPI_NONE, resid: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
            rows_per_gate: int = 0, variant: int = -1) -> torch.Tensor:
    """out[M, N] fp16 = dequant(int8 MFMA(a, w)) + bias, with the fused epilogue."""
    if a.K != w.K or a.Kp != w.Kp:
        raise VQError("K mismatch: activation %d/%d weight %d/%d" % (a.K, a.Kp, w.K, w.Kp))
    M, N = a.rows, w.N
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    else:
        if not out.is_cuda or out.dtype != torch.float16 or out.dim() != 2 or out.stride(1) != 1:
            raise VQError("out must be a GPU fp16 2-D tensor with unit column stride")
        assert out.shape[0] == M and out.shape[1] >= N
    ldo = out.stride(0)
    if bias is not None:
        _req(bias, torch.float32, "bias")
    if resid is not None:
        if not resid.is_cuda or resid.dtype != torch.float16 or resid.dim() != 2 or resid.stride(1) != 1:
            raise VQError("resid must be a GPU fp16 2-D tensor with unit column stride")
        assert resid.stride(0) == ldo and resid.shape[0] == M
    if gate is not None:
        _req(gate, torch.float32, "gate")
    if variant < 0:
        variant = DEFAULT_GEMM_VARIANT
    timing = GEMM_TIMING
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().vq_gemm_i8(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs),
                          _p(bias), _p(out), ldo, _p(resid), _p(gate), rows_per_gate, M, N, a.K, a.Kp,
                          w.n_bits, epilogue, variant, _stream()), "vq_gemm_i8")
    if timing is not None:
        e1.record()
        wbytes = N * a.K // 2 if w.n_bits <= 4 else N * a.K
        timing.append((e0, e1, 2.0 * M * N * a.K, M * a.K + wbytes + 2 * M * N * (2 if resid is not None else 1)))
    return out


def gemm_i8_stamped(a: QAct, w: PackedWeight, bias: Optional[torch.Tensor] = None):
    """Diagnostics: ONE launch of the default 8-bit GEMM (256 x 288 tile, plain epilogue) with per-wave stamps.
    -> (out [M, N] fp16, stamps [tiles, 8, 10] int64: 0-6 shader cycles per phase boundary, 7 / 8 the chip's 100 MHz wall
    clock at entry / exit).  `shader_clock_ghz(stamps)` turns them into the clock the kernel ran at."""
    if a.K != w.K or a.Kp != w.Kp or w.n_bits != 8:
        raise VQError("gemm_i8_stamped: 8-bit weights of the activation's K")
    M, N = a.rows, w.N
    if not a.xq.is_cuda:
        raise VQError("gemm_i8_stamped needs GPU tensors")
    out = torch.empty((M, N), dtype=torch.float16, device=a.xq.device)
    tiles = ((M + 255) // 256) * ((N + 287) // 288)
    stamps = torch.zeros((tiles, 8, 10), dtype=torch.int64, device=a.xq.device)
    check(_L().vq_gemm_i8_stamped(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs), _p(bias),
                                  _p(out), out.stride(0), M, N, a.K, a.Kp, _p(stamps), stamps.numel(), _stream()),
          "vq_gemm_i8_stamped")
    return out, stamps


def shader_clock_ghz(stamps: torch.Tensor) -> dict:
    """Median over waves of (shader cycles) / (10 ns wall-clock ticks) of a stamped GEMM launch, plus the tile phases in
    shader cycles (means over waves): what bench.py records beside its rates."""
    s = stamps.detach().cpu().double()
    cyc = s[:, :, 6] - s[:, :, 0]
    ticks = (s[:, :, 8] - s[:, :, 7]).clamp(min=1)
    ghz = float((cyc / ticks).median()) / 10.0
    ph = (s[:, :, 1:7] - s[:, :, 0:6]).mean(dim=(0, 1))
    names = ["prologue", "main_loop", "barrier_params", "dequant_slab", "store_issue", "store_drain"]
    span_us = float(s[:, :, 8].max() - s[:, :, 7].min()) / 100.0
    return {"ghz": ghz, "tile_cycles": float(cyc.mean()), "launch_span_us": span_us,
            "phase_cycles": {n: float(v) for n, v in zip(names, ph)}}


# --------------------------------------------------------------------------- attention
@dataclass
class PackedStack:
    """nbatch PackedWeights of identical shape stacked for vq_gemm_i8_batched (+ their fp32 biases)."""
    wq: torch.Tensor      # [nbatch, N, Kp] int8
    sw: torch.Tensor      # [nbatch, N]
    zw: torch.Tensor
    cs: torch.Tensor
    bias: Optional[torch.Tensor]
    nbatch: int
    N: int
    K: int
    Kp: int
    n_bits: int


def stack_packed(pws: Sequence[PackedWeight], biases: Sequence[Optional[torch.Tensor]]) -> PackedStack:
    p0 = pws[0]
    assert all(p.N == p0.N and p.K == p0.K and p.Kp == p0.Kp and p.n_bits == p0.n_bits for p in pws)
    if p0.n_bits <= 4:
        raise VQError("stack_packed: 8-bit (int8-stored) weights only")
    has_b = biases[0] is not None
    return PackedStack(torch.stack([p.wq for p in pws]).contiguous(), torch.stack([p.sw for p in pws]).contiguous(),
                       torch.stack([p.zw for p in pws]).contiguous(), torch.stack([p.cs for p in pws]).contiguous(),
                       torch.stack([b.float() for b in biases]).contiguous() if has_b else None,
                       len(pws), p0.N, p0.K, p0.Kp, p0.n_bits)


def gemm_i8_batched(a: QAct, w: PackedStack, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] [M, N] fp16 for every stacked weight set b, all from the same quantized activation ``a``."""
    if a.K != w.K or a.Kp != w.Kp:
        raise VQError("K mismatch: activation %d/%d weight %d/%d" % (a.K, a.Kp, w.K, w.Kp))
    M = a.rows
    if out is None:
        out = torch.empty((w.nbatch, M, w.N), dtype=torch.float16, device=a.xq.device)
    check(_L().vq_gemm_i8_batched(_p(a.xq), _p(a.sx), _p(a.zx), _p(a.R), _p(w.wq), _p(w.sw), _p(w.zw), _p(w.cs),
                                  _p(w.bias), _p(out), w.nbatch, M, w.N, a.K, a.Kp, w.n_bits, _stream()),
          "vq_gemm_i8_batched")
    return out


def _ptr_array(vals, n=3):
    arr = (C.c_void_p * n)(*[C.c_void_p(v) if v is not None else None for v in list(vals) + [None] * (n - len(vals))])
    return arr


def gemm_i8_grouped(acts: Sequence[QAct], ws: Sequence[PackedWeight], biases: Optional[Sequence] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """2 or 3 Linears of one shape in one launch: out[:, g*N:(g+1)*N] = Linear_g(acts[g]) (q | k | v blocks)."""
    G = len(acts)
    assert 1 <= G <= 3 and len(ws) == G
    a0, w0 = acts[0], ws[0]
    for a, w in zip(acts, ws):
        if (a.K, a.Kp, a.rows) != (a0.K, a0.Kp, a0.rows) or (w.N, w.K, w.Kp, w.n_bits) != (w0.N, w0.K, w0.Kp, w0.n_bits) \
                or a.K != w.K or a.Kp != w.Kp:
            raise VQError("grouped GEMM needs one shape for every group")
    M, N = a0.rows, w0.N
    if out is None:
        out = torch.empty((M, G * N), dtype=torch.float16, device=a0.xq.device)
    assert out.dtype == torch.float16 and out.stride(-1) == 1 and out.shape[-1] >= G * N
    bs = [None] * G if biases is None else [None if b is None else _req(b, torch.float32, "bias") for b in biases]
    keep = [_ptr_array([_p(getattr(a, f)) for a in acts]) for f in ("xq", "sx", "zx", "R")] + \
           [_ptr_array([_p(getattr(w, f)) for w in ws]) for f in ("wq", "sw", "zw", "cs")] + [_ptr_array([_p(b) for b in bs])]
    check(_L().vq_gemm_i8_grouped(G, *[C.cast(k, C.c_void_p) for k in keep], _p(out), out.stride(0), M, N, a0.K, a0.Kp,
                                  w0.n_bits, _stream()), "vq_gemm_i8_grouped")
    return out


def attn_fwd(q, k, v, o, n_seq, Lq, Lk, H, D, q_seq_stride, q_tok_stride, kv_seq_stride, kv_tok_stride,
             o_seq_stride, o_tok_stride, kv_off: Optional[torch.Tensor] = None, scale: Optional[float] = None):
    """Flash attention over strided fp16 views (pointers are the tensors' data_ptr())."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        if not t.is_cuda or t.dtype != torch.float16:
            raise VQError("%s must be a GPU fp16 tensor" % n)
    if kv_off is not None:
        _req(kv_off, torch.int32, "kv_off")
    scale = float(D) ** -0.5 if scale is None else float(scale)
    check(_L().vq_attn_fwd(_p(q), _p(k), _p(v), _p(o), n_seq, Lq, Lk, H, D, q_seq_stride, q_tok_stride,
                           kv_seq_stride, kv_tok_stride, o_seq_stride, o_tok_stride, _p(kv_off), scale, _stream()),
          "vq_attn_fwd")
    return o


def attn_temporal(q, k, v, o, B, T, S, H, D, ld_in, ld_out, scale: Optional[float] = None):
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        if not t.is_cuda or t.dtype != torch.float16:
            raise VQError("%s must be a GPU fp16 tensor" % n)
    scale = float(D) ** -0.5 if scale is None else float(scale)
    check(_L().vq_attn_temporal(_p(q), _p(k), _p(v), _p(o), B, T, S, H, D, ld_in, ld_out, scale, _stream()),
          "vq_attn_temporal")
    return o


def attn_temporal_rowquant(q, k, v, B, T, S, H, D, ld_in, scale: Optional[float] = None,
                           status: Optional[torch.Tensor] = None, o: Optional[torch.Tensor] = None,
                           s: Optional[torch.Tensor] = None) -> Optional[QAct]:
    """Temporal attention + the consuming Linear's per-token 8-bit dynamic quantizer in one kernel (B == 1 per
    forward: per-token scales are shared over the batch otherwise).  Returns what
    ``rowquant(attn_temporal(...).view(1, T*S, H*D), s=s)`` returns, bit for bit.  ``s``: the consuming Linear's
    smoothing vector; None is returned (caller runs the two kernels) when its reciprocal form is not available."""
    s_rcp = None
    if s is not None:
        s_rcp = smooth_rcp(s)
        if s_rcp is None:
            return None
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float16:
            raise VQError("%s must be a GPU fp16 tensor" % n)
    if B != 1:
        raise VQError("attn_temporal_rowquant: per-token scales are shared over the batch; B must be 1")
    Cc = H * D
    Kp = pad128(Cc)
    rows = B * T * S
    dev = q.device
    xq = torch.empty((rows, Kp), dtype=torch.int8, device=dev)
    sx = torch.empty(rows, dtype=torch.float32, device=dev)
    zx = torch.empty(rows, dtype=torch.int32, device=dev)
    R = torch.empty(rows, dtype=torch.int32, device=dev)
    scale = float(D) ** -0.5 if scale is None else float(scale)
    if o is not None:
        _req(o, torch.float16, "o")
        assert o.shape == (rows, Cc)
    check(_L().vq_attn_temporal_rowquant(_p(q), _p(k), _p(v), _p(s), _p(s_rcp), _p(xq), _p(sx), _p(zx), _p(R), _p(status),
                                         _p(o), B, T, S, H, D, ld_in, Kp, scale, _stream()), "vq_attn_temporal_rowquant")
    return QAct(xq, sx, zx, R, Cc, 8)


# --------------------------------------------------------------------------- misc
def adaln_table(table: torch.Tensor, t0: torch.Tensor) -> torch.Tensor:
    """mod[J, B, C] fp32 = table[J, C] + t0[B, J*C]  (fp16 inputs); mod[j] is a contiguous [B, C]."""
    _req(table, torch.float16, "table")
    _req(t0, torch.float16, "t0")
    J, Cc = table.shape
    B = t0.shape[0]
    assert t0.numel() == B * J * Cc
    mod = torch.empty((J, B, Cc), dtype=torch.float32, device=table.device)
    check(_L().vq_adaln_table(_p(table), _p(t0), _p(mod), B, J, Cc, _stream()), "vq_adaln_table")
    return mod


ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2


def linear_f16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act_in: int = ACT_NONE,
               act_out: int = ACT_NONE) -> torch.Tensor:
    """act_out(act_in(x) @ w.T + bias) for the FP edge Linears (embedders, t_block, final layer, patch embedding as a
    matmul): x [..., K] fp16 (last dim contiguous), w [N, K] fp16, bias [N] fp16; fp32 accumulation, fp16 result."""
    _req(x, torch.float16, "x")
    _req(w, torch.float16, "w")
    K = x.shape[-1]
    N = w.shape[0]
    assert w.dim() == 2 and w.shape[1] == K and w.stride(1) == 1
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    if bias is not None:
        _req(bias, torch.float16, "bias")
        assert bias.numel() == N and bias.is_contiguous()
    M = x2.shape[0]
    out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    check(_L().vq_linear_f16(_p(x2), _p(w), _p(bias), _p(out), M, N, K, x2.stride(0), w.stride(0), N, act_in, act_out,
                             _stream()), "vq_linear_f16")
    return out.reshape(*x.shape[:-1], N)


def cfg_ddim_step(cond: torch.Tensor, uncond: torch.Tensor, x: torch.Tensor, cfg: float, one_plus_k: float,
                  A: float, Bc: float, abar_prev: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(cond, torch.float32, "cond")
    _req(uncond, torch.float32, "uncond")
    _req(x, torch.float32, "x")
    n, Cc = x.shape[0], x.shape[1]
    inner = x[0, 0].numel()
    assert cond.shape[0] == n and cond.shape[1] == 2 * Cc and cond.shape == uncond.shape
    if out is None:
        out = torch.empty_like(x)
    check(_L().vq_cfg_ddim_step(_p(cond), _p(uncond), _p(x), _p(out), n, Cc, inner, float(cfg), float(one_plus_k),
                                float(A), float(Bc), float(abar_prev), _stream()), "vq_cfg_ddim_step")
    return out
