"""Oracle: the reference's fake-quant arithmetic, restated (torch CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows (paths relative to /root/reference).

All arithmetic is IEEE fp32 with torch.round (= round-half-to-even), exactly
what the reference executes when run on CPU in fp32.  The HIP path stores
activations in fp16 between kernels but does its quantizer arithmetic in fp32
on those fp16-representable values, so integer codes are compared bit-exactly
and dequantized / GEMM outputs within the tolerance stated in each test.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

EPS = 1.0e-6  # qdiff/quantizer/base_quantizer.py:219


def minmax_params(x2d: torch.Tensor, n_bits: int) -> Tuple[torch.Tensor, torch.Tensor, bool]:
    """Asymmetric min-max (delta, zero_point) per row of ``x2d`` [G, E].

    qdiff/quantizer/base_quantizer.py:191-228 (``sym: False``, ``scale_method:
    min_max``): min clamped to <=0, max to >=0, delta=(max-min)/(2^b-1); if ANY
    group's delta < 1e-6 then EVERY delta := 1e-6 (:220-222); zp =
    round(-min/delta).  Returns (delta[G], zp[G], eps_filled).
    """
    x2d = x2d.to(torch.float32)
    x_min = x2d.min(dim=-1)[0].clone()
    x_min[x_min > 0] = 0.0
    x_max = x2d.max(dim=-1)[0].clone()
    x_max[x_max < 0] = 0.0
    n_levels = 2 ** n_bits
    delta = (x_max - x_min) / (n_levels - 1)
    eps_filled = bool(delta.min() < EPS)
    if eps_filled:
        delta = torch.full_like(delta, EPS)
    zp = torch.round(-x_min / delta)
    return delta, zp, eps_filled


def quant_codes(x: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, n_bits: int) -> torch.Tensor:
    """Integer codes clamp(round(x/delta)+zp, 0, 2^b-1) as float.

    qdiff/quantizer/base_quantizer.py:134-140 (and dynamic_quantizer.py:36-41).
    """
    n_levels = 2 ** n_bits
    x_int = torch.round(x.to(torch.float32) / delta) + zp
    return torch.clamp(x_int, 0, n_levels - 1)


def dequant(codes: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor) -> torch.Tensor:
    """(x_q - zp) * delta.  qdiff/quantizer/base_quantizer.py:143."""
    return (codes - zp) * delta


# ----------------------------------------------------------------------------
# weight quantizer  (per out-channel, channel_dim 0)
# ----------------------------------------------------------------------------
def weight_params(W: torch.Tensor, n_bits: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-out-channel (delta, zp), each [N,1].

    qdiff/quantizer/base_quantizer.py:168-172 (``per_group: channel``,
    ``channel_dim: 0``) + :191-228, reshaped as :272-276.
    """
    N = W.shape[0]
    delta, zp, _ = minmax_params(W.reshape(N, -1), n_bits)
    return delta.reshape(N, 1), zp.reshape(N, 1)


def weight_fakequant(W: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, n_bits: int):
    """Returns (codes, dequantized) for a weight [N,K] on a given (delta, zp) grid.

    qdiff/quantizer/base_quantizer.py:129-144.  NB the reference always uses
    ``delta_list[bit_idx_at_PTQ, 0]`` (:126) - the range-0, PTQ-bit grid - no
    matter which time-range or mixed-precision bit-width is active; callers
    pass that grid and the *current* ``n_bits`` (only the clamp changes).
    """
    codes = quant_codes(W, delta, zp, n_bits)
    return codes, dequant(codes, delta, zp)


# ----------------------------------------------------------------------------
# activation quantizers
# ----------------------------------------------------------------------------
def token_params(x: torch.Tensor, n_bits: int):
    """Per-token (delta, zp), each [1, n_tok, 1]; scales are SHARED over batch.

    qdiff/quantizer/base_quantizer.py:177-189 (x[B,n,C] -> permute(1,0,2) ->
    [n, B*C]) and :269-271.
    """
    assert x.dim() == 3
    B, n, C = x.shape
    x2d = x.permute(1, 0, 2).reshape(n, B * C)
    delta, zp, eps_filled = minmax_params(x2d, n_bits)
    return delta.reshape(1, n, 1), zp.reshape(1, n, 1), eps_filled


def dyn_act_quant(x: torch.Tensor, n_bits: int = 8):
    """Dynamic per-token activation fake-quant.

    qdiff/quantizer/dynamic_quantizer.py:16-45.  Returns (codes, dequant,
    delta[1,n,1], zp[1,n,1], eps_filled).
    """
    delta, zp, eps_filled = token_params(x, n_bits)
    codes = quant_codes(x, delta, zp, n_bits)
    return codes, dequant(codes, delta, zp), delta, zp, eps_filled


def tensor_params(x: torch.Tensor, n_bits: int):
    """Tensor-wise (delta, zp) scalars (``per_group: False``).

    qdiff/quantizer/base_quantizer.py:188-189,191-228.
    """
    delta, zp, _ = minmax_params(x.reshape(1, -1), n_bits)
    return delta.reshape(()), zp.reshape(())


def static_act_quant(x: torch.Tensor, delta: torch.Tensor, zp: torch.Tensor, n_bits: int = 8):
    """Static activation fake-quant with calibrated (delta, zp).

    qdiff/quantizer/base_quantizer.py:129-144 (``ActQuantizer`` after
    ``init_done``).
    """
    codes = quant_codes(x, delta, zp, n_bits)
    return codes, dequant(codes, delta, zp)


# ----------------------------------------------------------------------------
# smooth-quant channel balancing
# ----------------------------------------------------------------------------
def find_interval(timerange: Sequence[Sequence[int]], timestep_id) -> Optional[int]:
    """qdiff/models/quant_layer.py:15-19."""
    for index, interval in enumerate(timerange):
        if interval[0] <= timestep_id <= interval[1]:
            return index
    return None


def smooth_scale(act_scale_r: torch.Tensor, W: torch.Tensor, alpha: float) -> torch.Tensor:
    """s[1,K] = act_scale[r]^alpha / (max_rows |W|)^(1-alpha).

    qdiff/models/quant_layer.py:128-136 (zeros in act_scale -> 1e-5 first).
    """
    a = act_scale_r.to(torch.float32).clone()
    a[a == 0] = 1.0e-5
    return a.pow(alpha) / W.to(torch.float32).abs().max(dim=0)[0].pow(1 - alpha)


def act_scale_stat(x: torch.Tensor) -> torch.Tensor:
    """cur_act_scale = |x|.max(dim=-2).mean(dim=0, keepdim)  -> [1, K].

    qdiff/models/quant_layer.py:120 / :147.
    """
    return x.to(torch.float32).abs().max(dim=-2)[0].mean(dim=0, keepdim=True)


# ----------------------------------------------------------------------------
# a complete quantized Linear (the layer-level oracle)
# ----------------------------------------------------------------------------
def quant_linear(
    x: torch.Tensor,               # [B, n_tok, K] (already reshaped the way the layer's act-quant sees it)
    W: torch.Tensor,               # [N, K]
    bias: Optional[torch.Tensor],
    *,
    w_bits: int = 8,
    a_bits: int = 8,
    w_delta: Optional[torch.Tensor] = None,   # [N,1]; None -> min-max of the (smoothed) weight
    w_zp: Optional[torch.Tensor] = None,
    smooth: Optional[torch.Tensor] = None,    # [1,K] channel-wise scale s, or None
    act_mode: Optional[str] = "dynamic",      # 'dynamic' | 'static' | None (act quant off)
    a_delta: Optional[torch.Tensor] = None,
    a_zp: Optional[torch.Tensor] = None,
    weight_quant: bool = True,
    return_parts: bool = False,
):
    """One fake-quantized Linear, as QuantLayer.forward / the STDiT subclasses do.

    qdiff/models/quant_layer.py:99-225 and stdit_quant_layer.py:15-99 (the
    subclasses differ only in the reshape applied before the act quantizer,
    which the caller performs).  Order: x/=s ; act fake-quant ; W*=s ; weight
    fake-quant ; F.linear.
    """
    x = x.to(torch.float32)
    W = W.to(torch.float32)
    if smooth is not None:
        x = x / smooth
        W_eff = W * smooth
    else:
        W_eff = W
    parts = {}
    if act_mode == "dynamic":
        codes, x_hat, dx, zx, eps_filled = dyn_act_quant(x, a_bits)
        parts.update(x_codes=codes, x_delta=dx, x_zp=zx, eps_filled=eps_filled)
    elif act_mode == "static":
        codes, x_hat = static_act_quant(x, a_delta, a_zp, a_bits)
        parts.update(x_codes=codes, x_delta=a_delta, x_zp=a_zp)
    else:
        x_hat = x
    if weight_quant:
        if w_delta is None:
            w_delta, w_zp = weight_params(W_eff, w_bits)
        w_codes, W_hat = weight_fakequant(W_eff, w_delta, w_zp, w_bits)
        parts.update(w_codes=w_codes, w_delta=w_delta, w_zp=w_zp)
    else:
        W_hat = W_eff
    out = F.linear(x_hat, W_hat, None if bias is None else bias.to(torch.float32))
    if return_parts:
        return out, parts
    return out


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU(approximate='tanh')  (opensora/models/layers/blocks.py:27)."""
    return F.gelu(x, approximate="tanh")


def layernorm_noaffine(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """nn.LayerNorm(C, eps=1e-6, elementwise_affine=False)  (blocks.py:30-39, stdit.py:64)."""
    return F.layer_norm(x.to(torch.float32), (x.shape[-1],), None, None, eps)


def t2i_modulate(x, shift, scale):
    """opensora/models/layers/blocks.py:51."""
    return x * (1 + scale) + shift
