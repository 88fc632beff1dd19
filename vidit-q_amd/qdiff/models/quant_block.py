"""Attention operator facade.

The reference's quant_block.py holds diffusers-UNet-era wrappers (QuantAttention,
QuantAttnProcessor, QuantTransformerBlock ...) that are dead for opensora / pixart:
``get_specials`` returns [] (quant_block.py:653-655) and every QK / softmax activation
quantizer call is commented out (:617-632).  What remains observable is (i) the ``BaseQuantBlock``
type that QuantModel tests for and (ii) the name ``QuantAttention`` as the attention operator.
Here QuantAttention is that operator for DiT models: fp16 attention with fp32 softmax and no
QK/softmax quantization - the semantics the reference executes through flash-attn / xformers
(opensora/models/layers/blocks.py:169-187, 292-310) - running on the gfx950 kernels.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import ops


class BaseQuantBlock(nn.Module):
    def __init__(self, act_quant_params: dict = {}):
        super().__init__()
        self.use_weight_quant = False
        self.use_act_quant = False

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant


class TransformerBlock(nn.Module):
    pass


class QuantTransformerBlock(BaseQuantBlock):
    pass


def get_specials(model_type):
    return {}


class QuantAttention(nn.Module):
    """softmax(q k^T / sqrt(d)) v over strided fp16 views; self, temporal and varlen cross forms."""

    def __init__(self, num_heads: int, head_dim: int, act_quant_params: Optional[dict] = None):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, head_dim
        self.scale = head_dim ** -0.5
        # kept for API shape; the reference never enables them (quant_block.py:617-632)
        self.use_act_quant = False

    def spatial(self, qkv: torch.Tensor, n_seq: int, L: int, out: Optional[torch.Tensor] = None):
        """qkv [n_seq*L, 3*C] (q | k | v column blocks) -> [n_seq*L, C]."""
        C = self.num_heads * self.head_dim
        ld = qkv.stride(0)
        if out is None:
            out = torch.empty((qkv.shape[0], C), dtype=torch.float16, device=qkv.device)
        ops.attn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], out, n_seq, L, L, self.num_heads, self.head_dim,
                     L * ld, ld, L * ld, ld, L * out.stride(0), out.stride(0), scale=self.scale)
        return out

    def temporal(self, qkv: torch.Tensor, B: int, T: int, S: int, out: Optional[torch.Tensor] = None):
        """rows ordered (b, t, s); attention over t for every (b, s, head)."""
        C = self.num_heads * self.head_dim
        ld = qkv.stride(0)
        if out is None:
            out = torch.empty((qkv.shape[0], C), dtype=torch.float16, device=qkv.device)
        if T <= 16:
            ops.attn_temporal(qkv, qkv[:, C:], qkv[:, 2 * C:], out, B, T, S, self.num_heads, self.head_dim,
                              ld, out.stride(0), scale=self.scale)
        else:  # general T: the flash kernel over strided sequences, one launch per sample
            for b in range(B):
                base = qkv[b * T * S:]
                ob = out[b * T * S:]
                ops.attn_fwd(base, base[:, C:], base[:, 2 * C:], ob, S, T, T, self.num_heads, self.head_dim,
                             ld, S * ld, ld, S * ld, out.stride(0), S * out.stride(0), scale=self.scale)
        return out

    def temporal_quantized(self, qkv: torch.Tensor, B: int, T: int, S: int, status=None, s=None):
        """:meth:`temporal` + the 8-bit dynamic per-token quantizer of the next Linear (behind its smoothing vector
        ``s`` when it has one), one kernel; None when that kernel does not apply (then call :meth:`temporal` and the
        layer's own quantizer)."""
        C = self.num_heads * self.head_dim
        if B != 1 or T > 16 or self.num_heads > 16 or C % 16 != 0 or self.head_dim not in (16, 32, 64, 72):
            return None
        return ops.attn_temporal_rowquant(qkv, qkv[:, C:], qkv[:, 2 * C:], B, T, S, self.num_heads, self.head_dim,
                                          qkv.stride(0), scale=self.scale, status=status, s=s)

    def cross(self, q: torch.Tensor, kv: torch.Tensor, kv_off: torch.Tensor, B: int, Nq: int,
              out: Optional[torch.Tensor] = None):
        """q [B*Nq, C]; kv [sum_L, 2*C] (k | v); sample b sees kv rows [kv_off[b], kv_off[b+1])."""
        C = self.num_heads * self.head_dim
        if out is None:
            out = torch.empty_like(q)
        # Lk with kv_off = a bound on every sample's kv length (the longest prompt when the offsets came from
        # seq_offsets, else all rows of kv): short prompts take the register-resident kernel
        bound = int(getattr(kv_off, "max_len", 0)) or kv.shape[0]
        ops.attn_fwd(q, kv, kv[:, C:], out, B, Nq, bound, self.num_heads, self.head_dim, Nq * q.stride(0),
                     q.stride(0), 0, kv.stride(0), Nq * out.stride(0), out.stride(0), kv_off=kv_off, scale=self.scale)
        return out
