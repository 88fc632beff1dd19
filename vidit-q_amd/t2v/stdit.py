"""STDiT (OpenSORA v1.0) block and model forward on the gfx950 kernels.

Module / parameter names mirror t2v/opensora/models/stdit/stdit.py and
t2v/opensora/models/layers/blocks.py so that reference state-dicts, quant-param dicts
(``ckpt.pth``) and the name routing of QuantModel apply unchanged:
  blocks.{i}.attn.{q,k,v,proj}  attn_temp.{q,k,v,proj}  cross_attn.{q_linear,kv_linear,proj}
  mlp.{fc1,fc2}  scale_shift_table   x_embedder.proj  t_embedder.mlp.{0,2}  t_block.1
  y_embedder.y_proj.{fc1,fc2}  y_embedder.y_embedding  final_layer.{linear,scale_shift_table}

Two routes through a block:
* ``forward_fused`` (hot path; taken when every Linear of the block is on the integer route):
  fused LN+modulate+quant -> int8 GEMMs with fused epilogues -> HIP attention, rows kept in one
  canonical [B][T][S] order, residual stream updated in place.  ~20 kernels per block, no torch op.
* ``forward_layerwise``: the reference's data flow (stdit.py:96-133) module by module - used for
  FP inference and PTQ / calibration states; Linears are QuantLayers (or nn.Linear), attention is
  the HIP kernel, the elementwise glue is torch.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..qdiff.models.quant_block import QuantAttention
from ..qdiff.models.quant_layer import QuantLayer
from ..qdiff.quantizer.dynamic_quantizer import DynamicActQuantizer


def approx_gelu():
    return nn.GELU(approximate="tanh")


def t2i_modulate(x, shift, scale):
    return x * (1 + scale) + shift


# --------------------------------------------------------------------------- sincos embeddings
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_1d_sincos_pos_embed(embed_dim, length, scale=1.0):
    pos = np.arange(0, length)[..., None] / scale
    return get_1d_sincos_pos_embed_from_grid(embed_dim, pos)


def get_2d_sincos_pos_embed(embed_dim, grid_size, scale=1.0, base_size=None):
    if not isinstance(grid_size, tuple):
        grid_size = (grid_size, grid_size)
    grid_h = np.arange(grid_size[0], dtype=np.float32) / scale
    grid_w = np.arange(grid_size[1], dtype=np.float32) / scale
    if base_size is not None:
        grid_h *= base_size / grid_size[0]
        grid_w *= base_size / grid_size[1]
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size[1], grid_size[0]])
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


# --------------------------------------------------------------------------- small modules
def fp_edge_linear(layer, x, act_in=0, act_out=0):
    """One of the Linears the FP lists keep in floating point (embedders, t_block, final layer: SURVEY 8 row F4), through
    the HIP kernel vq_linear_f16 with its neighbouring activation fused - when it is a Linear in FP state on fp16 GPU
    tensors; returns None otherwise (quantized by a non-default FP list, CPU, fp32: the caller takes the module path)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float16):
        return None
    if isinstance(layer, nn.Linear):
        w, b = layer.weight, layer.bias
    else:
        if getattr(layer, "fwd_func", None) is not F.linear or getattr(layer, "weight_quant", False) or (
                getattr(layer, "act_quant", False) and not getattr(layer, "disable_act_quant", False)) or getattr(
                layer, "smooth_quant", False) or getattr(layer, "smooth_quant_running_stat", False):
            return None
        w, b = layer.org_weight, layer.org_bias
    if w.dtype != torch.float16 or w.numel() == 0 or w.shape[1] % 8 or w.shape[0] % 4 or (b is not None and b.dtype != torch.float16):
        return None
    # the kernel is forward-only and by-passes the module's __call__: anything autograd or a hook would observe keeps the
    # module path (round-4 advisor finding: the graph was silently cut and forward hooks skipped)
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        return None
    if getattr(layer, "_forward_hooks", None) or getattr(layer, "_forward_pre_hooks", None) or getattr(
            layer, "_backward_hooks", None) or getattr(layer, "_backward_pre_hooks", None):
        return None
    # hooks registered for EVERY module (torch.nn.modules.module.register_module_forward_hook and friends: calibration /
    # observer tooling) see the module path only - round-5 advisor
    gm = torch.nn.modules.module
    if getattr(gm, "_global_forward_hooks", None) or getattr(gm, "_global_forward_pre_hooks", None) or getattr(
            gm, "_global_backward_hooks", None) or getattr(gm, "_global_backward_pre_hooks", None):
        return None
    # the edge kernel is built for SKINNY problems (one 16 x 16 output tile per workgroup, no operand reuse): few rows, few
    # columns or a short contraction.  A full-size FP Linear (16384 x 4608 x 1152: the block MLP of an FP calibration
    # pass) stays with the vendor GEMM - it is not part of the quantized hot path.
    M = x.numel() // x.shape[-1]
    if not (M <= 640 or w.shape[0] <= 64 or w.shape[1] <= 32):
        return None
    return ops.linear_f16(x.contiguous(), w.detach(), None if b is None else b.detach(), act_in=act_in, act_out=act_out)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=approx_gelu, bias=True, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        # FP state (the caption embedder under every shipped FP list): two launches of the edge kernel, GELU fused
        if isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "tanh":
            h = fp_edge_linear(self.fc1, x, act_out=ops.ACT_GELU)
            if h is not None:
                o = fp_edge_linear(self.fc2, h)
                if o is not None:
                    return o
        return self.fc2(self.act(self.fc1(x)))


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=96):
        super().__init__()
        self.patch_size = patch_size
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        _, _, D, H, W = x.size()
        if W % self.patch_size[2] != 0:
            x = F.pad(x, (0, self.patch_size[2] - W % self.patch_size[2]))
        if H % self.patch_size[1] != 0:
            x = F.pad(x, (0, 0, 0, self.patch_size[1] - H % self.patch_size[1]))
        if D % self.patch_size[0] != 0:
            x = F.pad(x, (0, 0, 0, 0, 0, self.patch_size[0] - D % self.patch_size[0]))
        proj = self.proj
        fp = not (getattr(proj, "weight_quant", False) or getattr(proj, "act_quant", False) or
                  getattr(proj, "smooth_quant", False))
        if fp and x.is_cuda:
            # FP patch embedding (remain_fp.txt keeps it FP): kernel == stride, so the convolution is a
            # [tokens, Cin*pt*ph*pw] x [that, E] matmul.  MIOpen has no tuned kernel for this Conv3d and runs its
            # naive one (377 us per forward); the matmul form takes ~10 us.
            w_ = getattr(proj, "org_weight", None)
            w_ = proj.weight if w_ is None else w_
            b_ = getattr(proj, "org_bias", None) if hasattr(proj, "org_weight") else proj.bias
            B, Cin, D, H, W = x.shape
            pt, ph, pw = self.patch_size
            xp = x.reshape(B, Cin, D // pt, pt, H // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
            xp = xp.reshape(B * (D // pt) * (H // ph) * (W // pw), Cin * pt * ph * pw)
            w2 = w_.reshape(w_.shape[0], -1)
            if w_.dtype == torch.float16 and w2.shape[1] % 8 == 0 and w2.shape[0] % 4 == 0:
                out = ops.linear_f16(xp.to(w_.dtype).contiguous(), w2.detach(), None if b_ is None else b_.detach().to(w_.dtype))
            else:
                out = F.linear(xp.to(w_.dtype), w2, None if b_ is None else b_.to(w_.dtype))
            return out.reshape(B, -1, w_.shape[0])
        x = self.proj(x)
        return x.flatten(2).transpose(1, 2)  # BCTHW -> BNC


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    _freq_cache = {}

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        key = (str(t.device), dim, max_period)
        freqs = TimestepEmbedder._freq_cache.get(key)
        if freqs is None:   # host exp() like the reference (blocks.py:430-433), uploaded once (graph-capture safe)
            freqs = torch.exp(-np.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
            freqs = TimestepEmbedder._freq_cache[key] = freqs.to(t.device)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def forward(self, t, dtype):
        t_freq = self.timestep_embedding(t, self.frequency_embedding_size)
        if t_freq.dtype != dtype:
            t_freq = t_freq.to(dtype)
        h = fp_edge_linear(self.mlp[0], t_freq, act_out=ops.ACT_SILU)      # Linear, SiLU fused
        if h is not None:
            o = fp_edge_linear(self.mlp[2], h)
            if o is not None:
                return o
        return self.mlp(t_freq)


class CaptionEmbedder(nn.Module):
    def __init__(self, in_channels, hidden_size, uncond_prob, act_layer=approx_gelu, token_num=120):
        super().__init__()
        self.y_proj = Mlp(in_features=in_channels, hidden_features=hidden_size, out_features=hidden_size,
                          act_layer=act_layer)
        self.register_buffer("y_embedding", torch.randn(token_num, in_channels) / in_channels ** 0.5)
        self.uncond_prob = uncond_prob

    def forward(self, caption, train=False, force_drop_ids=None):
        return self.y_proj(caption)


class T2IFinalLayer(nn.Module):
    def __init__(self, hidden_size, num_patch, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, num_patch * out_channels, bias=True)
        self.scale_shift_table = nn.Parameter(torch.randn(2, hidden_size) / hidden_size ** 0.5)
        self.out_channels = out_channels

    def forward(self, x, t):
        if x.is_cuda and x.dtype == torch.float16 and x.dim() == 3 and self.scale_shift_table.dtype == torch.float16:
            # LayerNorm + modulate in ONE pass of the fused kernel (its modulated fp16 output; the quantized codes it
            # also produces are not used here) instead of a LayerNorm and two broadcast elementwise launches
            mod = ops.adaln_table(self.scale_shift_table.detach(), torch.cat([t, t], dim=1).to(torch.float16).contiguous())
            _, xm = ops.ln_modulate_rowquant(x.contiguous(), mod[0], mod[1], 1e-6, want_xm=True)
            o = fp_edge_linear(self.linear, xm)                              # FP under the t2v list; quantized in t2i
            return o if o is not None else self.linear(xm)
        shift, scale = (self.scale_shift_table[None] + t[:, None]).chunk(2, dim=1)
        x = t2i_modulate(self.norm_final(x), shift, scale)
        return self.linear(x)


class Attention(nn.Module):
    """Self-attention with separate q/k/v Linears (blocks.py:113-195, ``separate_qkv=True``)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        assert dim % num_heads == 0
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.separate_qkv = True
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.core = QuantAttention(num_heads, self.head_dim)

    def forward(self, x):
        """x [B', N', C] -> [B', N', C]: attention over N' within each of the B' sequences."""
        Bp, Np, C = x.shape
        q = self.q(x).reshape(Bp * Np, C)
        k = self.k(x).reshape(Bp * Np, C)
        v = self.v(x).reshape(Bp * Np, C)
        dt = q.dtype
        q, k, v = [t_.half().contiguous() for t_ in (q, k, v)]
        o = torch.empty_like(q)
        ops.attn_fwd(q, k, v, o, Bp, Np, Np, self.num_heads, self.head_dim, Np * C, C, Np * C, C, Np * C, C,
                     scale=self.scale)
        return self.proj(o.reshape(Bp, Np, C).to(dt))


_OFFSETS = {}


class _SmallCache:
    """Last-N cache for step-invariant tensors (prompt embedding, per-block cross-attention K/V).  Tensor members of a
    key are compared by (address, shape, strides, dtype, version); the entry keeps the key tensors alive, so their
    storage cannot be freed and its address recycled into a false hit while the entry exists."""

    def __init__(self, n=8):
        self.n, self.items = n, []

    @staticmethod
    def _sig(key):
        return tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, t._version)
                     if isinstance(t, torch.Tensor) else t for t in key)

    def get(self, key):
        sig = self._sig(key)
        for s_, _, val in self.items:
            if s_ == sig:
                return val
        return None

    def put(self, key, val):
        if len(self.items) >= self.n:
            self.items.pop(0)
        self.items.append((self._sig(key), key, val))
        return val

    def __len__(self):
        return len(self.items)


def seq_offsets(lens, device) -> torch.Tensor:
    """int32 prefix sums [0, l0, l0+l1, ...] of the prompt lengths on ``device``; cached per (lengths,
    device) so that a HIP-graph capture never sees the host-to-device copy."""
    key = (tuple(int(v) for v in lens), str(device))
    off = _OFFSETS.get(key)
    if off is None:
        if len(_OFFSETS) > 4096:
            _OFFSETS.clear()
        off = _OFFSETS[key] = torch.tensor(np.concatenate([[0], np.cumsum(key[0])]), dtype=torch.int32).to(device)
        off.max_len = max(key[0]) if key[0] else 0      # host-side bound for kernels specialised on short sequences
    return off


class MultiHeadCrossAttention(nn.Module):
    """blocks.py:277-310: q from image tokens, k/v from the (mask-selected) prompt tokens."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        assert d_model % num_heads == 0
        self.d_model, self.num_heads, self.head_dim = d_model, num_heads, d_model // num_heads
        self.q_linear = nn.Linear(d_model, d_model)
        self.kv_linear = nn.Linear(d_model, d_model * 2)
        self.proj = nn.Linear(d_model, d_model)
        self.core = QuantAttention(num_heads, self.head_dim)

    def forward(self, x, cond, mask=None):
        B, N, C = x.shape
        dt = x.dtype
        q = self.q_linear(x).reshape(B * N, C).half().contiguous()
        kv = self.kv_linear(cond).reshape(-1, 2 * C).half().contiguous()
        lens = mask if mask is not None else [kv.shape[0] // B] * B
        off = seq_offsets(lens, x.device)
        o = self.core.cross(q, kv, off, B, N)
        return self.proj(o.reshape(B, N, C).to(dt))


# --------------------------------------------------------------------------- block
# GELU inside fc2's quantizer pass (vq_gelu_rowquant: an HBM-bound kernel whose VALU is idle) instead of the fc1 GEMM
# epilogue (22.0 M VALU instructions for 5.3 M MFMAs there, matrix pipe 31 % busy: profiles/r02_gemm_pmc.md).  It is what
# the reference's fp16 mode computes - act() on the fp16-rounded fc1 output (quant_layer.py:211, blocks.py:27) - and
# neutral for the step (24.81 vs 24.76 steps/s A/B on one box, round 3; round 1 measured -1 %), while the fc1 launch
# gets ~20 % shorter.  On by default since round 3; VQ_GELU_QUANT=0 restores the GELU epilogue.
_ATTN_QUANT = __import__("os").environ.get("VQ_ATTN_QUANT", "1") != "0"   # attention kernels that also run the next quantizer
_GELU_QUANT = __import__('os').environ.get('VQ_GELU_QUANT', '1') != '0'


# The one activation of the block that no LayerNorm precedes is the prompt (cross_attn.kv_linear): a (near-)constant
# prompt token there triggers the reference's GLOBAL eps fill (base_quantizer.py:219-223), which the integer route does
# not express.  With this on (default) one small kernel follows the integer route (vq_epsfill_fixup): it reads the
# quantizer's private status word, returns at once when it is clear and otherwise overwrites the K/V rows with the
# reference's fp16-mode result on the 1e-6 grid: outputs then equal the reference's, nothing synchronises (round 3's
# first form - both routes computed with torch ops and a device-side select - cost PixArt-Sigma 5 % of its step).
EXACT_KV_EPS_FILL = __import__("os").environ.get("VQ_EXACT_KV_EPS_FILL", "1") != "0"


def prompt_kv_exact_fill(layer, y3, r, sv, pw):
    """kv_linear on prompt tokens y3 [1, L, C]: integer route, replaced on the device by the exact eps-fill result when
    the quantizer flagged a token with step < 1e-6."""
    aq = layer.act_quantizer
    if not (EXACT_KV_EPS_FILL and isinstance(aq, DynamicActQuantizer)):
        return ops.gemm_i8(layer.quantize_input(y3, sv), pw, bias=layer.bias_f32())
    st = ops.new_status(y3.device)
    qa = ops.rowquant(y3, n_bits=aq.n_bits, s=sv, status=st)
    kv = ops.gemm_i8(qa, pw, bias=layer.bias_f32())
    b = layer.bias
    ops.epsfill_fixup(st, y3.contiguous(), sv, layer.dequantized_weight_f16(r, sv)[None],
                      None if b is None else b.detach().half().contiguous(), kv, n_bits=aq.n_bits)
    return kv


class STDiTBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, d_s=None, d_t=None, mlp_ratio=4.0, **unused):
        super().__init__()
        self.hidden_size = hidden_size
        self.norm1 = nn.LayerNorm(hidden_size, eps=1e-6, elementwise_affine=False)
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True)
        self.cross_attn = MultiHeadCrossAttention(hidden_size, num_heads)
        self.norm2 = nn.LayerNorm(hidden_size, eps=1e-6, elementwise_affine=False)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio), act_layer=approx_gelu)
        self.scale_shift_table = nn.Parameter(torch.randn(6, hidden_size) / hidden_size ** 0.5)
        self._kv_cache = _SmallCache(8)
        self.cache_prompt = False                      # set by STDiT.set_prompt_cache
        self.d_s, self.d_t = d_s, d_t
        self.attn_temp = Attention(hidden_size, num_heads=num_heads, qkv_bias=True)
        self._fused_w = {}

    # ---- which route -----------------------------------------------------------------------
    def hot_layers(self) -> List[nn.Module]:
        return [self.attn.q, self.attn.k, self.attn.v, self.attn.proj,
                self.attn_temp.q, self.attn_temp.k, self.attn_temp.v, self.attn_temp.proj,
                self.cross_attn.q_linear, self.cross_attn.kv_linear, self.cross_attn.proj,
                self.mlp.fc1, self.mlp.fc2]

    def fused_ok(self) -> bool:
        for m in self.hot_layers():
            if not (isinstance(m, QuantLayer) and m.int_route_ok()):
                return False
            aq = m.act_quantizer
            if not isinstance(aq, DynamicActQuantizer) and aq.per_group:
                return False  # static per-token grids need the reference's [B, n_prompt, C] views
            if getattr(m, "smooth_quant_running_stat", False):
                return False  # a live act-scale statistic is QuantLayer.forward's job (host-visible state)
        for att in (self.attn, self.attn_temp):
            # q, k and v are quantized from ONE pass over their common input: that needs one activation bit-width
            # and one kind of quantizer (a mixed-precision YAML may set them apart: then the layerwise route runs)
            aqs = [l.act_quantizer for l in (att.q, att.k, att.v)]
            if len({(a.n_bits, isinstance(a, DynamicActQuantizer)) for a in aqs}) != 1:
                return False
        return True

    @staticmethod
    def _ln_quant(x3, shift, scale, layers, svs, status):
        """LayerNorm + modulate + the activation quantizer(s) of ``layers`` (which share the input).  Dynamic
        per-token quantizers: ONE fused kernel, one output per distinct smoothing vector.  Static calibrated grids
        (``dynamic: False``, the *_naive / *_ptqd plans): the fused kernel computes min-max grids only, so it emits the
        modulated fp16 activation and every layer quantizes it on ITS calibrated (delta, zero_point)
        (base_quantizer.py:129-144) - never a silently substituted dynamic grid."""
        l0 = layers[0]
        if all(isinstance(l.act_quantizer, DynamicActQuantizer) for l in layers):
            smooth = [None] if all(s is None for s in svs) else list(svs)
            return ops.ln_modulate_rowquant(x3, shift, scale, 1e-6, smooth=smooth, n_bits=l0.act_quantizer.n_bits,
                                            status=status)
        _, xm = ops.ln_modulate_rowquant(x3, shift, scale, 1e-6, smooth=[None], n_bits=8, want_xm=True)
        # one pass per layer: each has its own calibrated grid tensor (comparing them would be a host sync)
        return [l.quantize_input(xm, s) for l, s in zip(layers, svs)]

    def forward(self, x, y, t, mask=None, tpe=None):
        if x.is_cuda and x.dtype == torch.float16 and self.fused_ok():
            B, N, C = x.shape
            x2 = x.reshape(B * N, C).clone()
            lens = mask if mask is not None else [y.reshape(-1, C).shape[0] // B] * B
            off = mask if isinstance(mask, torch.Tensor) else torch.tensor(
                np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=x.device)
            self.forward_fused(x2, y.reshape(-1, C).contiguous(), t.contiguous(), off, tpe, B)
            return x2.reshape(B, N, C)
        return self.forward_layerwise(x, y, t, mask, tpe)

    # ---- reference data flow ---------------------------------------------------------------
    def forward_layerwise(self, x, y, t, mask=None, tpe=None):
        B, N, C = x.shape
        T, S = self.d_t, self.d_s
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + t.reshape(B, 6, -1)).chunk(6, dim=1)
        x_m = t2i_modulate(self.norm1(x), shift_msa, scale_msa)
        x_s = x_m.reshape(B, T, S, C).reshape(B * T, S, C)
        x_s = self.attn(x_s).reshape(B, T * S, C)
        x = x + gate_msa * x_s
        x_t = x.reshape(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
        if tpe is not None:
            x_t = x_t + tpe
        x_t = self.attn_temp(x_t.contiguous())
        x_t = x_t.reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(B, T * S, C)
        x = x + gate_msa * x_t
        x = x + self.cross_attn(x, y, mask)
        x = x + gate_mlp * self.mlp(t2i_modulate(self.norm2(x), shift_mlp, scale_mlp))
        return x

    # ---- hot path --------------------------------------------------------------------------
    def _qkv_weights(self, att: Attention, r: int, svec):
        """One [3C, K] packed weight when q/k/v share the activation codes and the bit-width."""
        layers = (att.q, att.k, att.v)
        pws = [l.packed_weight(r, s) for l, s in zip(layers, svec)]
        if any(s is not None for s in svec) or len({p.n_bits for p in pws}) != 1:
            return None, pws
        key = (id(att), r, pws[0].n_bits)
        ent = self._fused_w.get(key)
        if ent is not None and all(a is b for a, b in zip(ent[1], pws)):
            return ent[0], pws
        cat = ops.PackedWeight(torch.cat([p.wq for p in pws]), torch.cat([p.sw for p in pws]),
                               torch.cat([p.zw for p in pws]), torch.cat([p.cs for p in pws]),
                               sum(p.N for p in pws), pws[0].K, pws[0].Kp, pws[0].n_bits)
        bias = None
        if layers[0].bias is not None:
            bias = torch.cat([l.bias_f32() for l in layers])
        self._fused_w[key] = ((cat, bias), pws)
        return (cat, bias), pws

    def prompt_kv(self, y2):
        """kv_linear of the prompt tokens y2 [sum_L, C] on the integer route -> [sum_L, 2C] fp16.
        K/V of the prompt depend neither on the latent nor on the timestep: with ``cache_prompt`` they are computed
        once per (prompt tokens, packed weight) and re-used by every later step (the reference recomputes them per
        forward with identical results); without it they are still independent of the block's main chain, which
        is why STDiT.forward runs this on a side stream."""
        ca, C = self.cross_attn, self.hidden_size
        r, alpha = ca.kv_linear._range_and_alpha()
        sv = ca.kv_linear.smooth_vector(r, alpha)
        pw_kv = ca.kv_linear.packed_weight(r, sv)
        kkey = (y2, pw_kv.wq, ca.kv_linear.act_quantizer.n_bits)
        kv = self._kv_cache.get(kkey) if self.cache_prompt else None
        if kv is None:
            kv = prompt_kv_exact_fill(ca.kv_linear, y2.view(1, -1, C), r, sv, pw_kv)
            if self.cache_prompt:
                self._kv_cache.put(kkey, kv)
        return kv

    def forward_fused(self, x2, y2, t0, y_lens, tpe, B, kv_ready=None, mod=None):
        """In-place update of the residual stream x2 [B*T*S, C] fp16, rows ordered (b, t, s)."""
        T, S, C = self.d_t, self.d_s, self.hidden_size
        N = T * S
        M = B * N
        dev = x2.device
        a1, a2, ca = self.attn, self.attn_temp, self.cross_attn
        r, _ = a1.q._range_and_alpha()

        def svec(layer):
            rr, alpha = layer._range_and_alpha()
            return layer.smooth_vector(rr, alpha)

        if mod is None:
            mod = ops.adaln_table(self.scale_shift_table.detach(), t0.reshape(B, -1))   # [6, B, C] fp32
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [mod[j] for j in range(6)]
        x3 = x2.view(B, N, C)
        st = a1.q.status

        def qkv_proj(att, qas):
            svs = [svec(l) for l in (att.q, att.k, att.v)]
            fused, pws = self._qkv_weights(att, r, svs)
            qkv = torch.empty((M, 3 * C), dtype=torch.float16, device=dev)
            if fused is not None and len(qas) == 1:
                ops.gemm_i8(qas[0], fused[0], bias=fused[1], out=qkv)
            elif len({(p.N, p.K, p.Kp, p.n_bits) for p in pws}) == 1 and len({(a.K, a.Kp) for a in qas}) == 1:
                # three smoothing vectors -> three quantized copies of the input: one grouped launch (q | k | v blocks)
                ops.gemm_i8_grouped([qas[j if len(qas) > 1 else 0] for j in range(3)], pws,
                                    [l.bias_f32() for l in (att.q, att.k, att.v)], out=qkv)
            else:
                for j, l in enumerate((att.q, att.k, att.v)):
                    ops.gemm_i8(qas[j if len(qas) > 1 else 0], pws[j], bias=l.bias_f32(),
                                out=qkv[:, j * C:(j + 1) * C])
            return qkv

        # ---- spatial branch: x += gate_msa * proj(attn(LN-mod(x)))          (stdit.py:103-109)
        svs = [svec(l) for l in (a1.q, a1.k, a1.v)]
        qas = self._ln_quant(x3, shift_msa, scale_msa, (a1.q, a1.k, a1.v), svs, st)
        qkv = qkv_proj(a1, qas)
        att_o = a1.core.spatial(qkv, B * T, S)
        qa = a1.proj.quantize_input(att_o.view(B, N, C), svec(a1.proj))
        ops.gemm_i8(qa, a1.proj.packed_weight(r, svec(a1.proj)), bias=a1.proj.bias_f32(), out=x2,
                    epilogue=ops.EPI_GATE_RESID, resid=x2, gate=gate_msa, rows_per_gate=N)

        # ---- temporal branch on the un-modulated x (+tpe in block 0)          (stdit.py:112-118)
        svs = [svec(l) for l in (a2.q, a2.k, a2.v)]
        tpe2 = None if tpe is None else tpe.reshape(T, C).contiguous()
        if all(s is None for s in svs) and isinstance(a2.q.act_quantizer, DynamicActQuantizer):
            qas = [a2.q.quantize_input(x3, None, add_rows=tpe2, add_div=S)]
        elif tpe2 is None and B == 1 and all(s is not None for s in svs) and all(
                isinstance(l.act_quantizer, DynamicActQuantizer) for l in (a2.q, a2.k, a2.v)) and len(
                {l.act_quantizer.n_bits for l in (a2.q, a2.k, a2.v)}) == 1:
            qas = ops.rowquant_multi(x3, svs, n_bits=a2.q.act_quantizer.n_bits, status=a2.q.status)   # one launch
        else:
            qas = [l.quantize_input(x3, s, add_rows=tpe2, add_div=S) for l, s in zip((a2.q, a2.k, a2.v), svs)]
        qkv = qkv_proj(a2, qas)
        qa = None
        if _ATTN_QUANT and isinstance(a2.proj.act_quantizer, DynamicActQuantizer) and a2.proj.act_quantizer.n_bits == 8:
            # attention + proj's quantizer (behind proj's smoothing vector, if any) in one kernel
            qa = a2.core.temporal_quantized(qkv, B, T, S, status=a2.proj.status, s=svec(a2.proj))
        if qa is None:
            att_o = a2.core.temporal(qkv, B, T, S, out=att_o)
            qa = a2.proj.quantize_input(att_o.view(B, N, C), svec(a2.proj))
        ops.gemm_i8(qa, a2.proj.packed_weight(r, svec(a2.proj)), bias=a2.proj.bias_f32(), out=x2,
                    epilogue=ops.EPI_GATE_RESID, resid=x2, gate=gate_msa, rows_per_gate=N)

        # ---- cross attention: x += proj(attn(q(x), kv(y)))                    (stdit.py:121)
        qa = ca.q_linear.quantize_input(x3, svec(ca.q_linear))
        q = ops.gemm_i8(qa, ca.q_linear.packed_weight(r, svec(ca.q_linear)), bias=ca.q_linear.bias_f32())
        kv = kv_ready if kv_ready is not None else self.prompt_kv(y2)   # batched over the blocks by STDiT.forward
        att_o = ca.core.cross(q, kv, y_lens, B, N, out=att_o)
        qa = ca.proj.quantize_input(att_o.view(B, N, C), svec(ca.proj))
        ops.gemm_i8(qa, ca.proj.packed_weight(r, svec(ca.proj)), bias=ca.proj.bias_f32(), out=x2,
                    epilogue=ops.EPI_RESID, resid=x2)

        # ---- MLP: x += gate_mlp * fc2(gelu(fc1(LN-mod(x))))                   (stdit.py:124-128)
        fc1, fc2 = self.mlp.fc1, self.mlp.fc2
        qa = self._ln_quant(x3, shift_mlp, scale_mlp, (fc1,), [svec(fc1)], st)[0]
        one_pass = _GELU_QUANT and fc2.gelu_one_pass_ok(B, fc2.in_features, svec(fc2))
        h = ops.gemm_i8(qa, fc1.packed_weight(r, svec(fc1)), bias=fc1.bias_f32(),
                        epilogue=ops.EPI_NONE if one_pass else ops.EPI_GELU)
        qa = fc2.quantize_gelu_input(h.view(B, N, -1), svec(fc2)) if one_pass else fc2.quantize_input(h.view(B, N, -1), svec(fc2))
        ops.gemm_i8(qa, fc2.packed_weight(r, svec(fc2)), bias=fc2.bias_f32(), out=x2,
                    epilogue=ops.EPI_GATE_RESID, resid=x2, gate=gate_mlp, rows_per_gate=N)
        return x2


# --------------------------------------------------------------------------- model
class STDiT(nn.Module):
    def __init__(self, input_size=(1, 32, 32), in_channels=4, patch_size=(1, 2, 2), hidden_size=1152, depth=28,
                 num_heads=16, mlp_ratio=4.0, class_dropout_prob=0.1, pred_sigma=True, drop_path=0.0,
                 no_temporal_pos_emb=False, caption_channels=4096, model_max_length=120, dtype=torch.float32,
                 space_scale=1.0, time_scale=1.0, freeze=None, enable_flashattn=True, enable_layernorm_kernel=False,
                 enable_sequence_parallelism=False, separate_qkv=True):
        super().__init__()
        assert not enable_sequence_parallelism, "sequence parallelism is training-only in the reference"
        self.pred_sigma = pred_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if pred_sigma else in_channels
        self.hidden_size, self.patch_size, self.input_size = hidden_size, patch_size, input_size
        self.num_patches = int(np.prod([input_size[i] // patch_size[i] for i in range(3)]))
        self.num_temporal = input_size[0] // patch_size[0]
        self.num_spatial = self.num_patches // self.num_temporal
        self.num_heads, self.dtype, self.depth, self.mlp_ratio = num_heads, dtype, depth, mlp_ratio
        self.no_temporal_pos_emb = no_temporal_pos_emb
        self.space_scale, self.time_scale = space_scale, time_scale
        self.separate_qkv = True

        self.register_buffer("pos_embed", self.get_spatial_pos_embed())
        self.register_buffer("pos_embed_temporal", self.get_temporal_pos_embed())
        self.x_embedder = PatchEmbed3D(patch_size, in_channels, hidden_size)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.t_block = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))
        self.y_embedder = CaptionEmbedder(in_channels=caption_channels, hidden_size=hidden_size,
                                          uncond_prob=class_dropout_prob, act_layer=approx_gelu,
                                          token_num=model_max_length)
        self.blocks = nn.ModuleList([
            STDiTBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio, d_t=self.num_temporal, d_s=self.num_spatial)
            for _ in range(depth)])
        self.final_layer = T2IFinalLayer(hidden_size, int(np.prod(patch_size)), self.out_channels)
        self.initialize_weights()
        self.initialize_temporal()
        self._mask_cache = None
        self._prompt_cache = _SmallCache(8)
        self.cache_prompt = False
        self._kv_stack = None
        self._adaln_stack = None

    # ---- embeddings / init (stdit.py:367-442) ------------------------------------------------
    def get_spatial_pos_embed(self):
        g = self.input_size[1:]
        pe = get_2d_sincos_pos_embed(self.hidden_size, (g[0] // self.patch_size[1], g[1] // self.patch_size[2]),
                                     scale=self.space_scale)
        return torch.from_numpy(pe).float().unsqueeze(0).requires_grad_(False)

    def get_temporal_pos_embed(self):
        pe = get_1d_sincos_pos_embed(self.hidden_size, self.input_size[0] // self.patch_size[0], scale=self.time_scale)
        return torch.from_numpy(pe).float().unsqueeze(0).requires_grad_(False)

    def initialize_temporal(self):
        for block in self.blocks:
            nn.init.constant_(block.attn_temp.proj.weight, 0)
            nn.init.constant_(block.attn_temp.proj.bias, 0)

    def initialize_weights(self):
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                torch.nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        self.apply(_basic_init)
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        nn.init.normal_(self.t_block[1].weight, std=0.02)
        nn.init.normal_(self.y_embedder.y_proj.fc1.weight, std=0.02)
        nn.init.normal_(self.y_embedder.y_proj.fc2.weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.cross_attn.proj.weight, 0)
            nn.init.constant_(block.cross_attn.proj.bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    # ---- prompt-token selection (stdit.py:272-301) --------------------------------------------
    def _mask_select(self) -> bool:
        aq = getattr(self.final_layer.linear, "act_quantizer", None)
        if aq is not None and not isinstance(aq, DynamicActQuantizer) and aq.per_group == "token":
            return False
        return True

    def _select_prompt_tokens(self, y, mask, C):
        """Returns y [1, sum_L, C] and (host list y_lens, device int32 offsets).  The host copy of the
        lengths (one sync, as mask.sum().tolist() in stdit.py:286) is cached per mask tensor."""
        B = y.shape[0]
        if mask is None:
            lens = [y.shape[2]] * B
            return y.squeeze(1).reshape(1, -1, C), lens, None
        if self._mask_select():
            key = (mask.data_ptr(), mask._version, tuple(mask.shape), B)
            if self._mask_cache is None or self._mask_cache[0] != key:
                m = mask if mask.shape[0] == B else mask.repeat(B // mask.shape[0], 1)
                m = m.reshape(B, -1)
                idx = torch.nonzero(m.reshape(-1) != 0, as_tuple=False).reshape(-1)
                lens = [int(v) for v in m.sum(dim=1).tolist()]
                off = seq_offsets(lens, y.device)
                self._mask_cache = (key, idx, lens, off, mask)   # the mask is kept alive: its address cannot be recycled
            _, idx, lens, off = self._mask_cache[:4]
            ysel = y.squeeze(1).reshape(-1, C).index_select(0, idx).reshape(1, -1, C)
            return ysel, lens, off
        mask_ = mask if mask.shape[0] == B else mask.repeat([2, 1])
        lens = [y.shape[2]] * B
        y = y * mask_.unsqueeze(-1).unsqueeze(1)
        return y.squeeze(1).reshape(1, -1, C), lens, None

    def set_prompt_cache(self, on: bool = True):
        """Opt-in: compute the step-invariant prompt work (y_embedder, token selection, every block's
        cross-attention K/V) once per (text embedding, mask) instead of once per forward as the reference does.
        Results are bit-identical; OFF by default so that a step does exactly the reference's per-step work."""
        self.cache_prompt = bool(on)
        for b in self.blocks:
            b.cache_prompt = bool(on)
            b._kv_cache = _SmallCache(8)
        self._prompt_cache = _SmallCache(8)

    def _all_adaln(self, t0, B):
        """mod[6 i + j] = scale_shift_table_i[j] + t0[:, j]  for every block i: one launch on the stacked tables
        instead of one per block (t0 is the same for all blocks: stdit.py:304-306)."""
        tabs = [b.scale_shift_table for b in self.blocks]
        sig = tuple((t.data_ptr(), t._version) for t in tabs)
        st = self._adaln_stack
        if st is None or st[0] != sig:
            st = self._adaln_stack = (sig, torch.cat([t.detach() for t in tabs], dim=0).contiguous(), tabs)
        n = len(tabs)
        return ops.adaln_table(st[1], t0.reshape(B, -1).repeat(1, n).contiguous())   # [6 n, B, C] fp32

    def _all_prompt_kv(self, y2):
        """K/V of the prompt for EVERY block in two launches: the blocks' kv_linear layers see the same input y2, so
        (without smooth quant) the same quantized activation, and their [2C, C] weights are stacked into one batched
        GEMM (28 x 8 tiles = one round of workgroups) instead of 28 quantizer + 28 GEMM launches of 120 rows each.
        Returns a list of [sum_L, 2C] views, or None when the layers differ in a way that rules the batch out."""
        layers = [b.cross_attn.kv_linear for b in self.blocks]
        l0 = layers[0]
        ok = all(getattr(l, "int_route_ok", None) is not None and l.int_route_ok() and not getattr(l, "smooth_quant", False)
                 and isinstance(l.act_quantizer, DynamicActQuantizer) and l.act_quantizer.n_bits == l0.act_quantizer.n_bits
                 and l.weight_quantizer.n_bits == l0.weight_quantizer.n_bits for l in layers)
        if not ok or any(b.cache_prompt for b in self.blocks) or l0.weight_quantizer.n_bits <= 4:
            return None
        pws = [l.packed_weight(0, None) for l in layers]
        key = tuple(p.wq.data_ptr() for p in pws)
        st = self._kv_stack
        if st is None or st[0] != key:
            stack = ops.stack_packed(pws, [l.bias_f32() for l in layers])
            st = self._kv_stack = (key, stack, pws, None)      # pws kept alive: their addresses identify the stack
        y3 = y2.view(1, -1, self.hidden_size)
        if not EXACT_KV_EPS_FILL:
            out = ops.gemm_i8_batched(l0.quantize_input(y3, None), st[1])
            return [out[i] for i in range(len(layers))]
        # integer route for all blocks at once + the exact eps-fill route (one exact fake-quant of the shared input, one
        # batched fp16 GEMM on the stacked dequantized weights), selected on the device by the quantizer's status bit
        flag = ops.new_status(y2.device)
        qa = ops.rowquant(y3, n_bits=l0.act_quantizer.n_bits, status=flag)
        out = ops.gemm_i8_batched(qa, st[1])
        if len(st) < 4 or st[3] is None:
            wdq = torch.stack([l.dequantized_weight_f16(0, None) for l in layers])             # [nb, 2C, K] fp16
            bdq = torch.stack([l.bias.detach().half() for l in layers]).contiguous() if l0.bias is not None else None
            st = self._kv_stack = (st[0], st[1], st[2], (wdq, bdq))
        wdq, bdq = st[3]
        ops.epsfill_fixup(flag, y3.contiguous(), None, wdq, bdq, out, n_bits=l0.act_quantizer.n_bits)
        return [out[i] for i in range(len(layers))]

    def _prompt_tokens(self, y, mask, C):
        """y_embedder + prompt-token selection (stdit.py:266-301).  Neither depends on the latent or the
        timestep, so for one (text embedding, mask) pair the result is computed once and the SAME tensor is
        handed to every later forward - which is also what lets the blocks keep their cross-attention K/V.
        Only when the embedder runs in FP (the shipped FP lists keep it FP): a quantized embedder with
        time-range smoothing would depend on the step."""
        fcs = (self.y_embedder.y_proj.fc1, self.y_embedder.y_proj.fc2)
        fp = all(not getattr(m, "weight_quant", False) and not getattr(m, "act_quant", False) for m in fcs)
        key = None
        if self.cache_prompt and fp and not self.training:
            key = (y, mask, fcs[0].weight, fcs[1].weight)
            hit = self._prompt_cache.get(key)
            if hit is not None:
                return hit
        ye = self.y_embedder(y.to(self.dtype), self.training)
        out = self._select_prompt_tokens(ye, mask, C)
        if key is not None:
            self._prompt_cache.put(key, out)
        return out

    # ---- forward (stdit.py:238-341) -------------------------------------------------------------
    def forward(self, x, timestep, y, mask=None):
        x = x.to(self.dtype)
        timestep = timestep.to(self.dtype)
        C = self.hidden_size
        x = self.x_embedder(x)
        B = x.shape[0]
        x = x.reshape(B, self.num_temporal, self.num_spatial, C) + self.pos_embed
        x = x.reshape(B, self.num_patches, C).contiguous()
        t = self.t_embedder(timestep, dtype=x.dtype)
        t0 = fp_edge_linear(self.t_block[1], t, act_in=ops.ACT_SILU)        # SiLU, Linear in one launch
        if t0 is None:
            t0 = self.t_block(t)
        y, y_lens, off = self._prompt_tokens(y, mask, C)

        fused = x.is_cuda and x.dtype == torch.float16 and all(b.fused_ok() for b in self.blocks)
        if fused:
            x2 = x.reshape(B * self.num_patches, C)
            y2 = y.reshape(-1, C).contiguous()
            if off is None:
                off = seq_offsets(y_lens, x.device)
            t0c = t0.contiguous()
            kvs = self._all_prompt_kv(y2)              # one quantizer + ONE batched GEMM for all blocks (or None)
            mods = self._all_adaln(t0c, B)             # every block's AdaLN table in one launch
            for i, block in enumerate(self.blocks):
                block.forward_fused(x2, y2, t0c, off, self.pos_embed_temporal if i == 0 else None, B,
                                    kv_ready=None if kvs is None else kvs[i], mod=mods[6 * i:6 * i + 6])
            x = x2.reshape(B, self.num_patches, C)
        else:
            for i, block in enumerate(self.blocks):
                x = block(x, y, t0, y_lens, self.pos_embed_temporal if i == 0 else None)
        x = self.final_layer(x, t)
        x = self.unpatchify(x)
        return x.to(torch.float32)

    def unpatchify(self, x):
        N_t, N_h, N_w = [self.input_size[i] // self.patch_size[i] for i in range(3)]
        T_p, H_p, W_p = self.patch_size
        B = x.shape[0]
        x = x.reshape(B, N_t, N_h, N_w, T_p, H_p, W_p, self.out_channels)
        x = x.permute(0, 7, 1, 4, 2, 5, 3, 6)
        return x.reshape(B, self.out_channels, N_t * T_p, N_h * H_p, N_w * W_p)


def STDiT_XL_2(from_pretrained=None, **kwargs):
    """STDiT-XL/2 (stdit.py:454-484); checkpoints with fused ``qkv`` rows are split into q/k/v
    exactly as the reference does (:460-481) when a state dict is loaded via ``load_split_qkv``."""
    model = STDiT(depth=28, hidden_size=1152, patch_size=(1, 2, 2), num_heads=16, **kwargs)
    if from_pretrained is not None:
        load_split_qkv(model, torch.load(from_pretrained, map_location="cpu"))
    return model


def load_split_qkv(model: STDiT, state_dict: dict):
    """Load an OpenSORA checkpoint, splitting ``*.qkv.{weight,bias}`` rows into q/k/v
    (stdit.py:460-481, t2v/scripts/split_ckpt.py:3-17)."""
    sd = {}
    for k, v in state_dict.items():
        if k.endswith(".qkv.weight") or k.endswith(".qkv.bias"):
            base, kind = k.rsplit(".qkv.", 1)
            for name, part in zip("qkv", v.chunk(3, dim=0)):
                sd["%s.%s.%s" % (base, name, kind)] = part
        else:
            sd[k] = v
    return model.load_state_dict(sd, strict=False)
