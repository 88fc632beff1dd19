"""PixArt-alpha / PixArt-Sigma (MS) block and model forward on the gfx950 kernels.

Module / parameter names mirror t2i/diffusion/model/nets/{PixArt,PixArtMS,PixArt_blocks}.py so the
name routing of QuantModel(model_type='pixart') applies unchanged:
  blocks.{i}.attn.{qkv,proj} -> QuantAttnLinearImg      (fused qkv Linear, PixArt_blocks.py:132)
  blocks.{i}.cross_attn.{q_linear,kv_linear,proj} -> QuantCrossAttnLinearImg
  blocks.{i}.mlp.{fc1,fc2}, final_layer.linear, t_block.1, t_embedder.mlp.*, y_embedder.y_proj.* -> QuantLayer
  x_embedder.proj (Conv2d) -> QuantLayer, kept FP by the t2i FP list (quant_txt2img.py:293-295).
Unlike t2v, final_layer.linear is NOT in the t2i FP list and is quantized.

Block = STDiT block without the temporal branch (PixArtMS.py:71-79); the fused route reuses the same
kernels: LN+modulate+quant -> fused-qkv int8 GEMM -> flash attention over N tokens (4096 at 1024^2)
-> proj GEMM (+gate, +residual) -> varlen cross attention -> MLP.  kv-compression (sr_ratio > 1) and
qk_norm are not used by the quantized configs; they are implemented (round 6) with their FP parts as torch ops around the HIP attention.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..qdiff.models.quant_block import QuantAttention
from ..qdiff.models.quant_layer import QuantLayer
from ..qdiff.quantizer.dynamic_quantizer import DynamicActQuantizer
from ..t2v.stdit import (CaptionEmbedder, Mlp, fp_edge_linear, MultiHeadCrossAttention, STDiTBlock, T2IFinalLayer, TimestepEmbedder,
                         approx_gelu, get_1d_sincos_pos_embed_from_grid, seq_offsets, t2i_modulate)


def get_2d_sincos_pos_embed(embed_dim, grid_size, pe_interpolation=1.0, base_size=16):
    """PixArt.py:258-275."""
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    grid_h = np.arange(grid_size[0], dtype=np.float32) / (grid_size[0] / base_size) / pe_interpolation
    grid_w = np.arange(grid_size[1], dtype=np.float32) / (grid_size[1] / base_size) / pe_interpolation
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size[1], grid_size[0]])
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=2, in_chans=4, embed_dim=1152, bias=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x):
        proj = self.proj
        fp = not (getattr(proj, "weight_quant", False) or getattr(proj, "act_quant", False) or
                  getattr(proj, "smooth_quant", False))
        if fp and x.is_cuda:      # kernel == stride: the FP patch embedding is one matmul (see t2v/stdit.py PatchEmbed3D)
            w_ = getattr(proj, "org_weight", None)
            w_ = proj.weight if w_ is None else w_
            b_ = getattr(proj, "org_bias", None) if hasattr(proj, "org_weight") else proj.bias
            B, Cin, H, W = x.shape
            ph, pw = self.patch_size
            xp = x.reshape(B, Cin, H // ph, ph, W // pw, pw).permute(0, 2, 4, 1, 3, 5).reshape(-1, Cin * ph * pw)
            import torch.nn.functional as F
            out = F.linear(xp.to(w_.dtype), w_.reshape(w_.shape[0], -1), None if b_ is None else b_.to(w_.dtype))
            return out.reshape(B, -1, w_.shape[0])
        return self.proj(x).flatten(2).transpose(1, 2)


class SizeEmbedder(TimestepEmbedder):
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__(hidden_size=hidden_size, frequency_embedding_size=frequency_embedding_size)
        self.outdim = hidden_size

    def forward(self, s, bs):
        if s.ndim == 1:
            s = s[:, None]
        if s.shape[0] != bs:
            s = s.repeat(bs // s.shape[0], 1)
        b, dims = s.shape
        s_freq = self.timestep_embedding(s.reshape(-1), self.frequency_embedding_size)
        s_emb = self.mlp(s_freq.to(next(self.parameters()).dtype))
        return s_emb.reshape(b, dims * self.outdim)


class AttentionKVCompress(nn.Module):
    """Self-attention with a fused qkv Linear, optional LayerNorm on q / k and optional compression of the key / value tokens
    (PixArt_blocks.py:63-160).  ``sampling`` in (None, 'conv', 'ave', 'uniform', 'uniform_every'), ``sr_ratio`` the factor per
    image axis, ``qk_norm``: LayerNorm over all ``dim`` channels (affine) before the heads are split.  No released quantized
    config turns these on; they are here so that a PixArt checkpoint trained with them loads and runs (round 6).
    The attention itself is the HIP kernel with Lq = N queries and Lk = N / sr^2 keys; the LayerNorms, the token picks and
    the depthwise convolution of 'conv' are torch ops on the fp16 q | k | v buffer (they are FP in the reference as well)."""

    SAMPLINGS = (None, "conv", "ave", "uniform", "uniform_every")

    def __init__(self, dim, num_heads=8, qkv_bias=True, sampling=None, sr_ratio=1, qk_norm=False):
        super().__init__()
        if sampling not in self.SAMPLINGS:
            raise ValueError("sampling is one of %r, got %r" % (self.SAMPLINGS, sampling))
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.sampling, self.sr_ratio = sampling, int(sr_ratio)
        if self.sr_ratio > 1 and sampling == "conv":                   # average-pool initialisation (PixArt_blocks.py:86-91)
            self.sr = nn.Conv2d(dim, dim, groups=dim, kernel_size=self.sr_ratio, stride=self.sr_ratio)
            self.sr.weight.data.fill_(1 / self.sr_ratio ** 2)
            self.sr.bias.data.zero_()
            self.norm = nn.LayerNorm(dim)
        self.q_norm = nn.LayerNorm(dim) if qk_norm else nn.Identity()
        self.k_norm = nn.LayerNorm(dim) if qk_norm else nn.Identity()
        self.core = QuantAttention(num_heads, self.head_dim)

    @property
    def plain(self) -> bool:
        """No q / k LayerNorm, no compression: attention straight on the q | k | v buffer."""
        return self.sr_ratio <= 1 and isinstance(self.q_norm, nn.Identity)

    def _norm_fp32(self, ln, tns):
        if isinstance(ln, nn.Identity):
            return tns
        return F.layer_norm(tns.float(), ln.normalized_shape, ln.weight.float(), ln.bias.float(), ln.eps).to(tns.dtype)

    def downsample_2d(self, tns, H, W):
        """[B, N, C] tokens of an H x W grid -> [B, N / sr^2, C] (PixArt_blocks.py:99-124)."""
        sr, B, C = self.sr_ratio, tns.shape[0], tns.shape[-1]
        if self.sampling is None or sr <= 1:
            return tns
        if self.sampling == "uniform_every":
            return tns[:, ::sr].contiguous()
        g = tns.reshape(B, H, W, C).permute(0, 3, 1, 2)
        if self.sampling == "ave":                                     # nearest-neighbour picks (no averaging, despite the name)
            g = F.interpolate(g, scale_factor=1 / sr, mode="nearest").permute(0, 2, 3, 1)
        elif self.sampling == "uniform":
            g = g[:, :, ::sr, ::sr].permute(0, 2, 3, 1)
        else:                                                          # 'conv'
            if isinstance(self.sr, QuantLayer) and any(self.sr.get_quant_state()):
                # the reference wraps `attn.sr` as a QuantAttnLinearImg and dies in its forward on the 4-D input: there is no
                # behaviour to match
                raise NotImplementedError("a quantized `attn.sr` convolution: the reference cannot run it either - keep "
                                          "'attn.sr' in fp_layer_list")
            # (an FP-listed `attn.sr` is a QuantLayer in the FP state aliasing the convolution's parameters)
            g = F.conv2d(g.float(), self.sr.weight.float(), self.sr.bias.float(), stride=sr, groups=C)
            g = g.reshape(B, C, -1).permute(0, 2, 1)
            g = self._norm_fp32(self.norm, g).to(tns.dtype)
        return g.reshape(B, int(H / sr) * int(W / sr), C).contiguous()

    def attend(self, qkv, B, N, HW=None):
        """qkv [B*N, 3C] fp16 (q | k | v column blocks) -> [B*N, C] fp16."""
        if self.plain:
            return self.core.spatial(qkv, B, N)
        C = self.num_heads * self.head_dim
        H, W = HW if HW is not None else (int(N ** 0.5), int(N ** 0.5))
        q3 = self._norm_fp32(self.q_norm, qkv[:, :C].reshape(B, N, C)).contiguous()
        k3 = self._norm_fp32(self.k_norm, qkv[:, C:2 * C].reshape(B, N, C))
        k3 = self.downsample_2d(k3, H, W).contiguous()
        v3 = self.downsample_2d(qkv[:, 2 * C:].reshape(B, N, C), H, W).contiguous()
        Lk = k3.shape[1]
        out = torch.empty((B * N, C), dtype=torch.float16, device=qkv.device)
        ops.attn_fwd(q3.reshape(B * N, C), k3.reshape(B * Lk, C), v3.reshape(B * Lk, C), out, B, N, Lk, self.num_heads,
                     self.head_dim, N * C, C, Lk * C, C, N * C, C, scale=self.scale)
        return out

    def forward(self, x, mask=None, HW=None, block_id=None):
        B, N, C = x.shape
        dt = x.dtype
        qkv = self.qkv(x).reshape(B * N, 3 * C).half().contiguous()   # q | k | v column blocks (qkv.reshape(B,N,3,C))
        o = self.attend(qkv, B, N, HW)
        return self.proj(o.reshape(B, N, C).to(dt))


class PixArtMSBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, sampling=None, sr_ratio=1, qk_norm=False, **unused):
        super().__init__()
        self.hidden_size = hidden_size
        self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = AttentionKVCompress(hidden_size, num_heads=num_heads, qkv_bias=True, sampling=sampling, sr_ratio=sr_ratio,
                                        qk_norm=qk_norm)
        self.cross_attn = MultiHeadCrossAttention(hidden_size, num_heads)
        self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio), act_layer=approx_gelu)
        self.scale_shift_table = nn.Parameter(torch.randn(6, hidden_size) / hidden_size ** 0.5)

    def hot_layers(self) -> List[nn.Module]:
        return [self.attn.qkv, self.attn.proj, self.cross_attn.q_linear, self.cross_attn.kv_linear,
                self.cross_attn.proj, self.mlp.fc1, self.mlp.fc2]

    def fused_ok(self) -> bool:
        for m in self.hot_layers():
            if not (isinstance(m, QuantLayer) and m.int_route_ok()):
                return False
            if not isinstance(m.act_quantizer, DynamicActQuantizer) and m.act_quantizer.per_group:
                return False
            if getattr(m, "smooth_quant_running_stat", False):
                # the released t2i script leaves the running act-scale statistic ON for blocks.27.mlp.fc2 at inference
                # (quant_txt2img.py:297-300): that layer re-derives s from every input, which is QuantLayer.forward's
                # job (host-visible statistics) - such a block takes the reference data flow, layer by layer
                return False
        return True

    def forward(self, x, y, t, mask=None, HW=None, **kwargs):
        """PixArtMS.py:71-79 (reference data flow)."""
        B, N, C = x.shape
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + t.reshape(B, 6, -1)).chunk(6, dim=1)
        x = x + gate_msa * self.attn(t2i_modulate(self.norm1(x), shift_msa, scale_msa), HW=HW)
        x = x + self.cross_attn(x, y, mask)
        x = x + gate_mlp * self.mlp(t2i_modulate(self.norm2(x), shift_mlp, scale_mlp))
        return x

    def forward_fused(self, x2, y2, t0, kv_off, B, HW=None):
        """In-place update of x2 [B*N, C] fp16 (hot path; same kernel sequence as the STDiT block minus
        the temporal branch)."""
        C = self.hidden_size
        M = x2.shape[0]
        N = M // B
        a1, ca, fc1, fc2 = self.attn, self.cross_attn, self.mlp.fc1, self.mlp.fc2

        def sv(layer):
            r, alpha = layer._range_and_alpha()
            return r, layer.smooth_vector(r, alpha)

        mod = ops.adaln_table(self.scale_shift_table.detach(), t0.reshape(B, -1))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [mod[j] for j in range(6)]
        x3 = x2.view(B, N, C)
        st = a1.qkv.status
        r, s = sv(a1.qkv)
        qa = STDiTBlock._ln_quant(x3, shift_msa, scale_msa, (a1.qkv,), [s], st)[0]   # dynamic or calibrated static grid
        qkv = ops.gemm_i8(qa, a1.qkv.packed_weight(r, s), bias=a1.qkv.bias_f32())
        att_o = a1.attend(qkv, B, N, HW)                   # (plain: the HIP kernel straight on the q | k | v buffer)
        r, s = sv(a1.proj)
        qa = a1.proj.quantize_input(att_o.view(B, N, C), s)
        ops.gemm_i8(qa, a1.proj.packed_weight(r, s), bias=a1.proj.bias_f32(), out=x2, epilogue=ops.EPI_GATE_RESID,
                    resid=x2, gate=gate_msa, rows_per_gate=N)
        r, s = sv(ca.q_linear)
        q = ops.gemm_i8(ca.q_linear.quantize_input(x3, s), ca.q_linear.packed_weight(r, s), bias=ca.q_linear.bias_f32())
        r, s = sv(ca.kv_linear)
        from ..t2v.stdit import prompt_kv_exact_fill
        kv = prompt_kv_exact_fill(ca.kv_linear, y2.view(1, -1, C), r, s, ca.kv_linear.packed_weight(r, s))
        att_o = ca.core.cross(q, kv, kv_off, B, N, out=att_o)
        r, s = sv(ca.proj)
        ops.gemm_i8(ca.proj.quantize_input(att_o.view(B, N, C), s), ca.proj.packed_weight(r, s),
                    bias=ca.proj.bias_f32(), out=x2, epilogue=ops.EPI_RESID, resid=x2)
        r, s = sv(fc1)
        qa = STDiTBlock._ln_quant(x3, shift_mlp, scale_mlp, (fc1,), [s], st)[0]
        from ..t2v.stdit import _GELU_QUANT
        r2, s2 = sv(fc2)
        # GELU inside fc2's quantizer pass (see t2v/stdit.py), also for the uncond | cond pair of the t2i loop
        one_pass = _GELU_QUANT and fc2.gelu_one_pass_ok(B, fc2.in_features, s2)
        h = ops.gemm_i8(qa, fc1.packed_weight(r, s), bias=fc1.bias_f32(), epilogue=ops.EPI_NONE if one_pass else ops.EPI_GELU)
        r, s = r2, s2
        qa = fc2.quantize_gelu_input(h.view(B, N, -1), s) if one_pass else fc2.quantize_input(h.view(B, N, -1), s)
        ops.gemm_i8(qa, fc2.packed_weight(r, s), bias=fc2.bias_f32(), out=x2,
                    epilogue=ops.EPI_GATE_RESID, resid=x2, gate=gate_mlp, rows_per_gate=N)
        return x2


class PixArtBlock(PixArtMSBlock):
    """PixArt-alpha block (PixArt.py:25-57): the same adaLN-single block, called as ``block(x, y, t, mask)``."""

    def forward(self, x, y, t, mask=None, **kwargs):
        return super().forward(x, y, t, mask)


class _PixArtBase(nn.Module):
    """What PixArt.py:63-256 (alpha) and PixArtMS.py:82-260 (Sigma / multi-scale) share: embedders, 28 adaLN-single
    blocks, final layer, prompt-token selection, the per-block choice between the fused HIP route and the reference
    data flow.  Subclasses provide the positional embedding, the timestep conditioning and ``unpatchify``."""
    block_cls = PixArtMSBlock

    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, pred_sigma=True, caption_channels=4096,
                 pe_interpolation=1.0, model_max_length=120, qk_norm=False, kv_compress_config=None,
                 dtype=torch.float32):
        super().__init__()
        self.pred_sigma = pred_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if pred_sigma else in_channels
        self.patch_size, self.num_heads, self.depth = patch_size, num_heads, depth
        self.hidden_size = hidden_size
        self.pe_interpolation = pe_interpolation
        self.base_size = input_size // patch_size
        self.dtype = dtype
        self.x_embedder = PatchEmbed(patch_size, in_channels, hidden_size, bias=True)
        self.x_embedder.num_patches = (input_size // patch_size) ** 2
        self.t_embedder = TimestepEmbedder(hidden_size)
        # PixArt.py:99 registers the buffer (alpha uses it; MS recomputes the embedding per call and the t2i script
        # deletes the key from loaded checkpoints, quant_txt2img.py:232-233) - kept in both for state-dict compatibility
        self.register_buffer("pos_embed", torch.zeros(1, self.x_embedder.num_patches, hidden_size))
        self.t_block = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))
        self.y_embedder = CaptionEmbedder(in_channels=caption_channels, hidden_size=hidden_size,
                                          uncond_prob=class_dropout_prob, act_layer=approx_gelu,
                                          token_num=model_max_length)
        # PixArt.py:113-128 / PixArtMS.py:145-157: which blocks compress their keys / values, how, and the q / k LayerNorm
        self.qk_norm = bool(qk_norm)
        self.kv_compress_config = kv_compress_config or {"sampling": None, "scale_factor": 1, "kv_compress_layer": []}
        self.h = self.w = 0
        self._mask_cache = None

    def _make_blocks(self, hidden_size, num_heads, mlp_ratio, depth, patch_size):
        kc = self.kv_compress_config
        self.blocks = nn.ModuleList([
            self.block_cls(hidden_size, num_heads, mlp_ratio=mlp_ratio, sampling=kc["sampling"],
                           sr_ratio=int(kc["scale_factor"]) if i in kc["kv_compress_layer"] else 1, qk_norm=self.qk_norm)
            for i in range(depth)])
        self.final_layer = T2IFinalLayer(hidden_size, patch_size * patch_size, self.out_channels)

    def initialize_weights(self):
        """PixArt.py:213-249 / PixArtMS.py:237-260."""
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
        for emb in ("csize_embedder", "ar_embedder"):
            if hasattr(self, emb):
                nn.init.normal_(getattr(self, emb).mlp[0].weight, std=0.02)
                nn.init.normal_(getattr(self, emb).mlp[2].weight, std=0.02)
        nn.init.normal_(self.y_embedder.y_proj.fc1.weight, std=0.02)
        nn.init.normal_(self.y_embedder.y_proj.fc2.weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.cross_attn.proj.weight, 0)
            nn.init.constant_(block.cross_attn.proj.bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    def _select(self, y, mask, C):
        """masked_select of the prompt tokens (PixArt.py:160-167); the host copy of the lengths is cached per mask."""
        B = y.shape[0]
        if mask is None:
            return y.squeeze(1).reshape(1, -1, C), [y.shape[2]] * B
        key = (mask.data_ptr(), mask._version, tuple(mask.shape), B)
        if self._mask_cache is None or self._mask_cache[0] != key:
            m = mask if mask.shape[0] == B else mask.repeat(B // mask.shape[0], 1)
            m = m.reshape(B, -1)
            idx = torch.nonzero(m.reshape(-1) != 0, as_tuple=False).reshape(-1)
            self._mask_cache = (key, idx, [int(v) for v in m.sum(dim=1).tolist()], mask)   # mask kept alive
        _, idx, lens = self._mask_cache[:3]
        return y.squeeze(1).reshape(-1, C).index_select(0, idx).reshape(1, -1, C), lens

    def _run_blocks(self, x, y, t0, y_lens):
        """Every block whose Linears are all on the integer route runs fused (residual stream updated in place);
        any other block (FP / calibration states, a running smooth-quant statistic) runs the reference data flow."""
        bs, N, C = x.shape
        x = x.contiguous()
        can_fuse = x.is_cuda and x.dtype == torch.float16
        y2 = off = t0c = None
        for block in self.blocks:
            if can_fuse and block.fused_ok():
                if y2 is None:
                    y2 = y.reshape(-1, C).contiguous()
                    off = seq_offsets(y_lens, x.device)
                    t0c = t0.contiguous()
                x2 = x.reshape(bs * N, C)
                block.forward_fused(x2, y2, t0c, off, bs, HW=(self.h, self.w))
                x = x2.reshape(bs, N, C)
            else:
                x = block(x, y, t0, y_lens, HW=(self.h, self.w))
        return x

    def forward_with_cfg(self, x, timestep, y, cfg_scale, mask=None, **kwargs):
        """PixArt.py:183-196: one batched (cond | uncond) forward, guidance on the first three channels."""
        half = x[: len(x) // 2]
        combined = torch.cat([half, half], dim=0)
        model_out = self.forward(combined, timestep, y, mask, **kwargs)
        eps, rest = model_out[:, :3], model_out[:, 3:]
        cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
        half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
        eps = torch.cat([half_eps, half_eps], dim=0)
        return torch.cat([eps, rest], dim=1)


class PixArt(_PixArtBase):
    """PixArt-alpha (t2i/diffusion/model/nets/PixArt.py:63-256): FIXED sin-cos positional embedding held as the
    ``pos_embed`` buffer (computed at construction for ``input_size``), no micro-conditioning embedders, square
    ``unpatchify``.  This is the net the 256x256 configuration runs (quant_txt2img.py:226-231)."""
    block_cls = PixArtBlock

    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, pred_sigma=True, drop_path=0.0, caption_channels=4096,
                 pe_interpolation=1.0, config=None, model_max_length=120, qk_norm=False, kv_compress_config=None,
                 dtype=torch.float32, **kwargs):
        super().__init__(input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio,
                         class_dropout_prob, pred_sigma, caption_channels, pe_interpolation, model_max_length, qk_norm,
                         kv_compress_config, dtype)
        self._make_blocks(hidden_size, num_heads, mlp_ratio, depth, patch_size)
        self.initialize_weights()

    def initialize_weights(self):
        super().initialize_weights()
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5),
                                     pe_interpolation=self.pe_interpolation, base_size=self.base_size)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))      # PixArt.py:224-229

    def forward(self, x, timestep, y, mask=None, data_info=None, **kwargs):
        """PixArt.py:143-173."""
        x = x.to(self.dtype)
        timestep = timestep.to(self.dtype)
        y = y.to(self.dtype)
        C = self.hidden_size
        self.h, self.w = x.shape[-2] // self.patch_size, x.shape[-1] // self.patch_size
        x = self.x_embedder(x) + self.pos_embed.to(self.dtype)
        t = self.t_embedder(timestep, dtype=x.dtype)
        t0 = fp_edge_linear(self.t_block[1], t, act_in=ops.ACT_SILU)        # SiLU, Linear in one launch (FP edge kernel)
        if t0 is None:
            t0 = self.t_block(t)
        y = self.y_embedder(y, self.training)
        y, y_lens = self._select(y, mask, C)
        x = self._run_blocks(x, y, t0, y_lens)
        x = self.final_layer(x, t)
        return self.unpatchify(x)

    def forward_with_dpmsolver(self, x, timestep, y, mask=None, **kwargs):
        """PixArt.py:175-181: DPM-Solver needs no variance prediction."""
        return self.forward(x, timestep, y, mask).chunk(2, dim=1)[0]

    def unpatchify(self, x):
        """PixArt.py:198-211: square grids only (h = w = sqrt(N))."""
        c, p = self.out_channels, self.patch_size
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, c)
        x = torch.einsum("nhwpqc->nchpwq", x)
        return x.reshape(x.shape[0], c, h * p, h * p)


class PixArtMS(_PixArtBase):
    """PixArt-Sigma / multi-scale (PixArtMS.py:82-260): positional embedding recomputed for the input's own
    (h, w) grid, optional micro-conditioning (image size + aspect ratio embedders added to t, alpha at 1024)."""

    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, learn_sigma=True, pred_sigma=True, drop_path=0.0,
                 caption_channels=4096, pe_interpolation=1.0, config=None, model_max_length=120,
                 micro_condition=False, qk_norm=False, kv_compress_config=None, dtype=torch.float32, **kwargs):
        super().__init__(input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio,
                         class_dropout_prob, pred_sigma, caption_channels, pe_interpolation, model_max_length, qk_norm,
                         kv_compress_config, dtype)
        self.micro_conditioning = micro_condition
        if micro_condition:
            self.csize_embedder = SizeEmbedder(hidden_size // 3)
            self.ar_embedder = SizeEmbedder(hidden_size // 3)
        self._make_blocks(hidden_size, num_heads, mlp_ratio, depth, patch_size)
        self._pe_cache = {}
        self.initialize_weights()

    def initialize(self):
        self.initialize_weights()

    def _pos_embed(self, device, dtype):
        key = (self.h, self.w, str(device), dtype)
        pe = self._pe_cache.get(key)
        if pe is None:
            pe = torch.from_numpy(get_2d_sincos_pos_embed(self.hidden_size, (self.h, self.w),
                                                          pe_interpolation=self.pe_interpolation,
                                                          base_size=self.base_size)).unsqueeze(0).to(device).to(dtype)
            self._pe_cache[key] = pe
        return pe

    def forward(self, x, timestep, y, mask=None, data_info=None, **kwargs):
        """PixArtMS.py:165-211."""
        bs = x.shape[0]
        x = x.to(self.dtype)
        timestep = timestep.to(self.dtype)
        y = y.to(self.dtype)
        C = self.hidden_size
        self.h, self.w = x.shape[-2] // self.patch_size, x.shape[-1] // self.patch_size
        x = self.x_embedder(x) + self._pos_embed(x.device, self.dtype)
        t = self.t_embedder(timestep, dtype=x.dtype)
        if self.micro_conditioning:
            c_size, ar = data_info["img_hw"].to(self.dtype), data_info["aspect_ratio"].to(self.dtype)
            t = t + torch.cat([self.csize_embedder(c_size, bs), self.ar_embedder(ar, bs)], dim=1)
        t0 = fp_edge_linear(self.t_block[1], t, act_in=ops.ACT_SILU)        # SiLU, Linear in one launch (FP edge kernel)
        if t0 is None:
            t0 = self.t_block(t)
        y = self.y_embedder(y, self.training)
        y, y_lens = self._select(y, mask, C)
        x = self._run_blocks(x, y, t0, y_lens)
        x = self.final_layer(x, t)
        return self.unpatchify(x)

    def forward_with_dpmsolver(self, x, timestep, y, data_info=None, **kwargs):
        """PixArtMS.py:213-218: DPM-Solver needs no variance prediction."""
        return self.forward(x, timestep, y, data_info=data_info, **kwargs).chunk(2, dim=1)[0]

    def unpatchify(self, x):
        c, p = self.out_channels, self.patch_size
        assert self.h * self.w == x.shape[1]
        x = x.reshape(x.shape[0], self.h, self.w, p, p, c)
        x = torch.einsum("nhwpqc->nchpwq", x)
        return x.reshape(x.shape[0], c, self.h * p, self.w * p)


def PixArtMS_XL_2(**kwargs):
    """PixArtMS.py:268-270."""
    return PixArtMS(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kwargs)


def PixArt_XL_2(**kwargs):
    """PixArt.py:313-315: the alpha net (fixed positional embedding)."""
    return PixArt(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kwargs)
