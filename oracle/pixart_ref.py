"""Oracle: PixArt-alpha / -Sigma (MS) block and model forward, restated (torch CPU fp32).

TEST INFRASTRUCTURE ONLY.  Follows t2i/diffusion/model/nets/PixArtMS.py:71-79 (block), :165-211
(model), PixArt_blocks.py:125-160 (fused-qkv self attention) and :43-60 (cross attention); quantized
Linears as qdiff/models/dit_quant_layer.py:14-79 (no smooth-quant branch; the mlp Linears are plain
QuantLayers and DO have one - QSpec.act_scale / running_stat).  The alpha net (PixArt.py:143-173) is the same
forward with ``pos_embed`` = the model's fixed buffer (``sd['pos_embed']``) and no micro-conditioning.
Pinned on goldens produced by the imported reference (tests/golden/make_golden.py::tiny_pixart,
::tiny_pixart_alpha, ::tiny_pixart_w4a8).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import fakequant as fq
from .stdit_ref import QSpec, attention_core, cross_attention, qlinear, timestep_embedding

T2I_FP_LAYERS = ("x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder")


def kv_downsample(sd, prefix, tns, hw, sr, sampling):
    """AttentionKVCompress.downsample_2d (PixArt_blocks.py:99-124): the key / value tokens [B, N, C] of an hh x ww grid reduced
    by `sr` per axis - 'uniform_every': every sr-th TOKEN; 'uniform': every sr-th row and column; 'ave': nearest-neighbour
    interpolation by 1 / sr (for an integer factor the same picks as 'uniform', despite the name); 'conv': the depthwise
    `sr` convolution (kernel = stride = sr) followed by the LayerNorm `norm`."""
    B, N, C = tns.shape
    if sampling == "uniform_every":
        return tns[:, ::sr]
    hh, ww = hw
    g = tns.reshape(B, hh, ww, C).permute(0, 3, 1, 2)
    if sampling == "ave":
        g = F.interpolate(g, scale_factor=1 / sr, mode="nearest").permute(0, 2, 3, 1)
    elif sampling == "uniform":
        g = g[:, :, ::sr, ::sr].permute(0, 2, 3, 1)
    elif sampling == "conv":
        g = F.conv2d(g, sd[prefix + ".sr.weight"].float(), sd[prefix + ".sr.bias"].float(), stride=sr, groups=C)
        g = g.reshape(B, C, -1).permute(0, 2, 1)
        g = F.layer_norm(g, (C,), sd[prefix + ".norm.weight"].float(), sd[prefix + ".norm.bias"].float())
    else:
        raise ValueError(sampling)
    return g.reshape(B, int(hh / sr) * int(ww / sr), C)


def pixart_block(sd, i, x, y, t0, y_lens, H, spec: QSpec, t_id=0, hw=None, kv=None, qk_norm=False):
    """``kv``: None or dict(sampling, sr, layers) - key / value compression of the blocks in ``layers``; ``qk_norm``: LayerNorm
    (over all C channels, affine) on q and k before the heads are split (PixArt_blocks.py:135-146)."""
    p = "blocks.%d" % i
    B, N, C = x.shape
    D = C // H
    mods = (sd[p + ".scale_shift_table"].float()[None] + t0.float().reshape(B, 6, -1)).chunk(6, dim=1)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mods
    xm = fq.t2i_modulate(fq.layernorm_noaffine(x), shift_msa, scale_msa)
    qkv = qlinear(sd, p + ".attn.qkv", xm, spec, t_id).reshape(B, N, 3, C)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if qk_norm:
        q = F.layer_norm(q, (C,), sd[p + ".attn.q_norm.weight"].float(), sd[p + ".attn.q_norm.bias"].float())
        k = F.layer_norm(k, (C,), sd[p + ".attn.k_norm.weight"].float(), sd[p + ".attn.k_norm.bias"].float())
    if kv is not None and i in kv["layers"] and kv["sr"] > 1:
        hw_ = hw if hw is not None else (int(N ** 0.5), int(N ** 0.5))
        k = kv_downsample(sd, p + ".attn", k, hw_, kv["sr"], kv["sampling"])
        v = kv_downsample(sd, p + ".attn", v, hw_, kv["sr"], kv["sampling"])
    o = attention_core(q.reshape(B, N, H, D), k.reshape(B, -1, H, D), v.reshape(B, -1, H, D), D ** -0.5).reshape(B, N, C)
    x = x + gate_msa * qlinear(sd, p + ".attn.proj", o, spec, t_id)
    x = x + cross_attention(sd, p + ".cross_attn", x, y, y_lens, H, spec, t_id)
    h = qlinear(sd, p + ".mlp.fc1", fq.t2i_modulate(fq.layernorm_noaffine(x), shift_mlp, scale_mlp), spec, t_id)
    return x + gate_mlp * qlinear(sd, p + ".mlp.fc2", fq.gelu_tanh(h), spec, t_id)


def pixart_forward(sd, cfg: dict, x, timestep, y, mask, spec: QSpec, pos_embed: torch.Tensor, t_id=None,
                   return_blocks: bool = False):
    """cfg: dict(H, depth, patch, out_ch[, kv=dict(sampling, sr, layers), qk_norm]).  ``pos_embed`` [1, N, C] as the model computes it (MS) or holds it
    (alpha: sd['pos_embed']).  ``t_id``: QuantModel pushes timestep[0] to every layer (quant_model.py:347)."""
    H, depth, p_ = cfg["H"], cfg["depth"], cfg["patch"]
    if t_id is None:
        t_id = int(timestep[0])
    x = F.conv2d(x.float(), sd["x_embedder.proj.weight"].float(), sd["x_embedder.proj.bias"].float(), stride=p_)
    hh, ww = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2) + pos_embed.float()
    B, N, C = x.shape
    t = timestep_embedding(timestep.float())
    t = F.linear(F.silu(F.linear(t, sd["t_embedder.mlp.0.weight"].float(), sd["t_embedder.mlp.0.bias"].float())),
                 sd["t_embedder.mlp.2.weight"].float(), sd["t_embedder.mlp.2.bias"].float())
    t0 = F.linear(F.silu(t), sd["t_block.1.weight"].float(), sd["t_block.1.bias"].float())
    yy = F.linear(fq.gelu_tanh(F.linear(y.float(), sd["y_embedder.y_proj.fc1.weight"].float(),
                                        sd["y_embedder.y_proj.fc1.bias"].float())),
                  sd["y_embedder.y_proj.fc2.weight"].float(), sd["y_embedder.y_proj.fc2.bias"].float())
    if mask is not None:
        m = mask if mask.shape[0] == yy.shape[0] else mask.repeat(yy.shape[0] // mask.shape[0], 1)
        y_lens = [int(v) for v in m.sum(dim=1).tolist()]
        yy = yy.squeeze(1).masked_select(m.unsqueeze(-1) != 0).view(1, -1, C)
    else:
        y_lens = [yy.shape[2]] * yy.shape[0]
        yy = yy.squeeze(1).reshape(1, -1, C)
    blocks = []
    for i in range(depth):
        x = pixart_block(sd, i, x, yy, t0, y_lens, H, spec, t_id, hw=(hh, ww), kv=cfg.get("kv"), qk_norm=cfg.get("qk_norm", False))
        if return_blocks:
            blocks.append(x.clone())
    shift, scale = (sd["final_layer.scale_shift_table"].float()[None] + t[:, None]).chunk(2, dim=1)
    xf = qlinear(sd, "final_layer.linear", fq.t2i_modulate(fq.layernorm_noaffine(x), shift, scale), spec, t_id)
    c = cfg["out_ch"]
    xf = xf.reshape(B, hh, ww, p_, p_, c)
    out = torch.einsum("nhwpqc->nchpwq", xf).reshape(B, c, hh * p_, ww * p_)
    return (out, blocks) if return_blocks else out
