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


def pixart_block(sd, i, x, y, t0, y_lens, H, spec: QSpec, t_id=0):
    p = "blocks.%d" % i
    B, N, C = x.shape
    D = C // H
    mods = (sd[p + ".scale_shift_table"].float()[None] + t0.float().reshape(B, 6, -1)).chunk(6, dim=1)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mods
    xm = fq.t2i_modulate(fq.layernorm_noaffine(x), shift_msa, scale_msa)
    qkv = qlinear(sd, p + ".attn.qkv", xm, spec, t_id).reshape(B, N, 3, H, D)
    o = attention_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5).reshape(B, N, C)
    x = x + gate_msa * qlinear(sd, p + ".attn.proj", o, spec, t_id)
    x = x + cross_attention(sd, p + ".cross_attn", x, y, y_lens, H, spec, t_id)
    h = qlinear(sd, p + ".mlp.fc1", fq.t2i_modulate(fq.layernorm_noaffine(x), shift_mlp, scale_mlp), spec, t_id)
    return x + gate_mlp * qlinear(sd, p + ".mlp.fc2", fq.gelu_tanh(h), spec, t_id)


def pixart_forward(sd, cfg: dict, x, timestep, y, mask, spec: QSpec, pos_embed: torch.Tensor, t_id=None,
                   return_blocks: bool = False):
    """cfg: dict(H, depth, patch, out_ch).  ``pos_embed`` [1, N, C] as the model computes it (MS) or holds it
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
        x = pixart_block(sd, i, x, yy, t0, y_lens, H, spec, t_id)
        if return_blocks:
            blocks.append(x.clone())
    shift, scale = (sd["final_layer.scale_shift_table"].float()[None] + t[:, None]).chunk(2, dim=1)
    xf = qlinear(sd, "final_layer.linear", fq.t2i_modulate(fq.layernorm_noaffine(x), shift, scale), spec, t_id)
    c = cfg["out_ch"]
    xf = xf.reshape(B, hh, ww, p_, p_, c)
    out = torch.einsum("nhwpqc->nchpwq", xf).reshape(B, c, hh * p_, ww * p_)
    return (out, blocks) if return_blocks else out
