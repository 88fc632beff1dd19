"""Oracle: STDiT block / model forward and the CFG + DDIM step, restated functionally (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Operates on a plain ``state_dict`` with the
reference's parameter names plus a small ``QSpec`` describing the quantization state, so it shares
no code with the product package.  Citations are into /root/reference.

Pinned against the imported reference (QuantModel(STDiT)) by tests/golden/make_golden.py; the
attention core is the reference's own non-flash branch (blocks.py:179-187) and a restated
block-diagonal SDPA for xformers (blocks.py:302-304, "parity unpinned" by any reference test).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import fakequant as fq


@dataclass
class QSpec:
    """Quantization state of the hot Linears (what the PTQ yaml + ckpt.pth determine)."""
    w_bits: int = 8
    a_bits: int = 8
    quant: bool = True                      # False -> FP forward
    fp_layers: Sequence[str] = ("x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer")
    # per-layer overrides of the CURRENT weight bit-width (mixed precision); grid stays at w_bits
    layer_w_bits: Dict[str, int] = field(default_factory=dict)
    # smooth quant: layer name -> act_scale [n_range, 1, K]; alpha list; timerange
    act_scale: Dict[str, torch.Tensor] = field(default_factory=dict)
    alpha: Optional[Sequence[float]] = None
    timerange: Sequence[Sequence[int]] = ((0, 1000),)
    # layers whose smooth quant was switched off (set_layer_smooth_quant on the script's FP list)
    smooth_off: Sequence[str] = ("x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer")
    # weight grids fixed at PTQ: name -> (delta [N,1], zp [N,1]); filled lazily when absent
    w_grid: Dict[str, tuple] = field(default_factory=dict)
    # activation quantizers: 'dynamic' (per token, online) or 'static' = calibrated grids a_grid[name] = (delta, zp),
    # tensor-wise (a_per_group False, scalars) or per token (a_per_group 'token', [1, n_tok, 1])
    # (base_quantizer.py:129-144 after init_done; configs w8a8_naive / *_ptqd)
    act_mode: str = "dynamic"
    a_per_group: object = "token"
    a_grid: Dict[str, tuple] = field(default_factory=dict)
    n_prompt: int = 0
    # layers whose smooth-quant act-scale statistic keeps running at inference (quant_layer.py:118-126; the t2i
    # scripts leave it on for the last block's mlp.fc2): updated in place in ``act_scale`` on every call
    running_stat: Sequence[str] = ()
    momentum: float = 0.95


def _is_fp(name: str, spec: QSpec) -> bool:
    return (not spec.quant) or any(name.startswith(p) for p in spec.fp_layers)


def qlinear(sd, name: str, x3: torch.Tensor, spec: QSpec, t_id: int = 0) -> torch.Tensor:
    """One (possibly) quantized Linear on the [B, n_tok, K] view its act quantizer sees.
    qdiff/models/quant_layer.py:99-225; grid quirk base_quantizer.py:126."""
    W = sd[name + ".weight"].float()
    b = sd.get(name + ".bias")
    b = None if b is None else b.float()
    if _is_fp(name, spec):
        xf = x3.float()
        if name in spec.act_scale and spec.alpha is not None and not any(name.startswith(p) for p in spec.smooth_off):
            # FP weight + smooth quant still on: QuantLayer folds s into W (quant_layer.py:188-189), the STDiT
            # attention Linears divide the input only (stdit_quant_layer.py:90,181,298) - as released
            r = fq.find_interval(spec.timerange, t_id)
            alpha = spec.alpha[r] if isinstance(spec.alpha, (list, tuple)) else spec.alpha
            sm = fq.smooth_scale(spec.act_scale[name][r], W, alpha)
            xf = xf / sm
            if ".mlp." in name:
                W = W * sm
        return F.linear(xf, W, b)
    smooth = None
    if name in spec.running_stat:
        # momentum statistic of max|x| per input channel, updated BEFORE s is derived (quant_layer.py:118-126)
        r = fq.find_interval(spec.timerange, t_id)
        cur = fq.act_scale_stat(x3)
        if name not in spec.act_scale:
            spec.act_scale[name] = torch.zeros([len(spec.timerange)] + list(cur.shape))
        a = spec.act_scale[name]
        a[r] = cur if float(a[r].abs().mean()) == 0 else a[r] * spec.momentum + cur * (1 - spec.momentum)
    if name in spec.act_scale and spec.alpha is not None:
        r = fq.find_interval(spec.timerange, t_id)
        alpha = spec.alpha[r] if isinstance(spec.alpha, (list, tuple)) else spec.alpha
        smooth = fq.smooth_scale(spec.act_scale[name][r], W, alpha)
    if spec.act_mode == "static":
        wd, wz = spec.w_grid[name] if name in spec.w_grid else fq.weight_params(W, spec.w_bits)
        spec.w_grid[name] = (wd, wz)
        ad, az = spec.a_grid[name]
        xq = x3
        if spec.a_per_group == "token" and name.endswith("kv_linear") and x3.shape[0] == 1 and spec.n_prompt:
            xq = x3.reshape(-1, spec.n_prompt, x3.shape[-1])          # stdit_quant_layer.py:272-278
        out = fq.quant_linear(xq, W, b, w_bits=spec.layer_w_bits.get(name, spec.w_bits), a_bits=spec.a_bits,
                              w_delta=wd, w_zp=wz, smooth=smooth, act_mode="static", a_delta=ad, a_zp=az)
        return out.reshape(*x3.shape[:-1], out.shape[-1])
    if name not in spec.w_grid:
        W0 = W
        if name in spec.act_scale and spec.alpha is not None:
            a0 = spec.alpha[0] if isinstance(spec.alpha, (list, tuple)) else spec.alpha
            W0 = W * fq.smooth_scale(spec.act_scale[name][0], W, a0)   # range-0 grid for every range
        spec.w_grid[name] = fq.weight_params(W0, spec.w_bits)
    wd, wz = spec.w_grid[name]
    return fq.quant_linear(x3, W, b, w_bits=spec.layer_w_bits.get(name, spec.w_bits), a_bits=spec.a_bits,
                           w_delta=wd, w_zp=wz, smooth=smooth)


def attention_core(q, k, v, scale):
    """q [n,Lq,H,D], k/v [n,Lk,H,D] -> [n,Lq,H,D]; fp32 softmax (blocks.py:179-187)."""
    a = torch.einsum("nqhd,nkhd->nhqk", q.float() * scale, k.float()).softmax(dim=-1)
    return torch.einsum("nhqk,nkhd->nqhd", a, v.float())


def self_attention(sd, prefix, x3v, Bp, Np, H, spec, t_id, view_B):
    """Attention.forward with separate q/k/v (blocks.py:151-195).  ``x3v`` is the [B, n_tok, C]
    view for the act quantizers; the attention itself runs over [Bp, Np]."""
    C = x3v.shape[-1]
    D = C // H
    q = qlinear(sd, prefix + ".q", x3v, spec, t_id).reshape(Bp, Np, H, D)
    k = qlinear(sd, prefix + ".k", x3v, spec, t_id).reshape(Bp, Np, H, D)
    v = qlinear(sd, prefix + ".v", x3v, spec, t_id).reshape(Bp, Np, H, D)
    o = attention_core(q, k, v, D ** -0.5).reshape(view_B, -1, C)
    return qlinear(sd, prefix + ".proj", o, spec, t_id)


def cross_attention(sd, prefix, x, y, y_lens, H, spec, t_id):
    """MultiHeadCrossAttention.forward with the block-diagonal mask (blocks.py:292-310)."""
    B, N, C = x.shape
    D = C // H
    q = qlinear(sd, prefix + ".q_linear", x, spec, t_id).reshape(B, N, H, D)
    kv = qlinear(sd, prefix + ".kv_linear", y, spec, t_id).reshape(-1, 2, H, D)
    outs, s = [], 0
    for b, L in enumerate(y_lens):
        k, v = kv[s:s + L, 0][None], kv[s:s + L, 1][None]
        outs.append(attention_core(q[b:b + 1], k, v, D ** -0.5))
        s += L
    o = torch.cat(outs).reshape(B, N, C)
    return qlinear(sd, prefix + ".proj", o, spec, t_id)


def stdit_block(sd, i: int, x, y, t0, y_lens, tpe, T, S, H, spec: QSpec, t_id: int = 0):
    """STDiTBlock.forward (opensora/models/stdit/stdit.py:96-133)."""
    p = "blocks.%d" % i
    B, N, C = x.shape
    x = x.float()
    mods = (sd[p + ".scale_shift_table"].float()[None] + t0.float().reshape(B, 6, -1)).chunk(6, dim=1)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mods
    x_m = fq.t2i_modulate(fq.layernorm_noaffine(x), shift_msa, scale_msa)
    # spatial: tokens regrouped (B T) S C, act-quant view [B, T*S, C] (stdit_quant_layer.py:68-73)
    x_s = self_attention(sd, p + ".attn", x_m, B * T, S, H, spec, t_id, B)
    x = x + gate_msa * x_s
    # temporal on the un-modulated x: (B S) T C; act-quant view [B, S*T, C] (:159-164)
    x_t = x.reshape(B, T, S, C).permute(0, 2, 1, 3)
    if tpe is not None:
        x_t = x_t + tpe.float().reshape(1, 1, T, C)
    x_t = self_attention(sd, p + ".attn_temp", x_t.reshape(B, S * T, C), B * S, T, H, spec, t_id, B)
    x_t = x_t.reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(B, N, C)
    x = x + gate_msa * x_t
    x = x + cross_attention(sd, p + ".cross_attn", x, y, y_lens, H, spec, t_id)
    h = qlinear(sd, p + ".mlp.fc1", fq.t2i_modulate(fq.layernorm_noaffine(x), shift_mlp, scale_mlp), spec, t_id)
    h = qlinear(sd, p + ".mlp.fc2", fq.gelu_tanh(h), spec, t_id)
    return x + gate_mlp * h


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-np.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def stdit_forward(sd, cfg: dict, x, timestep, y, mask, spec: QSpec, t_id: Optional[int] = None,
                  return_blocks: bool = False):
    """STDiT.forward (stdit.py:238-341).  Prompt tokens: masked_select (MASK_SELECT True) unless the activation
    quantizers are static AND per token - then the padding tokens are zeroed and every sample keeps all n_prompt
    rows (stdit.py:272-301).  cfg: dict(T,S,H,depth,patch,in_ch,out_ch,input_size)."""
    T, S, H, depth = cfg["T"], cfg["S"], cfg["H"], cfg["depth"]
    if t_id is None:
        t_id = int(timestep[0])
    x = x.float()
    w = sd["x_embedder.proj.weight"].float()
    x = F.conv3d(x, w, sd["x_embedder.proj.bias"].float(), stride=cfg["patch"])
    x = x.flatten(2).transpose(1, 2)
    B, N, C = x.shape
    x = (x.reshape(B, T, S, C) + sd["pos_embed"].float()).reshape(B, N, C)
    t = timestep_embedding(timestep.float())
    t = F.linear(F.silu(F.linear(t, sd["t_embedder.mlp.0.weight"].float(), sd["t_embedder.mlp.0.bias"].float())),
                 sd["t_embedder.mlp.2.weight"].float(), sd["t_embedder.mlp.2.bias"].float())
    t0 = F.linear(F.silu(t), sd["t_block.1.weight"].float(), sd["t_block.1.bias"].float())
    yy = F.linear(fq.gelu_tanh(F.linear(y.float(), sd["y_embedder.y_proj.fc1.weight"].float(),
                                        sd["y_embedder.y_proj.fc1.bias"].float())),
                  sd["y_embedder.y_proj.fc2.weight"].float(), sd["y_embedder.y_proj.fc2.bias"].float())
    mask_select = not (spec.quant and spec.act_mode == "static" and spec.a_per_group == "token")
    if mask is not None and mask_select:
        m = mask if mask.shape[0] == yy.shape[0] else mask.repeat(yy.shape[0] // mask.shape[0], 1)
        y_lens = [int(v) for v in m.sum(dim=1).tolist()]
        yy = yy.squeeze(1).masked_select(m.unsqueeze(-1) != 0).view(1, -1, C)
    elif mask is not None:
        m = mask if mask.shape[0] == yy.shape[0] else mask.repeat(2, 1)
        y_lens = [yy.shape[2]] * yy.shape[0]
        yy = (yy * m.unsqueeze(-1).unsqueeze(1)).squeeze(1).reshape(1, -1, C)
    else:
        y_lens = [yy.shape[2]] * yy.shape[0]
        yy = yy.squeeze(1).reshape(1, -1, C)
    blocks = []
    for i in range(depth):
        tpe = sd["pos_embed_temporal"] if i == 0 else None
        x = stdit_block(sd, i, x, yy, t0, y_lens, tpe, T, S, H, spec, t_id)
        if return_blocks:
            blocks.append(x.clone())
    shift, scale = (sd["final_layer.scale_shift_table"].float()[None] + t[:, None]).chunk(2, dim=1)
    xf = fq.t2i_modulate(fq.layernorm_noaffine(x), shift, scale)
    xf = qlinear(sd, "final_layer.linear", xf, spec, t_id)
    Nt, Nh, Nw = [cfg["input_size"][k] // cfg["patch"][k] for k in range(3)]
    Tp, Hp, Wp = cfg["patch"]
    out = xf.reshape(B, Nt, Nh, Nw, Tp, Hp, Wp, cfg["out_ch"]).permute(0, 7, 1, 4, 2, 5, 3, 6)
    out = out.reshape(B, cfg["out_ch"], Nt * Tp, Nh * Hp, Nw * Wp)
    return (out, blocks) if return_blocks else out


# ----------------------------------------------------------------------------- sampler
def linear_betas(n=1000):
    scale = 1000 / n
    return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)


def spaced_schedule(num_steps: int, diffusion_steps: int = 1000):
    """(timestep_map, alphas_cumprod) of SpacedDiffusion (respace.py:7-77) for "N" respacing."""
    base = np.cumprod(1.0 - linear_betas(diffusion_steps))
    size, frac = diffusion_steps, (diffusion_steps - 1) / (num_steps - 1) if num_steps > 1 else 1
    use, cur = [], 0.0
    for _ in range(num_steps):
        use.append(round(cur))
        cur += frac
    use = sorted(set(use))
    last, betas = 1.0, []
    for i in use:
        betas.append(1 - base[i] / last)
        last = base[i]
    acp = np.cumprod(1.0 - np.array(betas))
    return use, acp


def cfg_ddim_step(x, cond, uncond, acp, i, cfg_scale, k=0.0):
    """forward_with_cfg tail (iddpm/__init__.py:168-184) + p_mean_variance/ddim_sample with eta=0
    (gaussian_diffusion.py:252-335,514-552) for the kept half.  Coefficients go through float32
    exactly like ``_extract_into_tensor(...).float()``."""
    f32 = lambda v: torch.tensor(v, dtype=torch.float64).float()  # noqa: E731
    A = f32(np.sqrt(1.0 / acp[i]))
    Bc = f32(np.sqrt(1.0 / acp[i] - 1))
    abp = f32(1.0 if i == 0 else acp[i - 1])
    mo_c, mo_u = cond / (1 + k), uncond / (1 + k)
    C = x.shape[1]
    eps = torch.cat([mo_u[:, :3] + cfg_scale * (mo_c[:, :3] - mo_u[:, :3]), mo_c[:, 3:C]], dim=1)
    x0 = A * x - Bc * eps
    e2 = (A * x - x0) / Bc
    return x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - 0.0) * e2
