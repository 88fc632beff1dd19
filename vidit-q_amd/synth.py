"""Synthetic workloads (no checkpoints / datasets are reachable): SURVEY.md 8d.

STDiT-XL/2 is built with the reference initialisation under ``torch.manual_seed(0)``; every
zero-initialised tensor (attn_temp.proj, cross_attn.proj, final_layer.linear; stdit.py:407-442) is
re-drawn N(0, 0.02^2) so all branches carry signal; weights are cast to fp16.  Latents
z ~ N(0,1) [n,4,16,64,64] seed 42 (+prompt index), text embeds y ~ 0.1*N(0,1) [n,2,1,120,4096]
seed 43, prompt lengths uniform in [20,120] seed 44 (mask = prefix ones), cfg 4.0, ks = 0.
PTQ on synthetic weights: weights min-max per out-channel; activations dynamic per token.
"""
from __future__ import annotations

from typing import Optional

import torch

from .config import QuantConfig, to_config
from .qdiff.models.quant_model import QuantModel
from .qdiff.models.quant_layer import QuantLayer
from .t2v.stdit import STDiT

REMAIN_FP = ["x_embedder", "t_block", "t_embedder", "y_embedder", "final_layer"]   # remain_fp.txt

W8A8_DYNAMIC = """
cfg_split: True
mixed_precision: [4,6,8]
quant:
    weight:
        quantizer: {n_bits: 8, per_group: channel, channel_dim: 0, scale_method: min_max, round_mode: nearest}
    activation:
        quantizer:
            n_bits: 8
            per_group: token
            scale_method: min_max
            round_mode: nearest_ste
            running_stat: False
            dynamic: True
            sym: False
            n_spatial_token: 1024
            n_temporal_token: 16
            n_prompt: 120
            smooth_quant: {enable: False, channel_wise_scale_type: momentum_act_max, momentum: 0.95, alpha: 0.625}
"""


def quant_params_from_config(cfg: QuantConfig, T: Optional[int] = None, S: Optional[int] = None,
                             n_prompt: Optional[int] = None):
    """(wq_params, aq_params) the way quant_txt2video.py:120-139 derives them from the PTQ yaml."""
    wq = to_config(dict(cfg.quant.weight.quantizer))
    aq = to_config(dict(cfg.quant.activation.quantizer))
    if cfg.get("mixed_precision") is not None:
        wq["mixed_precision"] = cfg.mixed_precision           # quant_txt2video.py:137
    if T is not None:
        aq["n_temporal_token"] = T
    if S is not None:
        aq["n_spatial_token"] = S
    if n_prompt is not None:
        aq["n_prompt"] = n_prompt
    return wq, aq


def redraw_zero_init(model: torch.nn.Module, seed: int = 1, std: float = 0.02):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.abs().sum() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * std)


def build_stdit(device, depth=28, hidden_size=1152, num_heads=16, input_size=(16, 64, 64), model_max_length=120,
                caption_channels=4096, seed=0) -> STDiT:
    torch.manual_seed(seed)
    m = STDiT(input_size=input_size, depth=depth, hidden_size=hidden_size, num_heads=num_heads,
              model_max_length=model_max_length, caption_channels=caption_channels, dtype=torch.float16)
    redraw_zero_init(m, seed + 1)
    return m.half().to(device).eval()


def init_weight_quantizers(qnn: QuantModel):
    """Data-free weight PTQ for configs without smooth-quant: one min-max init per layer and
    bit-width (what the first weight-quantized forward of ptq.py:266-293 does), without running the
    model."""
    for name, layer in qnn.quant_layers():
        if getattr(layer, "smooth_quant", False):
            raise RuntimeError("smooth-quant configs need calibration data: use viditq_amd.ptq.calibrate")
        layer.weight_quantizer(layer.weight.detach())
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")


def quantize_model(model: STDiT, cfg: QuantConfig, fp_layers=REMAIN_FP) -> QuantModel:
    """model -> QuantModel in the state quant_txt2video.py:141-207 leaves it in (dynamic act configs)."""
    wq, aq = quant_params_from_config(cfg, T=model.num_temporal, S=model.num_spatial)
    qnn = QuantModel(model, wq, aq, model_type="opensora")
    qnn.cfg_split = bool(cfg.get("cfg_split", False))
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = list(fp_layers)
    init_weight_quantizers(qnn)
    qnn.set_quant_state(True, True)
    return qnn


def synthetic_prompts(n: int, device, model_max_length=120, caption_channels=4096, seed=43):
    """precompute_text_embeds-style dict: y [n,2,1,L,Cc] (cond, null on dim 1) and mask [n,L]."""
    g = torch.Generator().manual_seed(seed)
    y = (torch.randn(n, 2, 1, model_max_length, caption_channels, generator=g) * 0.1).half()
    gl = torch.Generator().manual_seed(seed + 1)
    lens = torch.randint(20, model_max_length + 1, (n,), generator=gl)
    mask = (torch.arange(model_max_length)[None, :] < lens[:, None]).to(torch.int64)
    return dict(y=y.to(device), mask=mask.to(device)), lens.tolist()


def synthetic_latent(prompt_index: int, z_size=(4, 16, 64, 64), seed=42, device="cpu"):
    """Per-prompt generator (seed + prompt index) so that 1-GPU and N-GPU runs draw identical noise."""
    g = torch.Generator().manual_seed(seed + prompt_index)
    return torch.randn(1, *z_size, generator=g).to(device)
