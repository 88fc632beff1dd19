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


# the ViDiT-Q W4A8 plan (values of configs/quant/opensora/w4a8_timestep_aware_cb.yaml): 4-bit per-channel
# weights with grids for [4,6,8], dynamic per-token uint8 activations, momentum channel balancing with
# alpha 0.11 in two time-ranges.  calib_data is the synthetic recipe of SURVEY.md 8d (2 prompts x 10 steps).
W4A8_TIMESTEP_AWARE = """
cfg_split: True
mixed_precision: [4,6,8]
calib_data: {n_steps: 10, batch_size: 2, n_samples: 2}
quant:
    weight:
        quantizer: {n_bits: 4, per_group: channel, channel_dim: 0, scale_method: min_max, round_mode: nearest}
    activation:
        quantizer:
            n_bits: 8
            per_group: token
            dynamic: True
            scale_method: min_max
            round_mode: nearest_ste
            running_stat: False
            sym: False
            n_spatial_token: 1024
            n_temporal_token: 16
            n_prompt: 120
            smooth_quant:
                enable: True
                channel_wise_scale_type: momentum_act_max
                momentum: 0.95
                alpha: [0.11, 0.11]
                timerange: [[0, 500], [501, 1000]]
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


def uses_smooth_quant(cfg: QuantConfig) -> bool:
    sq = cfg.quant.activation.quantizer.get("smooth_quant")
    return bool(sq and sq.get("enable"))


def wrap_model(model: STDiT, cfg: QuantConfig, fp_layers=REMAIN_FP) -> QuantModel:
    wq, aq = quant_params_from_config(cfg, T=model.num_temporal, S=model.num_spatial)
    qnn = QuantModel(model, wq, aq, model_type="opensora")
    qnn.cfg_split = bool(cfg.get("cfg_split", False))
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = list(fp_layers)
    return qnn


def set_inference_state(qnn: QuantModel, cfg: QuantConfig, fp_layers=REMAIN_FP):
    """The flags quant_txt2video.py:156-207 sets before load_quant_params (``--part_fp`` run): smooth quant
    on except for the FP list, weight+act quant on except for the FP list, init-done on both."""
    if uses_smooth_quant(cfg):
        qnn.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
        qnn.set_layer_smooth_quant(model=qnn, module_name_list=list(fp_layers), smooth_quant=False,
                                   smooth_quant_running_stat=False)
    qnn.set_quant_state(True, True)
    qnn.set_layer_quant(model=qnn, module_name_list=list(fp_layers), quant_level="per_layer", weight_quant=False,
                        act_quant=False, prefix="")
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")


def calibrate_synthetic(qnn: QuantModel, cfg: QuantConfig, fp_layers=REMAIN_FP, seed: int = 7):
    """Synthetic PTQ for smooth-quant configs: ``calib_data.n_samples`` synthetic prompts run
    through an FP DDIM trajectory of ``calib_data.n_steps`` steps, then the three passes of ptq.calibrate."""
    from . import ptq
    from .t2v.iddpm import IDDPM
    dev = next(qnn.model.parameters()).device
    n = int(cfg.calib_data.n_samples)
    m = qnn.model
    embeds, _ = synthetic_prompts(n, dev, model_max_length=m.y_embedder.y_embedding.shape[0],
                                  caption_channels=m.y_embedder.y_embedding.shape[1], seed=seed)
    sh = embeds["y"].shape
    y = embeds["y"].permute(1, 0, 2, 3, 4).reshape(n * sh[1], sh[2], sh[3], sh[4])
    z_size = (m.in_channels,) + tuple(m.input_size)
    z = torch.cat([synthetic_latent(1000 + i, z_size=z_size, seed=seed, device=dev) for i in range(n)])
    sch = IDDPM(num_sampling_steps=int(cfg.calib_data.n_steps), cfg_scale=4.0)
    traj = ptq.collect_calib_data(qnn, sch, z, y, embeds["mask"])
    data = ptq.get_quant_calib_data(cfg, traj)
    return ptq.calibrate(qnn, cfg, data, fp_layer_list=fp_layers, seed=seed)


def synthetic_mp_config(qnn: QuantModel, num_steps: int = 20):
    """Mixed-precision configs with the structure and bit allocation of
    configs/quant/opensora/mixed_precision/t20_weight_4_mp.yaml / t20_act_8_mp.yaml: four equal step ranges,
    attention Linears 4-bit and MLP Linears 8-bit weights in every range, 8-bit activations, and the
    (non-matching) ``fc1_`` / ``fc2_`` FP patterns the released file carries."""
    q = num_steps // 4
    keys = ["%d-%d" % (3 * q - 1, 2 * q), "%d-%d" % (4 * q - 1, 3 * q), "%d-%d" % (q - 1, 0), "%d-%d" % (2 * q - 1, q)]
    if hasattr(qnn, "quant_layers"):
        names = sorted("model." + n for n, _ in qnn.quant_layers() if n.startswith("blocks."))
    else:       # the bare model, before wrapping (shard.quantize_and_distribute wants the config up front): same names
        names = sorted("model." + n for n, m in qnn.named_modules()
                       if n.startswith("blocks.") and isinstance(m, torch.nn.Linear))
    w = {k: {n: (8 if ".mlp." in n else 4) for n in names} for k in keys}
    w["fp_layers"] = {k: ["fc1_", "fc2_"] for k in keys}
    a = {k: {n: 8 for n in names} for k in keys}
    return w, a


def quantize_model(model: STDiT, cfg: QuantConfig, fp_layers=REMAIN_FP) -> QuantModel:
    """model -> QuantModel in the state quant_txt2video.py:141-207 leaves it in.  Dynamic configs without
    smooth quant need no data (weights min-max); smooth-quant configs run the synthetic calibration."""
    qnn = wrap_model(model, cfg, fp_layers)
    if uses_smooth_quant(cfg):
        calibrate_synthetic(qnn, cfg, fp_layers)
        set_inference_state(qnn, cfg, fp_layers)
    else:
        init_weight_quantizers(qnn)
        qnn.set_quant_state(True, True)
    return qnn


def synthetic_prompts(n: int, device, model_max_length=120, caption_channels=4096, seed=43):
    """precompute_text_embeds-style dict: y [n,2,1,L,Cc] (cond, null on dim 1) and mask [n,L]."""
    g = torch.Generator().manual_seed(seed)
    y = (torch.randn(n, 2, 1, model_max_length, caption_channels, generator=g) * 0.1).half()
    gl = torch.Generator().manual_seed(seed + 1)
    lens = torch.randint(min(20, max(1, model_max_length // 2)), model_max_length + 1, (n,), generator=gl)
    mask = (torch.arange(model_max_length)[None, :] < lens[:, None]).to(torch.int64)
    return dict(y=y.to(device), mask=mask.to(device)), lens.tolist()


def synthetic_latent(prompt_index: int, z_size=(4, 16, 64, 64), seed=42, device="cpu"):
    """Per-prompt generator (seed + prompt index) so that 1-GPU and N-GPU runs draw identical noise."""
    g = torch.Generator().manual_seed(seed + prompt_index)
    return torch.randn(1, *z_size, generator=g).to(device)
