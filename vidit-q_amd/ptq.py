"""PTQ calibration producer: calibration data -> quant params in the reference's ``ckpt.pth`` schema.

Mirrors the training-free part of t2v/scripts/ptq.py (:207-362), the calibration-data collector
t2v/scripts/get_calib_data.py (the ``return_trajectory`` branch of ddim_sample_loop_progressive,
gaussian_diffusion.py:678-689) and qdiff/utils.py (get_quant_calib_data :20-63, load_quant_params
:65-70).  Runs once, offline; nothing here is on the per-step path.  The passes:

  1. smooth-quant statistics: FP forwards with ``smooth_quant_running_stat`` on; every QuantLayer keeps a
     momentum average of max|x| per input channel and per time-range (quant_layer.py:118-128);
  2. weight grids: weight quant on, act quant off, ONE forward per time-range start so that each
     WeightQuantizer computes min/max grids of W*s for every bit width of ``mixed_precision``;
  3. activation grids: skipped for dynamic quantizers; static tensor-wise / token-wise params take the
     last calibration batch (``running_stat: False``) exactly as in the reference.

Batch order and shuffling follow the reference (numpy RNG; pass ``seed`` to pin it).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import yaml

from .qdiff.models.quant_model import QuantModel


# --------------------------------------------------------------------------- calibration data
@torch.no_grad()
def collect_calib_data(qnn: QuantModel, scheduler, z: torch.Tensor, y: torch.Tensor,
                       mask: Optional[torch.Tensor]) -> Dict[str, List[torch.Tensor]]:
    """FP sampling with the trajectory recorded (get_calib_data.py): per DDIM step the model inputs
    ``xs`` [2n, ...] (kept half duplicated), ``ts`` [2n] (respaced -> raw timestep), ``cond_emb`` =
    y [2n, 1, L, Cc] and ``mask`` repeated to 2n rows; lists are ordered first step (t high) first."""
    state = qnn.get_quant_state()
    qnn.set_quant_state(False, False)
    # plain FP model, as get_calib_data.py runs it: no channel balancing (its statistics do not exist yet)
    flags = [(layer, layer.smooth_quant, getattr(layer, "smooth_quant_running_stat", False))
             for _, layer in qnn.quant_layers()]
    for layer, _, _ in flags:
        layer.smooth_quant = False
        layer.smooth_quant_running_stat = False
    data = {"xs": [], "ts": [], "cond_emb": [], "mask": []}
    n = z.shape[0]
    x = z.float()
    buf = torch.empty_like(x)
    from .t2v.iddpm import model_forward_pair
    cfg_split = bool(getattr(qnn, "cfg_split", False))
    for i in list(range(scheduler.num_timesteps))[::-1]:
        t_id = scheduler.timestep_map[i]
        t = torch.full((n,), t_id, device=x.device, dtype=torch.long)
        data["xs"].append(torch.cat([x, x]).cpu())
        data["ts"].append(torch.cat([t, t]).cpu())
        data["cond_emb"].append(y.cpu())
        m = mask
        if m is not None and m.shape[0] != y.shape[0]:
            m = m.repeat(y.shape[0] // m.shape[0], 1)
        data["mask"].append(None if m is None else m.cpu())
        cond, uncond = model_forward_pair(qnn, x, t, y[:n], y[n:], mask, cfg_split, t_id, {})
        out = scheduler.ddim_step(x, cond, uncond, i, scheduler.cfg_scale, 0.0, out=buf)
        x, buf = out, x
    for layer, sq, rs in flags:
        layer.smooth_quant, layer.smooth_quant_running_stat = sq, rs
    qnn.set_quant_state(*state)
    return data


def get_quant_calib_data(config, sample_data, custom_steps=None, repeat_interleave: bool = False):
    """qdiff/utils.py:20-63: pick ``calib_data.n_steps`` evenly spaced steps and the first
    ``2 * n_samples`` rows of each; returns (xs, ts, cond_embs, masks) concatenated over steps."""
    num_samples = config.calib_data.n_samples
    num_st = config.calib_data.n_steps
    nsteps = len(sample_data["ts"])
    if custom_steps is None:
        custom_steps = num_st
    assert nsteps >= custom_steps
    if repeat_interleave:
        raise NotImplementedError("timestep_wise calibration is commented out in every shipped config")
    timesteps = list(range(0, nsteps, nsteps // num_st))
    take = lambda key: torch.cat([sample_data[key][i][:num_samples * 2].reshape(-1, *sample_data[key][i].shape[1:])
                                  for i in timesteps], dim=0)   # noqa: E731
    return take("xs"), take("ts"), take("cond_emb"), take("mask")


# --------------------------------------------------------------------------- the three passes
@torch.no_grad()
def calibrate(qnn: QuantModel, config, calib_data, fp_layer_list: Optional[Sequence[str]] = None,
              samples_per_step: Optional[int] = None, batch_size: Optional[int] = None,
              seed: Optional[int] = None) -> dict:
    """ptq.py:207-362 for ``model_type: opensora``.  ``calib_data`` = (xs, ts, cond_embs, masks) as returned by
    get_quant_calib_data; ``fp_layer_list`` = the ``part_fp_list`` lines (None: everything quantized).
    ``samples_per_step`` / ``batch_size`` default to ``2 * calib_data.n_samples`` / ``2 * calib_data.batch_size``.
    Returns the quant-param dict (``get_quant_params_dict``); the model is left in state (True, True)."""
    if seed is not None:
        np.random.seed(seed)
    aq_params = config.quant.activation.quantizer
    dev = next(qnn.model.parameters()).device
    xs, ts, cs, masks = calib_data
    bs = batch_size if batch_size is not None else config.calib_data.batch_size * 2
    per_step = samples_per_step if samples_per_step is not None else config.calib_data.n_samples * 2
    fp = list(fp_layer_list) if fp_layer_list is not None else None
    smooth = bool(aq_params.get("smooth_quant") and aq_params.smooth_quant.get("enable"))

    def part_state(w, a):
        qnn.set_quant_state(w, a)
        if fp is not None:
            qnn.set_layer_quant(model=qnn, module_name_list=fp, quant_level="per_layer", weight_quant=False,
                                act_quant=False, prefix="")

    def fwd(x, t, c, m):
        return qnn(x.to(dev), t.to(dev), c.to(dev), mask=None if m is None else m.to(dev))

    tmp_mask = None if masks is None else masks[:bs][::2]         # the model takes 2*bs conds and bs masks
    qnn.set_module_name_for_quantizer(module=qnn.model)

    # ---- 1. smooth-quant statistics (:219-264)
    if smooth:
        qnn.set_smooth_quant(smooth_quant=False, smooth_quant_running_stat=True)
        qnn.set_quant_state(False, False)
        ts2 = ts.reshape(-1, per_step)
        n_steps = ts2.shape[0]
        xs2 = xs.reshape(n_steps, per_step, *xs.shape[1:])
        cs2 = cs.reshape(n_steps, per_step, *cs.shape[1:])
        ms2 = masks.reshape(n_steps, per_step, *masks.shape[1:])
        inds = np.arange(per_step)
        np.random.shuffle(inds)
        rounds = per_step // bs
        for i_ts in range(n_steps):
            assert torch.all(ts2[i_ts] == ts2[i_ts, 0])
            for i in range(rounds):
                sel = torch.as_tensor(inds[i * bs:(i + 1) * bs])
                fwd(xs2[i_ts, sel], ts2[i_ts, sel], cs2[i_ts, sel], ms2[i_ts, sel])
        qnn.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
        if fp is not None:
            qnn.set_layer_smooth_quant(model=qnn, module_name_list=fp, smooth_quant=False,
                                       smooth_quant_running_stat=False)

    # ---- 2. weight grids (:266-293)
    part_state(True, False)
    if smooth and aq_params.smooth_quant.get("timerange") is not None:
        for range_start in [tr[0] for tr in aq_params.smooth_quant.timerange]:
            fwd(xs[:bs], ts[:bs].clone().fill_(range_start), cs[:bs], tmp_mask)
    else:
        fwd(xs[:bs], ts[:bs], cs[:bs], tmp_mask)
    qnn.set_quant_init_done("weight")

    # ---- 3. activation grids (:296-361)
    part_state(True, True)
    if not aq_params.get("dynamic", False):
        if config.get("timestep_wise", False):
            raise NotImplementedError("timestep_wise calibration is commented out in every shipped config")
        rounds = xs.shape[0] // bs
        for i in range(rounds):
            sl = slice(i * bs, (i + 1) * bs)
            fwd(xs[sl], ts[sl], cs[sl], None if masks is None else masks[sl][::2])
    qnn.set_quant_init_done("activation")
    return qnn.get_quant_params_dict()


PIXART_FP_LAYERS = ("x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder")


@torch.no_grad()
def calibrate_pixart(qnn: QuantModel, config, calib_data, batch_size: Optional[int] = None,
                     fp_layer_list: Sequence[str] = PIXART_FP_LAYERS,
                     smooth_quant_layer_list: Sequence[str] = ("blocks.27.mlp.fc2",), fwd_kwargs: Optional[dict] = None) -> dict:
    """The training-free part of t2i/scripts/ptq.py (:218-318) for ``model_type: pixart``; its order differs from the
    t2v script: (0) smooth quant - if enabled at all - is switched on together with its running statistics for the
    listed layers only (the script hard-codes the last block's fc2, :222-226), (1) one FP forward, (2) weight grids
    from one forward with weight quant on, (3) the FP layer list of the script (final_layer stays quantized), then
    static activation grids over the calibration batches unless the quantizer is dynamic.  ``calib_data`` =
    (xs, ts, cond_embs, masks); ``fwd_kwargs``: extra model kwargs (``data_info`` ...).  Returns the quant-param
    dict; the model is left in state (True, True)."""
    aq_params = config.quant.activation.quantizer
    dev = next(qnn.model.parameters()).device
    xs, ts, cs, masks = calib_data
    bs = batch_size if batch_size is not None else config.calib_data.batch_size
    kw = dict(fwd_kwargs or {})

    def fwd(sl, mask_rows):
        return qnn(xs[sl].to(dev), ts[sl].to(dev), cs[sl].to(dev), mask=None if masks is None else mask_rows.to(dev), **kw)

    first = slice(0, bs)
    if aq_params.get("smooth_quant") and aq_params.smooth_quant.get("enable"):
        qnn.set_smooth_quant(smooth_quant=False, smooth_quant_running_stat=False)
        qnn.set_layer_smooth_quant(model=qnn, module_name_list=list(smooth_quant_layer_list), smooth_quant=True,
                                   smooth_quant_running_stat=True)
    fwd(first, None if masks is None else masks[first])                       # :236
    qnn.set_module_name_for_quantizer(module=qnn.model)
    qnn.set_quant_state(True, False)                                          # :240-244
    fwd(first, None if masks is None else masks[first])
    qnn.set_quant_init_done("weight")
    qnn.set_quant_state(True, True)                                           # :249-252
    qnn.fp_layer_list = list(fp_layer_list)
    qnn.set_layer_quant(model=qnn, module_name_list=list(fp_layer_list), quant_level="per_layer", weight_quant=False,
                        act_quant=False, prefix="")
    if not aq_params.get("dynamic", False):
        if config.get("timestep_wise", False):
            raise NotImplementedError("timestep_wise activation calibration (t2i/scripts/ptq.py:275-308)")
        for i in range(xs.shape[0] // bs):                                    # :259-272
            sl = slice(i * bs, (i + 1) * bs)
            fwd(sl, None if masks is None else masks[sl][::2])
    qnn.set_quant_init_done("activation")
    return qnn.get_quant_params_dict()


# --------------------------------------------------------------------------- ckpt.pth / yaml IO
def save_quant_params(qnn: QuantModel, path: str, dtype=torch.float32) -> dict:
    """ptq.py:426-428: ``torch.save(qnn.get_quant_params_dict(), <outdir>/ckpt.pth)``."""
    d = qnn.get_quant_params_dict(dtype=dtype)
    torch.save(d, path)
    return d


@torch.no_grad()
def load_quant_params(qnn: QuantModel, ckpt_path: str, dtype=torch.float32):
    """qdiff/utils.py:65-70."""
    ckpt = torch.load(ckpt_path, map_location="cpu")
    qnn.set_module_name_for_quantizer(module=qnn.model)
    qnn.set_quant_params_dict(ckpt, dtype=dtype)


def load_mp_config(path: str) -> dict:
    """Mixed-precision YAML (configs/quant/opensora/mixed_precision/*.yaml): plain
    ``{"hi-lo": {layer: bits}, ..., "fp_layers": {"hi-lo": [patterns]}}`` (quant_txt2video_mp.py:533-537)."""
    with open(path) as f:
        return yaml.safe_load(f)


def enable_timestep_wise_mp(qnn: QuantModel, weight_cfg: dict, act_cfg: dict):
    """The three attributes the reference script sets (quant_txt2video_mp.py:373,539-540)."""
    qnn.timestep_wise_mp = True
    qnn.time_mp_config_weight = weight_cfg
    qnn.time_mp_config_act = act_cfg
