"""Prompt sharding over the GPUs of one node.

The reference's quantized inference is single-GPU (quant_txt2video.py:71-72).  Sampling
trajectories of different prompts are independent, so the multi-GPU form is: prompt i -> rank
i mod R, each rank runs the whole 100-step loop locally, and the only collectives are
  (1) ONE broadcast from rank 0 of the packed int weights + quant grids, as a single flat byte
      buffer (W8: ~0.75 GB; xGMI ring broadcast is per-link bound, one large message amortises
      launch latency), before the loop;
  (2) an optional gather of the final latents [n,4,16,64,64] after it.
No collective runs inside a denoising step.  Backend: ``nccl`` (= RCCL) on GPUs, ``gloo`` in the
CPU tests of the sharding logic.

Per-token activation scales are reduced over the batch dimension in the reference
(base_quantizer.py:185), so batching several prompts into one forward would change results; every
forward therefore carries one prompt (B=1 with cfg_split, B=2 cond+uncond without), exactly like the
reference's ``batch_size = 1`` config.  Noise is drawn from a per-prompt generator (seed + prompt
index) so 1-GPU and N-GPU runs produce identical latents.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import synth
from .qdiff.models.quant_layer import QuantLayer
from .qdiff.models.quant_model import QuantModel

_ALIGN = 256


def prompts_of_rank(n_prompts: int, rank: int, world: int) -> List[int]:
    """Round-robin partition: prompt i -> rank i mod world."""
    return [i for i in range(n_prompts) if i % world == rank]


# --------------------------------------------------------------------------- flat blob
def _layout(tensors: List[Tuple[str, torch.Tensor]]):
    meta, off = [], 0
    for name, t in tensors:
        nbytes = t.numel() * t.element_size()
        meta.append((name, str(t.dtype).replace("torch.", ""), tuple(t.shape), off, nbytes))
        off += (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    return meta, off


def pack_blob(tensors: List[Tuple[str, torch.Tensor]], device) -> Tuple[list, torch.Tensor]:
    meta, total = _layout(tensors)
    blob = torch.zeros(max(total, 1), dtype=torch.uint8, device=device)
    for (name, dt, shape, off, nbytes), (_, t) in zip(meta, tensors):
        blob[off:off + nbytes] = t.contiguous().reshape(-1).view(torch.uint8)
    return meta, blob


def unpack_blob(meta: list, blob: torch.Tensor) -> Dict[str, torch.Tensor]:
    out = {}
    for name, dt, shape, off, nbytes in meta:
        out[name] = blob[off:off + nbytes].view(getattr(torch, dt)).reshape(shape)   # zero-copy views
    return out


# --------------------------------------------------------------------------- what travels
def _small_state(qnn: QuantModel) -> List[Tuple[str, torch.Tensor]]:
    """Weight-quantizer grids and smooth-quant statistics of every QuantLayer (a few MB in total)."""
    items = []
    for name, layer in qnn.quant_layers():
        wq = layer.weight_quantizer
        for b in ("delta_list", "zero_point_list", "delta", "zero_point"):
            v = getattr(wq, b)
            if v is not None:
                items.append(("%s|wq|%s" % (name, b), v))
        aq = layer.act_quantizer
        if getattr(aq, "act_scale", None) is not None:
            items.append(("%s|aq|act_scale" % name, aq.act_scale))
    return items


def _pack_jobs(qnn: QuantModel):
    """(layer name, layer, time-range id, representative timestep) of every packed weight the hot loop will ask for."""
    jobs = []
    for name, layer in qnn.quant_layers():
        if layer.weight_quant and layer.act_quant and layer._can_pack():
            n_r = len(layer.timerange) if getattr(layer, "smooth_quant", False) else 1
            for r in range(n_r):
                jobs.append((name, layer, r, layer.timerange[r][0] if n_r > 1 else None))
    return jobs


def _pack_one(layer: QuantLayer, r: int, t_id, out=None):
    saved = layer.cur_timestep_id
    if t_id is not None:
        layer.cur_timestep_id = t_id
    rr, alpha = layer._range_and_alpha()
    pw = layer.packed_weight(rr, layer.smooth_vector(rr, alpha), out=out)
    layer.cur_timestep_id = saved
    return pw


def prepack(qnn: QuantModel):
    """Pack every quantized Linear for every time-range so nothing is packed inside the loop."""
    for _, layer, r, t_id in _pack_jobs(qnn):
        _pack_one(layer, r, t_id)


def prepack_into_arena(qnn: QuantModel):
    """Like :func:`prepack`, but the packed codes and per-channel terms are written straight into ONE flat byte
    buffer - the buffer that is then broadcast as is.  Rank 0 therefore never holds a second copy of the ~0.75 GB of
    packed weights (W8A8 STDiT-XL/2); the small state (grids, act scales) is copied behind them.  Returns
    (meta, arena)."""
    from . import ops
    dev = next(qnn.model.parameters()).device
    jobs = _pack_jobs(qnn)
    specs = []                                            # (key, shape, dtype) in arena order
    for name, layer, r, _ in jobs:
        N, K = layer.weight.shape
        nb = layer.weight_quantizer.n_bits
        for f, (shape, dt) in zip(("wq", "sw", "zw", "cs"), ops.packed_shapes(N, K, nb)):
            specs.append(("%s|pw|%d|%d|%s" % (name, r, nb, f), tuple(shape), dt))
    small = _small_state(qnn)
    specs += [(k, tuple(v.shape), v.dtype) for k, v in small]
    meta, total = _layout([(k, torch.empty(shape, dtype=dt, device="meta")) for k, shape, dt in specs])
    arena = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
    views = unpack_blob(meta, arena)
    for name, layer, r, t_id in jobs:
        nb = layer.weight_quantizer.n_bits
        out = [views["%s|pw|%d|%d|%s" % (name, r, nb, f)] for f in ("wq", "sw", "zw", "cs")]
        _pack_one(layer, r, t_id, out=out)
    for k, v in small:
        views[k].copy_(v)
    return meta, arena


def _install_quant_state(qnn: QuantModel, tensors: Dict[str, torch.Tensor]):
    from . import ops
    layers = dict(qnn.quant_layers())
    packed: Dict[tuple, dict] = {}
    for k, v in tensors.items():
        parts = k.split("|")
        layer = layers[parts[0]]
        if parts[1] == "wq":
            setattr(layer.weight_quantizer, parts[2], v)
        elif parts[1] == "aq":
            setattr(layer.act_quantizer, parts[2], v)
        else:
            packed.setdefault((parts[0], int(parts[2]), int(parts[3])), {})[parts[4]] = v
    for (lname, r, nb), f in packed.items():
        layer = layers[lname]
        K = layer.weight.shape[1]
        pw = ops.PackedWeight(f["wq"], f["sw"], f["zw"], f["cs"], f["sw"].numel(), K, ops.pad128(K), nb)
        layer.install_packed(r, pw)


def broadcast_quant_state(qnn: QuantModel, rank: int, src: int = 0, group=None, packed=None):
    """One flat-buffer broadcast of grids + packed weights from ``src`` to all ranks.  ``packed`` = the (meta, arena)
    of :func:`prepack_into_arena` on ``src`` (packed in place, no copy); without it ``src`` packs now."""
    import torch.distributed as dist
    dev = next(qnn.model.parameters()).device
    if rank == src:
        meta, blob = packed if packed is not None else prepack_into_arena(qnn)
        obj = [meta, int(blob.numel())]
    else:
        obj = [None, None]
    dist.broadcast_object_list(obj, src=src, group=group)
    meta, nbytes = obj
    if rank != src:
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(blob, src=src, group=group)
    if rank != src:
        _install_quant_state(qnn, unpack_blob(meta, blob))
    qnn._packed_arena = blob        # the views installed above / packed in place live in this buffer
    return int(nbytes)


def quantize_and_distribute(model, cfg, rank: int, world: int, fp_layers=synth.REMAIN_FP,
                            force_collective: bool = False) -> QuantModel:
    """Every rank wraps its (identically seeded / loaded) fp16 model; rank 0 runs weight PTQ and packs;
    the result reaches the other ranks by ONE broadcast.  ``force_collective``: take the arena + broadcast route at
    world 1 too (a one-rank process group: how the RCCL calls are exercised on a single device)."""
    if world == 1 and not force_collective:
        qnn = synth.quantize_model(model, cfg, fp_layers)
        prepack(qnn)
        return qnn
    qnn = synth.wrap_model(model, cfg, fp_layers)
    smooth = synth.uses_smooth_quant(cfg)
    packed = None
    if rank == 0:
        if smooth:
            synth.calibrate_synthetic(qnn, cfg, fp_layers)
            synth.set_inference_state(qnn, cfg, fp_layers)
        else:
            synth.init_weight_quantizers(qnn)
            qnn.set_quant_state(True, True)
        packed = prepack_into_arena(qnn)      # packed IN the buffer that is broadcast: no second copy on rank 0
    elif smooth:
        synth.set_inference_state(qnn, cfg, fp_layers)
    else:
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
    broadcast_quant_state(qnn, rank, 0, packed=packed)
    return qnn


def gather_latents(x_local: torch.Tensor, idx_local: List[int], n_prompts: int, rank: int, world: int):
    """all_gather of the per-rank final latents back into prompt order (rank-local no-op at world 1)."""
    if world == 1:
        return x_local
    import torch.distributed as dist
    per = (n_prompts + world - 1) // world
    pad = torch.zeros((per,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[:x_local.shape[0]] = x_local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    full = torch.empty((n_prompts,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    for r in range(world):
        for j, i in enumerate(prompts_of_rank(n_prompts, r, world)):
            full[i] = outs[r][j]
    return full


def sample_sharded(qnn: QuantModel, scheduler, embeds: dict, n_prompts: int, rank: int, world: int,
                   z_size=(4, 16, 64, 64), seed: int = 42, gather: bool = True):
    """The sharded sampling job: each rank runs the full DDIM loop for its prompts, one at a time."""
    dev = next(qnn.model.parameters()).device
    mine = prompts_of_rank(n_prompts, rank, world)
    outs = []
    for i in mine:
        z = synth.synthetic_latent(i, z_size=z_size, seed=seed, device=dev)
        y = embeds["y"][i:i + 1]
        sh = y.shape
        y = y.permute(1, 0, 2, 3, 4).reshape(sh[1], sh[2], sh[3], sh[4])
        outs.append(scheduler.ddim_sample_loop(qnn, z, dict(y=y, mask=embeds["mask"][i:i + 1])))
    x_local = torch.cat(outs) if outs else torch.zeros((0,) + tuple(z_size), device=dev)
    return gather_latents(x_local, mine, n_prompts, rank, world) if gather else x_local
