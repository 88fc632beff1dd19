"""Prompt sharding over the GPUs of one node.

The reference's quantized inference is single-GPU (quant_txt2video.py:71-72).  Sampling
trajectories of different prompts are independent, so the multi-GPU form is: prompt i -> rank
i mod R, each rank runs the whole 100-step loop locally, and the only collectives are
  (1) ONE broadcast from rank 0 of the packed int weights + quant grids, as a single flat byte
      buffer (W8: ~0.75 GB; xGMI ring broadcast is per-link bound, one large message amortises
      launch latency), before the loop - preceded by a 32-byte int64 header and the layout record
      (a few hundred KB of JSON), both plain tensor broadcasts (nothing is pickled);
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


_DTYPES = {n: getattr(torch, n) for n in ("uint8", "int8", "int16", "int32", "int64", "float16", "bfloat16", "float32",
                                          "float64", "bool")}


def unpack_blob(meta: list, blob: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Zero-copy views; every record is checked against the payload it claims to describe (a corrupt or truncated layout
    record raises here instead of producing views of the wrong bytes)."""
    out = {}
    size = blob.numel()
    for name, dt, shape, off, nbytes in meta:
        if dt not in _DTYPES:
            raise RuntimeError("packed-weight arena: tensor %r has dtype %r" % (name, dt))
        n_el = 1
        for d in shape:
            n_el *= int(d)
        if off < 0 or nbytes < 0 or off + nbytes > size or n_el * _DTYPES[dt].itemsize != nbytes:
            raise RuntimeError("packed-weight arena: record of %r (%s %r, offset %d, %d bytes) does not fit a payload of %d "
                               "bytes" % (name, dt, tuple(shape), off, nbytes, size))
        out[name] = blob[off:off + nbytes].view(_DTYPES[dt]).reshape(shape)
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


def mp_plan_of(qnn: QuantModel, mp_weight_cfg) -> Tuple[Dict[str, set], set]:
    """What a timestep-wise mixed-precision weight config (``{"hi-lo": {"model.<layer>": bits}, ..., "fp_layers": {"hi-lo":
    [patterns]}}``, quant_txt2video_mp.py:533-540, applied per step by iddpm.TimestepMP) can ask of each layer over a
    whole trajectory: ({layer name: every bit width some range assigns it}, {layer names some range switches to FP}).
    Names are relative to ``qnn.model`` as in :meth:`QuantModel.quant_layers`."""
    from .qdiff.models.quant_model import pattern_in
    bits: Dict[str, set] = {}
    fp: set = set()
    if not mp_weight_cfg:
        return bits, fp
    names = [n for n, _ in qnn.quant_layers()]
    for key, table in mp_weight_cfg.items():
        if key == "fp_layers":
            for pats in table.values():
                for n in names:        # the match set_layer_quant(quant_level="per_layer") performs on "model." + n
                    if any(pattern_in("model." + n, p_) or pattern_in("model." + n, "model." + p_) for p_ in pats):
                        fp.add(n)
            continue
        for full, b in table.items():
            n = full[len("model."):] if full.startswith("model.") else full
            bits.setdefault(n, set()).add(int(b))
    return bits, fp


def _pack_jobs(qnn: QuantModel, mp_weight_cfg=None):
    """(layer name, layer, time-range id, representative timestep, n_bits) of every packed weight the hot loop will ask
    for: the layer's current bit width, plus - with a mixed-precision config - every other packable width a step range
    assigns it (the grid is the same for all of them: base_quantizer.py:126 / bitwidth_refactor widen the clamp only)."""
    jobs = []
    extra, _ = mp_plan_of(qnn, mp_weight_cfg)
    for name, layer in qnn.quant_layers():
        if layer.weight_quant and layer.act_quant and layer._can_pack():
            n_r = len(layer.timerange) if getattr(layer, "smooth_quant", False) else 1
            base = layer.weight_quantizer.n_bits
            widths = [base] + sorted(b for b in extra.get(name, ()) if b != base and b <= 8)
            for nb in widths:
                for r in range(n_r):
                    jobs.append((name, layer, r, layer.timerange[r][0] if n_r > 1 else None, nb))
    return jobs


def _pack_one(layer: QuantLayer, r: int, t_id, out=None, nb=None):
    saved = layer.cur_timestep_id
    wq = layer.weight_quantizer
    saved_bits = wq.n_bits
    if t_id is not None:
        layer.cur_timestep_id = t_id
    if nb is not None and nb != saved_bits:
        wq.bitwidth_refactor(nb)
    try:
        rr, alpha = layer._range_and_alpha()
        pw = layer.packed_weight(rr, layer.smooth_vector(rr, alpha), out=out)
    finally:
        if wq.n_bits != saved_bits:
            wq.bitwidth_refactor(saved_bits)
        layer.cur_timestep_id = saved
    return pw


def prepack(qnn: QuantModel, mp_weight_cfg=None):
    """Pack every quantized Linear for every time-range (and every bit width of a mixed-precision config) so nothing
    is packed inside the loop."""
    for _, layer, r, t_id, nb in _pack_jobs(qnn, mp_weight_cfg):
        _pack_one(layer, r, t_id, nb=nb)


_MAGIC = 0x56514152454E41   # "VQARENA"
_HDR_WORDS = 4             # int64: magic, layout version, bytes of the layout record, bytes of the whole arena
_HDR_BYTES = 8 * _HDR_WORDS


def _payload_base(rec_n: int) -> int:
    return (_HDR_BYTES + rec_n + _ALIGN - 1) // _ALIGN * _ALIGN


def _encode_meta(meta: list) -> bytes:
    import json
    return json.dumps(meta, separators=(",", ":")).encode()


def _decode_meta(raw: torch.Tensor) -> list:
    import json
    return [(n, dt, tuple(shape), off, nb) for n, dt, shape, off, nb in json.loads(raw.cpu().numpy().tobytes().decode())]


def arena_views(arena: torch.Tensor):
    """(meta, {name: zero-copy view}) of a self-describing arena: [header | layout record | tensors]."""
    if arena.numel() < _HDR_BYTES:
        raise RuntimeError("not a packed-weight arena: %d bytes" % arena.numel())
    hdr = arena[:_HDR_BYTES].view(torch.int64).tolist()
    if hdr[0] != _MAGIC or hdr[1] != 1 or hdr[3] != arena.numel() or hdr[2] < 0 or _payload_base(hdr[2]) > arena.numel():
        raise RuntimeError("not a packed-weight arena: bad header %r, %d bytes" % (hdr, arena.numel()))
    meta = _decode_meta(arena[_HDR_BYTES:_HDR_BYTES + hdr[2]])
    return meta, unpack_blob(meta, arena[_payload_base(hdr[2]):])


def prepack_into_arena(qnn: QuantModel, mp_weight_cfg=None):
    """Like :func:`prepack`, but the packed codes and per-channel terms are written straight into ONE flat byte
    buffer - the buffer that is then broadcast as is.  Rank 0 therefore never holds a second copy of the ~0.75 GB of
    packed weights (W8A8 STDiT-XL/2); the small state (grids, act scales) is copied behind them.  The buffer describes
    itself: a 32-byte int64 header (magic, version, size of the layout record, total size), the layout record (JSON:
    name, dtype, shape, offset, bytes of every tensor; offsets relative to the aligned end of the record) and the
    tensors - nothing else has to travel, and nothing is pickled.  Returns (meta, arena)."""
    from . import ops
    dev = next(qnn.model.parameters()).device
    jobs = _pack_jobs(qnn, mp_weight_cfg)
    specs = []                                            # (key, shape, dtype) in arena order
    for name, layer, r, _, nb in jobs:
        N, K = layer.weight.shape
        for f, (shape, dt) in zip(("wq", "sw", "zw", "cs"), ops.packed_shapes(N, K, nb)):
            specs.append(("%s|pw|%d|%d|%s" % (name, r, nb, f), tuple(shape), dt))
    small = _small_state(qnn)
    specs += [(k, tuple(v.shape), v.dtype) for k, v in small]
    meta, payload = _layout([(k, torch.empty(shape, dtype=dt, device="meta")) for k, shape, dt in specs])
    rec = _encode_meta(meta)
    base = _payload_base(len(rec))
    arena = torch.zeros(base + max(payload, 1), dtype=torch.uint8, device=dev)
    arena[:_HDR_BYTES] = torch.tensor([_MAGIC, 1, len(rec), arena.numel()], dtype=torch.int64).view(torch.uint8).to(dev)
    arena[_HDR_BYTES:_HDR_BYTES + len(rec)] = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    views = unpack_blob(meta, arena[base:])
    for name, layer, r, t_id, nb in jobs:
        out = [views["%s|pw|%d|%d|%s" % (name, r, nb, f)] for f in ("wq", "sw", "zw", "cs")]
        _pack_one(layer, r, t_id, out=out, nb=nb)
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


def broadcast_quant_state(qnn: QuantModel, rank: int, src: int = 0, group=None, packed=None, mp_weight_cfg=None):
    """Grids + packed weights from ``src`` to all ranks: the arena's own 32-byte header first (a fixed-size int64
    tensor - the receivers learn the size to allocate), then the arena as ONE message; the layout record sits inside
    it (:func:`prepack_into_arena`), so no Python object is pickled and nothing else travels.  ``packed`` = the
    (meta, arena) of :func:`prepack_into_arena` on ``src`` (packed in place, no copy); without it ``src`` packs now.
    Returns what travelled (bytes, seconds) for the bench line."""
    import time
    import torch.distributed as dist
    dev = next(qnn.model.parameters()).device
    hdr = torch.zeros(_HDR_WORDS, dtype=torch.int64, device=dev)
    arena = None
    if rank == src:
        _, arena = packed if packed is not None else prepack_into_arena(qnn, mp_weight_cfg)
        hdr.copy_(arena[:_HDR_BYTES].view(torch.int64))
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dist.broadcast(hdr, src=src, group=group)
    magic, ver, rec_n, total = [int(v) for v in hdr.tolist()]
    if magic != _MAGIC or ver != 1 or rec_n < 0 or total <= 0 or _payload_base(rec_n) > total:
        raise RuntimeError("broadcast_quant_state: bad header %r" % ((magic, ver, rec_n, total),))
    if rank != src:
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
    dist.broadcast(arena, src=src, group=group)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if rank != src:
        _install_quant_state(qnn, arena_views(arena)[1])
    qnn._packed_arena = arena       # the views installed above / packed in place live in this buffer
    stats = {"bytes": int(total), "seconds": dt, "messages": 2, "pickled_objects": 0}
    qnn._broadcast_stats = stats
    return stats


def release_fp_weights(qnn: QuantModel, mp_weight_cfg=None) -> int:
    """Free the fp16 master weight of every Linear whose packed integer form is installed for all its time-ranges
    (ranks > 0 of a sharded job never re-quantize: the master copy would sit beside the arena for nothing - 1.4 GB at
    STDiT-XL/2).  The smoothing vectors (they need max|W| per input channel) are derived and cached first.  Afterwards
    a re-pack (other bit width, changed statistic) raises instead of quantizing garbage.  With a mixed-precision
    config the layers that some step range runs in FP or at a width the integer route does not pack (> 8 bits) KEEP
    their master weight; every other width of the config must have arrived in packed form.  Returns the bytes freed."""
    freed = 0
    extra, keep = mp_plan_of(qnn, mp_weight_cfg)
    keep = set(keep) | {n for n, bs in extra.items() if any(b > 8 for b in bs)}
    jobs = _pack_jobs(qnn, mp_weight_cfg)
    for name, layer, r, t_id, nb in jobs:
        if (r, nb) not in layer._packed:
            raise RuntimeError("%s: no packed weight for time-range %d at %d bits arrived - releasing the master weight "
                               "would leave this rank unable to run that step range" % (name, r, nb))
    for name, layer, r, t_id, nb in jobs:
        if nb != layer.weight_quantizer.n_bits:
            continue
        saved = layer.cur_timestep_id
        if t_id is not None:
            layer.cur_timestep_id = t_id
        rr, alpha = layer._range_and_alpha()
        sv = layer.smooth_vector(rr, alpha)
        layer.packed_weight(rr, sv)                                # cache hit (installed) - and s is cached now
        layer.bias_f32()
        if name.endswith("kv_linear"):
            # the exact eps-fill route of the prompt K/V multiplies with the DEQUANTIZED weight (t2v/stdit.py): derive
            # and cache it while the master copy (or, for int8 codes, the packed form) can still give it
            layer.dequantized_weight_f16(rr, sv)
        layer.cur_timestep_id = saved
    done = set()
    for name, layer, r, _, _nb in jobs:
        if id(layer) in done or not layer.int_route_ok() or name in keep:
            continue
        done.add(id(layer))
        w = layer.weight
        freed += w.numel() * w.element_size()
        layer.released_shape = tuple(w.shape)
        w.data = torch.empty(0, dtype=w.dtype, device=w.device)     # keeps the Parameter object and its version counter
    return freed


def quantize_and_distribute(model, cfg, rank: int, world: int, fp_layers=synth.REMAIN_FP,
                            force_collective: bool = False, release_fp: bool = True, mp_weight_cfg=None) -> QuantModel:
    """Every rank wraps its (identically seeded / loaded) fp16 model; rank 0 runs weight PTQ and packs;
    the result reaches the other ranks by ONE broadcast.  ``force_collective``: take the arena + broadcast route at
    world 1 too (a one-rank process group: how the RCCL calls are exercised on a single device).  ``release_fp``: ranks
    other than 0 drop the fp16 master weights of the Linears they received in packed form (:func:`release_fp_weights`).
    ``mp_weight_cfg``: the timestep-wise mixed-precision weight config the trajectory will switch through
    (iddpm.TimestepMP) - every bit width it names travels in the arena and the layers it runs in FP keep their master
    weight, so no rank ever has to re-pack; a job that will switch bit widths WITHOUT naming them here must pass
    ``release_fp=False``."""
    if world == 1 and not force_collective:
        qnn = synth.quantize_model(model, cfg, fp_layers)
        prepack(qnn, mp_weight_cfg)
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
        packed = prepack_into_arena(qnn, mp_weight_cfg)   # packed IN the buffer that is broadcast: no second copy on rank 0
    elif smooth:
        synth.set_inference_state(qnn, cfg, fp_layers)
    else:
        qnn.set_quant_init_done("weight")
        qnn.set_quant_init_done("activation")
        qnn.set_quant_state(True, True)
    broadcast_quant_state(qnn, rank, 0, packed=packed)
    if rank != 0 and release_fp:
        qnn._released_bytes = release_fp_weights(qnn, mp_weight_cfg)
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
