"""Multi-process CPU tests (gloo, world_size 2) of the prompt-sharding path: the partition, the
single flat-buffer broadcast of quant grids + packed weights, and the gather of final latents.
On GPUs the same code runs over RCCL (backend 'nccl'); the logic is backend independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


TINY = dict(input_size=(4, 8, 8), depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)


def _install_cpu_ops(counters):
    """The ONLY mocked boundary: the weight-side HIP entry points (vq_weight_minmax, vq_pack_weight, the
    calibrated-grid form of vq_fakequant_act), restated with the oracle on CPU tensors - they need a GPU.  Everything above them - quantizer classes, QuantLayer caches,
    QuantModel state, synth / shard control flow, the arena, the collectives - is the product code."""
    from oracle import fakequant as fq
    from viditq_amd import ops

    def weight_minmax(W, n_bits, s=None, force_eps=False, status=None):
        Wf = W.float() if s is None else W.float() * s.reshape(1, -1)
        d, z, eps = fq.minmax_params(Wf.reshape(Wf.shape[0], -1), n_bits)
        if eps and status is not None:
            status |= 1
        return d, z

    def pack_weight(W, delta, zp, n_bits, s=None, out=None):
        counters["pack"] = counters.get("pack", 0) + 1
        N, K = W.shape
        Kp = ops.pad128(K)
        Wf = W.float() if s is None else W.float() * s.reshape(1, -1)
        codes = fq.quant_codes(Wf, delta.reshape(-1, 1), zp.reshape(-1, 1), n_bits)
        cx = 128 if n_bits == 8 else 0
        if out is None:
            out = [torch.empty(sh, dtype=dt) for sh, dt in ops.packed_shapes(N, K, n_bits)]
        wq, sw, zw, cs = out
        if n_bits <= 4:
            c = torch.zeros(N, Kp, dtype=torch.uint8)
            c[:, :K] = codes.to(torch.uint8)
            c4 = c.reshape(N, Kp // 8, 2, 4)                      # stand-in nibble order (layout is the kernel's business)
            wq.copy_((c4[:, :, 0] | (c4[:, :, 1] << 4)).reshape(N, Kp // 2))
        else:
            wq.zero_()
            wq[:, :K] = (codes - cx).to(torch.int8)
        sw.copy_(delta.reshape(-1))
        zw.copy_((zp.reshape(-1) - cx).to(torch.int32))
        cs.copy_(((codes - cx).sum(dim=1) - K * (zp.reshape(-1) - cx)).to(torch.int32))
        return ops.PackedWeight(wq, sw, zw, cs, N, K, Kp, n_bits)

    def new_status(device):
        return torch.zeros(1, dtype=torch.int32)

    def fakequant_act(x, n_bits=8, delta=None, zp=None, status=None, want_codes=False):
        assert delta is not None, "only the calibrated-grid form is used on weights"
        n = delta.numel()
        d = delta.reshape(1, n, 1) if n > 1 else delta.reshape(1, 1, 1)
        z = zp.reshape(1, n, 1) if n > 1 else zp.reshape(1, 1, 1)
        codes = fq.quant_codes(x.float(), d, z, n_bits)
        return fq.dequant(codes, d, z).half(), None, delta, zp

    ops.weight_minmax, ops.pack_weight, ops.new_status, ops.fakequant_act = weight_minmax, pack_weight, new_status, fakequant_act


def _fake_calibration(qnn, cfg, fp_layers=None, seed=7):
    """Stand-in for synth.calibrate_synthetic on CPU (it samples an FP DDIM trajectory through the HIP attention
    kernels): the same END STATE - a momentum act-scale statistic per layer and time-range, weight grids of W*s for
    every range from the product's own WeightQuantizer - from seeded random statistics."""
    g = torch.Generator().manual_seed(seed)
    qnn.set_smooth_quant(smooth_quant=True, smooth_quant_running_stat=False)
    qnn.set_layer_smooth_quant(model=qnn, module_name_list=list(fp_layers), smooth_quant=False,
                               smooth_quant_running_stat=False)
    for name, layer in qnn.quant_layers():
        if not layer.smooth_quant:
            layer.weight_quantizer(layer.weight.detach().half())
            continue
        K = layer.weight.shape[1]
        layer.act_quantizer.act_scale = torch.rand(len(layer.timerange), 1, K, generator=g) + 0.5
        wq = layer.weight_quantizer
        wq.timestep_wise, wq.n_timestep = True, len(layer.timerange)
        for r in range(len(layer.timerange)):
            wq.cur_timestep_id = r
            s = layer.channel_wise_scale(r, layer._alpha_of(r)).reshape(-1)
            for nb in wq.mixed_precision:
                wq.init_quant_params(layer.weight.detach().half(), wq.per_group, n_bits=nb, smooth=s)
        wq.delta, wq.zero_point = wq.delta_list[wq.bit_idx, 0], wq.zero_point_list[wq.bit_idx, 0]
    return qnn


def _worker(rank, world, port, ret, plan):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import viditq_amd  # noqa
        from viditq_amd import shard, synth
        from viditq_amd.config import loads_yaml
        from viditq_amd.t2v import STDiT
        counters = {}
        _install_cpu_ops(counters)
        synth.calibrate_synthetic = _fake_calibration
        torch.manual_seed(0)                                     # every rank builds the same fp16 model
        m = STDiT(dtype=torch.float16, **TINY)
        synth.redraw_zero_init(m, 1)
        m = m.half().eval()
        cfg = loads_yaml(synth.W8A8_DYNAMIC if plan == "w8a8" else synth.W4A8_TIMESTEP_AWARE)
        cfg.quant.activation.quantizer["n_spatial_token"], cfg.quant.activation.quantizer["n_temporal_token"] = 16, 4
        if plan == "w4a8_mp":
            return _mp_case(m, cfg, rank, world, counters, ret)
        qnn = shard.quantize_and_distribute(m, cfg, rank, world)              # the REAL control flow, both branches
        n_ranges = 1 if plan == "w8a8" else 2
        bits = 8 if plan == "w8a8" else 4
        hot = [(n, l) for n, l in qnn.quant_layers() if n.startswith("blocks.")]
        assert len(hot) == 2 * 13
        if rank != 0:
            assert counters.get("pack", 0) == 0                  # nothing was packed here: everything arrived
        packs_before = counters.get("pack", 0)
        acc = torch.zeros(1, dtype=torch.float64)
        arena = qnn._packed_arena
        lo, hi = arena.data_ptr(), arena.data_ptr() + arena.numel()
        for name, layer in sorted(hot):
            assert layer.get_quant_state() == (True, True) and layer.int_route_ok()
            assert layer.weight_quantizer.init_done and layer.act_quantizer.init_done
            for r in range(n_ranges):
                layer.cur_timestep_id = 0 if r == 0 else 600
                rr, alpha = layer._range_and_alpha()
                assert rr == r
                pw = layer.packed_weight(rr, layer.smooth_vector(rr, alpha))   # what the hot loop asks for
                assert pw.n_bits == bits and pw.wq.dtype == (torch.int8 if bits == 8 else torch.uint8)
                for t_ in pw.tensors():                          # zero-copy: views of the ONE broadcast buffer, on rank 0 too
                    assert lo <= t_.data_ptr() < hi
                acc += sum(t_.double().sum() for t_ in pw.tensors())
            acc += layer.weight_quantizer.delta_list.double().sum() + layer.weight_quantizer.zero_point_list.double().sum()
            if plan != "w8a8":
                acc += layer.act_quantizer.act_scale.double().sum()
        assert counters.get("pack", 0) == packs_before           # ... and asking again packs nothing, on any rank
        for n, l in qnn.quant_layers():                          # the FP list stays FP everywhere
            if not n.startswith("blocks."):
                assert l.get_quant_state() == (False, False)
        both = [torch.zeros_like(acc) for _ in range(world)]
        dist.all_gather(both, acc)
        assert torch.equal(both[0], both[1]) and float(acc) != 0.0
        # the set-up traffic: the arena's 32-byte header, then the arena itself; no pickled object
        st = qnn._broadcast_stats
        assert st["messages"] == 2 and st["pickled_objects"] == 0 and st["bytes"] == arena.numel()
        meta2, views2 = shard.arena_views(arena)                 # the buffer describes itself
        assert len(meta2) == len(views2) and any(k.endswith("|pw|0|%d|wq" % bits) for k in views2)
        # ranks other than 0 hold no fp16 master copy of what they received in packed form; FP layers keep theirs
        w_bytes = sum(l.weight.numel() * l.weight.element_size() for _, l in hot)
        if rank == 0:
            assert all(l.weight.numel() > 0 for _, l in hot)
        else:
            assert w_bytes == 0 and qnn._released_bytes > 0
            for n, l in qnn.quant_layers():
                if not n.startswith("blocks."):
                    assert l.weight.numel() > 0
            hot[0][1].invalidate_packed()                        # ... and a re-pack here is refused, not faked
            try:
                hot[0][1].packed_weight(0)
                raise AssertionError("re-pack of a released weight must raise")
            except RuntimeError as e:
                assert "released" in str(e)
        # prepack() FIRST, then a broadcast that packs "now" (packed=None): the cached entries must be copied into the
        # arena, not returned untouched (an arena of zeros used to be shipped)
        torch.manual_seed(0)
        m2 = STDiT(dtype=torch.float16, **TINY)
        synth.redraw_zero_init(m2, 1)
        m2 = m2.half().eval()
        if rank == 0:
            q2 = synth.quantize_model(m2, cfg)
            shard.prepack(q2)
        else:
            q2 = synth.wrap_model(m2, cfg)
            if synth.uses_smooth_quant(cfg):
                synth.set_inference_state(q2, cfg, synth.REMAIN_FP)
            else:
                q2.set_quant_init_done("weight")
                q2.set_quant_init_done("activation")
                q2.set_quant_state(True, True)
        shard.broadcast_quant_state(q2, rank, 0)
        acc2 = torch.zeros(1, dtype=torch.float64)
        a2 = q2._packed_arena
        for name, layer in sorted((n, l) for n, l in q2.quant_layers() if n.startswith("blocks.")):
            for r in range(n_ranges):
                layer.cur_timestep_id = 0 if r == 0 else 600
                rr, alpha = layer._range_and_alpha()
                pw = layer.packed_weight(rr, layer.smooth_vector(rr, alpha))
                assert pw.wq.abs().sum() > 0 and pw.sw.abs().sum() > 0
                for t_ in pw.tensors():
                    assert a2.data_ptr() <= t_.data_ptr() < a2.data_ptr() + a2.numel()
                acc2 += sum(t_.double().sum() for t_ in pw.tensors())
        both2 = [torch.zeros_like(acc2) for _ in range(world)]
        dist.all_gather(both2, acc2)
        assert torch.equal(both2[0], both2[1]) and float(acc2) != 0.0
        # partition + gather: 5 prompts over 2 ranks, round robin
        n_prompts = 5
        mine = shard.prompts_of_rank(n_prompts, rank, world)
        x_local = torch.stack([torch.full((2, 3), float(i)) for i in mine])
        full = shard.gather_latents(x_local, mine, n_prompts, rank, world)
        assert torch.equal(full[:, 0, 0], torch.arange(n_prompts, dtype=torch.float32))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _mp_case(m, cfg, rank, world, counters, ret):
    """The mixed-precision plan AFTER quantize_and_distribute (bench.py --plan w4a8_mp --gpus N): per-step-range bit
    widths and FP layers are switched by iddpm.TimestepMP on every rank.  Ranks > 0 have released their fp16 master
    weights, so every width the config names must have travelled in the arena and the layers it runs in FP must have
    kept their master copy (round-3 advisor finding: rank 1 raised 'master weight was released' and rank 0 hung)."""
    from viditq_amd import ptq, shard, synth
    from viditq_amd.t2v.iddpm import TimestepMP
    w_cfg, a_cfg = synth.synthetic_mp_config(m, 20)
    keys = [k for k in w_cfg if k != "fp_layers"]
    w_cfg[keys[1]]["model.blocks.0.attn.proj"] = 8                 # a width that differs between step ranges
    w_cfg[keys[2]]["model.blocks.1.mlp.fc1"] = 6
    w_cfg["fp_layers"][keys[3]] = ["blocks.1.cross_attn.proj"]      # ... and a layer one range runs in FP
    qnn = shard.quantize_and_distribute(m, cfg, rank, world, mp_weight_cfg=w_cfg)
    packs_before = counters.get("pack", 0)
    if rank != 0:
        assert packs_before == 0 and qnn._released_bytes > 0
    ptq.enable_timestep_wise_mp(qnn, w_cfg, a_cfg)
    mp_ = TimestepMP(qnn)
    layers = dict(qnn.quant_layers())
    acc = torch.zeros(1, dtype=torch.float64)
    seen = set()
    for i in range(19, -1, -1):                                     # the 20-step schedule, high to low
        key = mp_.apply(i)
        seen.add(key)
        for name, layer in sorted(layers.items()):
            if not name.startswith("blocks."):
                continue
            want = w_cfg[key]["model." + name]
            assert layer.weight_quantizer.n_bits == want
            for t_id in (0, 600):
                layer.cur_timestep_id = t_id
                rr, alpha = layer._range_and_alpha()
                if layer.get_quant_state() == (False, False):       # FP in this range: needs the master weight
                    assert name == "blocks.1.cross_attn.proj" and key == keys[3]
                    assert layer._master_weight().numel() > 0
                    continue
                pw = layer.packed_weight(rr, layer.smooth_vector(rr, alpha))
                assert pw.n_bits == want
                acc += sum(t_.double().sum() for t_ in pw.tensors())
    assert seen == set(keys)
    assert counters.get("pack", 0) == packs_before                  # nothing was packed inside the loop, on any rank
    if rank != 0:                                                   # released everywhere else
        assert layers["blocks.0.attn.proj"].weight.numel() == 0 and layers["blocks.1.cross_attn.proj"].weight.numel() > 0
    both = [torch.zeros_like(acc) for _ in range(world)]
    dist.all_gather(both, acc)
    assert torch.equal(both[0], both[1]) and float(acc) != 0.0
    # a job that switches to a width it did NOT name fails loudly on the ranks that cannot re-pack
    layers["blocks.0.attn.q"].weight_quantizer.bitwidth_refactor(6)
    if rank != 0:
        try:
            layers["blocks.0.attn.q"].packed_weight(0)
            raise AssertionError("re-pack of a released weight must raise")
        except RuntimeError as e:
            assert "released" in str(e)
    ret[rank] = "ok"


def test_arena_header_and_records_are_validated():
    import viditq_amd  # noqa
    from viditq_amd import shard
    with pytest.raises(RuntimeError, match="arena"):
        shard.arena_views(torch.zeros(8, dtype=torch.uint8))
    bad = torch.zeros(512, dtype=torch.uint8)
    bad[:32] = torch.tensor([shard._MAGIC, 1, 10 ** 9, 512], dtype=torch.int64).view(torch.uint8)   # record longer than the arena
    with pytest.raises(RuntimeError, match="bad header"):
        shard.arena_views(bad)
    blob = torch.zeros(64, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="dtype"):
        shard.unpack_blob([("a", "complex_nonsense", (4,), 0, 4)], blob)
    with pytest.raises(RuntimeError, match="does not fit"):
        shard.unpack_blob([("a", "int32", (32,), 0, 128)], blob)          # past the end of the payload
    with pytest.raises(RuntimeError, match="does not fit"):
        shard.unpack_blob([("a", "int32", (4,), 0, 8)], blob)             # shape and byte count disagree
    assert shard.unpack_blob([("a", "int32", (4,), 16, 16)], blob)["a"].shape == (4,)


def test_partition_is_a_bijection():
    import viditq_amd  # noqa
    from viditq_amd import shard
    for n, w in ((64, 8), (5, 2), (3, 4), (1, 1)):
        parts = [shard.prompts_of_rank(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_blob_roundtrip_is_zero_copy_and_aligned():
    import viditq_amd  # noqa
    from viditq_amd import shard
    ts = [("a", torch.arange(7, dtype=torch.int8)), ("b", torch.rand(3, 5)), ("c", torch.arange(4, dtype=torch.int32))]
    meta, blob = shard.pack_blob(ts, "cpu")
    out = shard.unpack_blob(meta, blob)
    for name, t in ts:
        assert torch.equal(out[name], t) and out[name].dtype == t.dtype
        assert out[name].data_ptr() >= blob.data_ptr() and (out[name].data_ptr() - blob.data_ptr()) % 256 == 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("plan", ["w8a8", "w4a8", "w4a8_mp"])
def test_quantize_and_distribute_world2_gloo(plan):
    """shard.quantize_and_distribute as bench.py calls it, on two gloo ranks: rank 0 runs weight PTQ (dynamic plan) or
    calibration (smooth-quant plan, two time-ranges), packs INTO the arena and broadcasts it once; rank 1 only sets the
    inference state and installs views of what arrived.  Checks: both control-flow branches on both ranks, identical
    grids / packed weights / act scales (checksum of checksums), zero packing work off rank 0, zero-copy views."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret, plan), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
