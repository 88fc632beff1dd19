"""GPU parity of the plans and model families round 1 left untested, and the achieved-vs-asserted record.

* static activation plans (w8a8_naive / *_ptqd: calibrated tensor-wise grids; the static per-token variant with
  the zero-mask prompt path), PTQD with k != 0, the act-grid pass of the PTQ producer;
* the PixArt-alpha net (BASELINE config 1) and PixArt-MS W4A8 with the running smooth-quant statistic (config 5);
* full-size parity: one STDiT-XL/2 block at 16 x 1024 tokens (W8A8 and W4A8 timestep-aware) and one PixArt-XL/2 block at
  4096 tokens / 300-token prompts / B = 2, each against the CPU oracle on identical weights;
* every model-level comparison is made against the reference's fp32 output AND against the reference's own fp16-mode
  output (what it computes on a GPU), and the achieved rel-L2 values go to gpurun_out/parity.json.

Tolerances.  north_star: 1e-3 rel-L2.  Per LAYER that bound is asserted (test_model_gpu.py).  Above the layer level the
reference itself, run the way its scripts run it (fp16 model and buffers), deviates from its fp32 result by
1.0e-3 (block 0) ... 1.9e-3 (tiny model) ... 3.1e-3 (W4A8 tiny model): fp16 storage between layers flips
quantization codes downstream.  The HIP path stores fp16 between kernels too, so the bound asserted here is
"within 1.25 x the reference's own fp16-mode deviation from fp32" (and in absolute terms the figures recorded in
profiles/r02_parity.json).
"""
import json

import numpy as np
import pytest
import torch

from helpers import FP_LAYERS, grids_of, load_npz, quant_params_of, rel_l2, state_dict_of

pytestmark = pytest.mark.gpu

TINY = dict(input_size=(4, 8, 8), depth=2, hidden_size=64, num_heads=4, model_max_length=12, caption_channels=32)
PIX_FP = ["x_embedder", "t_embedder", "t_block", "y_embedder", "csize_embedder", "ar_embedder"]


def _cfgs(w_bits, dynamic=True, per_group="token", smooth=None, mixed_precision=None, T=4, S=16, running_stat=False):
    from viditq_amd.config import to_config
    wq = dict(n_bits=w_bits, per_group="channel", channel_dim=0, scale_method="min_max", round_mode="nearest")
    if mixed_precision:
        wq["mixed_precision"] = mixed_precision
    sq = dict(enable=False)
    if smooth:
        sq = dict(enable=True, channel_wise_scale_type="momentum_act_max", momentum=0.95, **smooth)
    aq = dict(n_bits=8, per_group=per_group, scale_method="min_max", round_mode="nearest_ste", running_stat=running_stat,
              dynamic=dynamic, sym=False, n_spatial_token=S, n_temporal_token=T, n_prompt=12, smooth_quant=sq)
    return to_config(wq), to_config(aq)


def _load_qp(qnn, qp):
    from viditq_amd.qdiff.quantizer import BaseQuantizer
    full = {mod.module_name: [qp.get(mod.module_name, {}), {}] for mod in qnn.model.modules()
            if isinstance(mod, BaseQuantizer)}
    qnn.set_quant_params_dict(full)
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")
    qnn.set_quant_state(True, True)


def _stdit(gold, dev, wq, aq, cfg_split, fp=FP_LAYERS):
    import viditq_amd  # noqa
    from viditq_amd.qdiff.models import QuantModel
    from viditq_amd.t2v import STDiT
    m = STDiT(dtype=torch.float16, **TINY)
    m.load_state_dict(state_dict_of(gold), strict=True)
    qnn = QuantModel(m.half().to(dev).eval(), wq, aq)
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = list(fp)
    qnn.cfg_split = cfg_split
    return qnn


_FLOOR = {}


def parity_floor(name):
    """The MEASURED floor at a full-size checkpoint: rel-L2 of the fp32 oracle from the reference's fp32 mode on the same
    forward (tools/parity_floor.py -> profiles/rNN_parity_floor.json, CPU, committed; the newest one is read) - what two fp32 implementations of
    the same arithmetic are apart there.  None for places without a measurement."""
    if not _FLOOR:
        import json
        import os
        from helpers import parity_floor_file
        f = parity_floor_file()                        # the newest committed record (round 6: every full-size assert has one)
        _FLOOR["_"] = None
        if f and os.path.exists(f):
            with open(f) as fh:
                _FLOOR.update({k: v["oracle_fp32_vs_ref_fp32"] for k, v in json.load(fh)["records"].items()})
    return _FLOOR.get(name)


def within_floor_bound(ent):
    """DESIGN section 2 as restated in round 5: the HIP path is within max(1.25e-3, 1.75 x the measured fp32-vs-fp32 floor)
    of the reference's fp32 mode (1.25e-3 = north_star's 1e-3 + the fp16 storage of one block's output; beyond depth ~2
    the floor term takes over)."""
    fl = ent.get("floor_fp32_oracle_vs_ref_fp32")
    return fl is None or ent["vs_ref_fp32"] < max(1.25e-3, 1.75 * fl)


def _rec(parity, name, got, ref32, ref16=None, **extra):
    """achieved rel-L2 vs the reference's fp32 output, vs its fp16-mode output, and the reference's own fp16 deviation
    (+ the measured fp32-vs-fp32 floor of the place, where tools/parity_floor.py recorded one)"""
    ent = {"vs_ref_fp32": rel_l2(got, ref32)}
    if ref16 is not None:
        ent["vs_ref_fp16"] = rel_l2(got, ref16)
        ent["ref_fp16_vs_ref_fp32"] = rel_l2(ref16, ref32)
    fl = parity_floor(name)
    if fl is not None:
        ent["floor_fp32_oracle_vs_ref_fp32"] = fl
        ent["over_floor"] = ent["vs_ref_fp32"] / fl if fl > 0 else None
    ent.update(extra)
    parity[name] = ent
    return ent


# ----------------------------------------------------------------------------- yardstick: the reference's fp16 mode
def test_tiny_stdit_against_fp32_and_fp16_mode_goldens(dev, ops, parity):
    """Every block output, the three forwards and the 3-step DDIM trajectory of the W8A8 tiny model; the W4A8
    two-range / mixed-precision forwards and the 4-step MP trajectory: HIP path vs the reference in fp32 AND in its own
    fp16 mode.  Asserted: not further from the fp32 result than 1.25 x the reference's fp16 mode is (+1e-4)."""
    from test_model_gpu import _build
    from viditq_amd.t2v import IDDPM
    import viditq_amd.t2v.stdit as st
    g, g16 = load_npz("tiny_stdit_w8a8.npz"), load_npz("tiny_stdit_fp16ref.npz")
    qnn = _build(g, dev, 8)
    x, y, mask, t = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev), g["t"].to(dev)
    blocks = []
    orig = st.STDiTBlock.forward_fused

    def spy(self, x2, *a, **k):
        r = orig(self, x2, *a, **k)
        blocks.append(x2.clone())
        return r
    st.STDiTBlock.forward_fused = spy
    try:
        cond = qnn(x, t, y[:1], mask=mask).cpu()
    finally:
        st.STDiTBlock.forward_fused = orig
    cases = [("w8a8_block%d" % i, b.cpu().float().reshape(1, 64, 64)) for i, b in enumerate(blocks)]
    cases += [("w8a8_cond", cond), ("w8a8_uncond", qnn(x, t, y[1:], mask=mask).cpu()),
              ("w8a8_joint", qnn(torch.cat([x, x]), torch.cat([t, t]), y, mask=mask).cpu())]
    sch = IDDPM(num_sampling_steps=3, cfg_scale=4.0)
    cases.append(("ddim_final", sch.ddim_sample_loop(qnn, g["ddim_z"].to(dev), dict(y=y, mask=mask)).cpu()))
    for name, got in cases:
        e = _rec(parity, "tiny_stdit/" + name, got, g[name], g16[name + "_ref_fp16"])
        assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, (name, e)
    # W4A8, two smooth-quant time-ranges, mixed precision
    g = load_npz("tiny_stdit_w4a8.npz")
    qnn = _build(g, dev, 4, smooth=dict(alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]]), mixed_precision=[4, 6, 8])
    qnn.set_layer_smooth_quant(model=qnn, module_name_list=FP_LAYERS, smooth_quant=False, smooth_quant_running_stat=False)
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    for tv in (721, 300):
        got = qnn(x, torch.tensor([tv], device=dev), y[:1], mask=mask).cpu()
        e = _rec(parity, "tiny_stdit/w4a8_cond_t%d" % tv, got, g["w4a8_cond_t%d" % tv], g16["w4a8_cond_t%d_ref_fp16" % tv])
        assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e
    qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 8, "model.blocks.1.attn.q": 8}, "weight")
    got = qnn(x, torch.tensor([721], device=dev), y[:1], mask=mask).cpu()
    e = _rec(parity, "tiny_stdit/w4a8_mp_cond_t721", got, g["w4a8_mp_cond_t721"], g16["w4a8_mp_cond_t721_ref_fp16"])
    assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e
    qnn.load_bitwidth_config(qnn, {"model.blocks.0.mlp.fc1": 4, "model.blocks.1.attn.q": 4}, "weight")
    qnn.timestep_wise_mp = True
    qnn.time_mp_config_weight = json.loads(g["mp_weight_cfg_json"])
    qnn.time_mp_config_act = json.loads(g["mp_act_cfg_json"])
    sch = IDDPM(num_sampling_steps=4, cfg_scale=4.0)
    got = sch.ddim_sample_loop(qnn, g["mp_ddim_z"].to(dev), dict(y=g["mp_ddim_y"].half().to(dev), mask=mask)).cpu()
    e = _rec(parity, "tiny_stdit/mp_ddim_final", got, g["mp_ddim_final"], g16["mp_ddim_final_ref_fp16"])
    # a trajectory's deviation is dominated by WHICH codes flip, not how many: 2 x the reference's own figure
    assert e["vs_ref_fp32"] < 2.0 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e


# ----------------------------------------------------------------------------- static activation plans (A3)
@pytest.mark.parametrize("tag,per_group", [("tw", False), ("tk", "token")])
def test_static_activation_plans_match_reference(dev, ops, parity, tag, per_group):
    """w8a8_naive / *_ptqd (``per_group: False, dynamic: False``, cfg_split False): the calibrated grids of the
    reference, loaded through the ckpt.pth schema, must be the grids the HIP path quantizes on - on the fused route
    (tensor-wise) and, for static per-token grids, on the layer-by-layer route with the zero-mask prompt path
    (stdit.py:272-301) and the [B, n_prompt, C] kv view (stdit_quant_layer.py:272-278)."""
    g = load_npz("tiny_stdit_static.npz")
    wq, aq = _cfgs(8, dynamic=False, per_group=per_group, mixed_precision=[4, 6, 8])
    qnn = _stdit(g, dev, wq, aq, cfg_split=False)
    _load_qp(qnn, quant_params_of(g, "qp_" + tag))
    fused = all(b.fused_ok() for b in qnn.model.blocks)
    assert fused == (tag == "tw")
    assert qnn.model._mask_select() == (tag == "tw")
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    t = torch.tensor([721, 721], device=dev)
    joint = qnn(torch.cat([x, x]), t, y, mask=mask).cpu()
    e = _rec(parity, "tiny_stdit_static/%s_joint_t721" % tag, joint, g[tag + "_joint_t721"], g[tag + "_joint_t721_ref_fp16"])
    assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e
    cond = qnn(x, torch.tensor([300], device=dev), y[:1], mask=mask).cpu()
    e = _rec(parity, "tiny_stdit_static/%s_cond_t300" % tag, cond, g[tag + "_cond_t300"])
    assert e["vs_ref_fp32"] < 2.0e-3, e          # recorded 1.59e-3 (tw) / 1.32e-3 (tk)
    if tag == "tw":
        # a dynamic grid silently substituted for the calibrated one is a DIFFERENT result
        # (measured: 2.6e-3 from the reference against 1.6e-3 on the calibrated grid)
        wq_d, aq_d = _cfgs(8, mixed_precision=[4, 6, 8])
        qd = _stdit(g, dev, wq_d, aq_d, cfg_split=False)
        _load_qp(qd, {k: v for k, v in quant_params_of(g, "qp_tw").items() if k.endswith("weight_quantizer")})
        dyn = qd(torch.cat([x, x]), t, y, mask=mask).cpu()
        assert rel_l2(dyn, joint) > 1e-3
        assert rel_l2(dyn, g["tw_joint_t721"]) > 1.3 * rel_l2(joint, g["tw_joint_t721"])


def test_ptqd_ddim_with_nonzero_k_matches_reference(dev, ops, parity):
    """PTQD correlated-noise correction (iddpm/__init__.py:168-172): every model output divided by
    1 + ks[(999 - t) // 50] with a NON-zero table, 3 guided DDIM steps of the static tensor-wise plan."""
    from viditq_amd.t2v import IDDPM
    g = load_npz("tiny_stdit_static.npz")
    wq, aq = _cfgs(8, dynamic=False, per_group=False, mixed_precision=[4, 6, 8])
    qnn = _stdit(g, dev, wq, aq, cfg_split=False)
    _load_qp(qnn, quant_params_of(g, "qp_tw"))
    sch = IDDPM(num_sampling_steps=3, cfg_scale=4.0)
    assert sch.timestep_map == [int(v) for v in g["tw_ptqd_timestep_map"]]
    y, mask = g["y"].half().to(dev), g["mask"].to(dev)
    out = sch.ddim_sample_loop(qnn, g["ddim_z"].to(dev), dict(y=y, mask=mask), ks=g["ks"]).cpu()
    e = _rec(parity, "tiny_stdit_static/tw_ptqd_ddim_final", out, g["tw_ptqd_ddim_final"], g["tw_ptqd_ddim_final_ref_fp16"])
    assert e["vs_ref_fp32"] < 2.0 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e
    out0 = sch.ddim_sample_loop(qnn, g["ddim_z"].to(dev), dict(y=y, mask=mask)).cpu()      # k = 0 is a different result
    assert rel_l2(out0, g["tw_ptqd_ddim_final"]) > 3 * e["vs_ref_fp32"]                  # 3.6e-3 vs 7.8e-4


@pytest.mark.parametrize("tag,per_group", [("tw", False), ("tk", "token")])
def test_ptq_calibrate_static_activation_grids(dev, ops, tag, per_group):
    """Pass 3 of ptq.calibrate (t2v/scripts/ptq.py:296-318): static activation grids re-initialised by every
    calibration batch, the last one stays - against the grids the reference classes produced on the same batches."""
    from viditq_amd import ptq
    from viditq_amd.config import to_config
    g = load_npz("tiny_stdit_static.npz")
    wq, aq = _cfgs(8, dynamic=False, per_group=per_group, mixed_precision=[4, 6, 8])
    qnn = _stdit(g, dev, wq, aq, cfg_split=False)
    del qnn.fp_layer_list
    cfg = to_config({"calib_data": {"n_samples": 1, "batch_size": 1, "n_steps": 3},
                     "quant": {"weight": {"quantizer": wq}, "activation": {"quantizer": aq}}})
    data = (g["calib_xs"], g["calib_ts"], g["calib_cs"].half(), g["calib_masks"])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the static per-token kv grid hits the reference's eps fill (expected)
        qd = ptq.calibrate(qnn, cfg, data, fp_layer_list=FP_LAYERS)
    ref = quant_params_of(g, "qp_" + tag)
    n = 0
    for name, (bufs, _) in qd.items():
        if not name.startswith("blocks") or not name.endswith("act_quantizer"):
            continue
        a, b = bufs["delta"].float().cpu(), ref[name]["delta"].float()
        # fp16 activations vs the reference's fp32: min / max of an activation tensor agree to fp16 rounding of the
        # extreme element plus upstream code flips
        assert torch.allclose(a.reshape(b.shape), b, rtol=2e-2, atol=1e-7), (name, float((a.reshape(b.shape) - b).abs().max()))
        za, zb = bufs["zero_point"].float().cpu(), ref[name]["zero_point"].float()
        # round(-min / delta): a few codes normally; under the eps fill (delta = 1e-6) the zero point is ~5e4 and moves
        # with the fp16 rounding of the minimum
        assert ((za.reshape(zb.shape) - zb).abs() <= torch.clamp(1e-3 * zb.abs(), min=3)).all(), name
        n += 1
    assert n == 2 * 13
    if per_group == "token":
        assert torch.all(qd["blocks.0.cross_attn.kv_linear.act_quantizer"][0]["delta"] == 1e-6)   # global eps fill
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    out = qnn(torch.cat([x, x]), torch.tensor([721, 721], device=dev), y, mask=mask).cpu()
    assert rel_l2(out, g[tag + "_joint_t721"]) < 2e-2      # own calibration: grids differ in the last bits -> other codes


# ----------------------------------------------------------------------------- PixArt-alpha (config 1)
def _pixart(cls_name, gold, dev, wq, aq):
    import viditq_amd  # noqa
    from viditq_amd import t2i
    from viditq_amd.qdiff.models import QuantModel
    m = getattr(t2i, cls_name)(input_size=16, depth=2, hidden_size=64, num_heads=4, model_max_length=12,
                               caption_channels=32, dtype=torch.float16)
    m.load_state_dict(state_dict_of(gold), strict=True)
    qnn = QuantModel(m.half().to(dev).eval(), wq, aq, model_type="pixart")
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = list(PIX_FP)
    return qnn


def test_pixart_alpha_net_matches_reference(dev, ops, parity):
    """BASELINE config 1 (PixArt-alpha 256^2 W8A8): ``PixArt`` is the alpha class of PixArt.py:63-256 - fixed
    pos_embed buffer, PixArtBlock - not an alias of the multi-scale net.  FP, W8A8 dynamic (B = 2 shared scales and
    the single-prompt B = 1 case on the fused route), the static 'naive' plan, and a DPM-Solver++ trajectory through
    the alpha entry point."""
    from viditq_amd import t2i
    from viditq_amd.t2i.dpm_solver import DPMS_alpha
    assert t2i.PixArt is not t2i.PixArtMS and not issubclass(t2i.PixArt, t2i.PixArtMS)
    g = load_npz("tiny_pixart_alpha.npz")
    wq, aq = _cfgs(8, T=1, S=64)
    qnn = _pixart("PixArt", g, dev, wq, aq)
    assert type(qnn.model.blocks[0]).__name__ == "PixArtBlock"
    assert torch.equal(qnn.model.pos_embed.cpu().float(), state_dict_of(g)["pos_embed"])
    x, y, mask, t = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev), g["t"].to(dev)
    assert rel_l2(qnn(x, t, y, mask=mask).cpu().float(), g["fp"]) < 3e-3          # FP model, fp16 storage
    _load_qp(qnn, quant_params_of(g, "qp"))
    assert all(b.fused_ok() for b in qnn.model.blocks)
    e = _rec(parity, "tiny_pixart_alpha/w8a8", qnn(x, t, y, mask=mask).cpu().float(), g["w8a8"], g["w8a8_ref_fp16"])
    assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e
    out1 = qnn(x[:1], t[:1], y[:1], mask=mask[:1]).cpu().float()
    e = _rec(parity, "tiny_pixart_alpha/w8a8_b1", out1, g["w8a8_b1"], g["w8a8_b1_ref_fp16"])
    assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e
    solver = DPMS_alpha(qnn.forward_with_dpmsolver, condition=g["y"][:1].half().to(dev),
                        uncondition=g["dpm_null_y"].half().to(dev), cfg_scale=4.5,
                        model_kwargs=dict(data_info=None, mask=g["mask"][:1].to(dev)))
    out = solver.sample(g["dpm_z"].to(dev), steps=4, order=2, skip_type="time_uniform", method="multistep")
    e = _rec(parity, "tiny_pixart_alpha/dpm_final", out.cpu().float(), g["dpm_final"])
    assert e["vs_ref_fp32"] < 3.4e-3, e           # recorded 2.7e-3: 4 guided steps (cfg 4.5) of this more sensitive tiny net
    # static tensor-wise plan (alpha/w8a8_naive.yaml) on the fused route
    wq, aq = _cfgs(8, dynamic=False, per_group=False, T=1, S=64)
    qn = _pixart("PixArt", g, dev, wq, aq)
    _load_qp(qn, quant_params_of(g, "qp_naive"))
    assert all(b.fused_ok() for b in qn.model.blocks)
    e = _rec(parity, "tiny_pixart_alpha/naive", qn(x, t, y, mask=mask).cpu().float(), g["naive"], g["naive_ref_fp16"])
    assert e["vs_ref_fp32"] < 1.25 * e["ref_fp16_vs_ref_fp32"] + 1e-4, e       # recorded 5.3e-3; the reference's fp16 mode: 6.3e-3


def test_pixart_w4a8_running_smooth_quant_statistic(dev, ops, parity):
    """BASELINE config 5 in miniature (PixArt-MS, 4-bit weights with grids for [4,6,8], dynamic 8-bit activations) with
    the t2i scripts' smooth-quant arrangement: channel balancing on the last block's mlp.fc2 only, its act-scale
    statistic still RUNNING at inference (quant_txt2img.py:297-300).  Every call moves the statistic, hence s and
    W*s: the layer must re-derive and re-pack (stale cache), the other block stays on the fused route."""
    g = load_npz("tiny_pixart_w4a8.npz")
    wq, aq = _cfgs(4, T=1, S=64, smooth=dict(alpha=0.3), mixed_precision=[4, 6, 8])
    qnn = _pixart("PixArtMS", g, dev, wq, aq)
    qnn.set_smooth_quant(smooth_quant=False, smooth_quant_running_stat=False)
    qnn.set_layer_smooth_quant(model=qnn, module_name_list=["blocks.1.mlp.fc2"], smooth_quant=True,
                               smooth_quant_running_stat=True)
    _load_qp(qnn, quant_params_of(g, "qp_after_ptq"))
    assert qnn.model.blocks[0].fused_ok() and not qnn.model.blocks[1].fused_ok()
    fc2 = qnn.model.blocks[1].mlp.fc2
    x, y, mask = g["x"].to(dev), g["y"].half().to(dev), g["mask"].to(dev)
    for j, tv in enumerate((820, 400, 90)):
        out = qnn(x, torch.tensor([tv, tv], device=dev), y, mask=mask).cpu().float()
        a, b = fc2.act_quantizer.act_scale.cpu().float(), g["act_scale_after_call%d" % j]
        assert torch.allclose(a.reshape(b.shape), b, rtol=2e-2, atol=1e-4)             # max|x| of fp16 activations
        e = _rec(parity, "tiny_pixart_w4a8/call%d_t%d" % (j, tv), out, g["w4a8_call%d_t%d" % (j, tv)])
        assert e["vs_ref_fp32"] < 5.2e-3, e           # recorded 4.0 - 4.2e-3 (4-bit weights, quantized final layer)
    qnn.load_bitwidth_config(qnn, {"model.blocks.0.attn.qkv": 8, "model.blocks.1.mlp.fc1": 6}, "weight")
    out = qnn(x, torch.tensor([820, 820], device=dev), y, mask=mask).cpu().float()
    e = _rec(parity, "tiny_pixart_w4a8/mp_call3_t820", out, g["w4a8_mp_call3_t820"])
    assert e["vs_ref_fp32"] < 5.0e-3, e               # recorded 3.9e-3
    assert qnn.model.blocks[0].attn.qkv.packed_weight(0).n_bits == 8


# ----------------------------------------------------------------------------- full-size blocks vs the oracle
def _sd_of(m):
    return {k: v.detach().cpu().float() for k, v in m.state_dict().items()
            if "weight_quantizer" not in k and "act_quantizer" not in k}


@pytest.mark.parametrize("plan", ["w8a8", "w4a8"])
def test_full_size_stdit_block_matches_oracle(dev, ops, parity, plan):
    """ONE STDiT-XL/2 block at the benchmark's size - 16 frames x 1024 tokens, C = 1152, 16 heads of 72, mlp 4608, 97
    prompt tokens - through the fused HIP route vs the CPU oracle on identical weights (BASELINE configs 2 and 3).
    W4A8 = the timestep-aware plan: 4-bit weights, two smooth-quant time-ranges, grids of range 0."""
    import viditq_amd  # noqa
    from viditq_amd import synth
    from viditq_amd.config import loads_yaml
    from oracle import stdit_ref as sr
    m = synth.build_stdit(dev, depth=1, caption_channels=64, seed=11)
    cfg = loads_yaml(synth.W8A8_DYNAMIC if plan == "w8a8" else synth.W4A8_TIMESTEP_AWARE)
    qnn = synth.quantize_model(m, cfg)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    gx = torch.Generator().manual_seed(12)
    x = torch.randn(1, 4, 16, 64, 64, generator=gx).to(dev)
    y = (torch.randn(1, 1, 120, 64, generator=gx) * 0.3).half().to(dev)
    mask = torch.zeros(1, 120, dtype=torch.int64)
    mask[0, :97] = 1
    import viditq_amd.t2v.stdit as st
    blocks = []
    orig = st.STDiTBlock.forward_fused

    def spy(self, x2, *a, **k):
        r = orig(self, x2, *a, **k)
        blocks.append(x2.clone())
        return r
    sd = _sd_of(m)
    cfgd = dict(T=16, S=1024, H=16, depth=1, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(16, 64, 64))
    spec = sr.QSpec(w_bits=8)
    if plan == "w4a8":
        act_scale = {n: l.act_quantizer.act_scale.detach().cpu().float() for n, l in qnn.quant_layers()
                     if n.startswith("blocks") and getattr(l.act_quantizer, "act_scale", None) is not None}
        assert len(act_scale) == 13
        spec = sr.QSpec(w_bits=4, act_scale=act_scale, alpha=[0.11, 0.11], timerange=[[0, 500], [501, 1000]])
    for tv in ((721,) if plan == "w8a8" else (721, 300)):
        t = torch.tensor([tv], device=dev)
        blocks.clear()
        st.STDiTBlock.forward_fused = spy
        try:
            out = qnn(x, t, y, mask=mask.to(dev)).cpu()
        finally:
            st.STDiTBlock.forward_fused = orig
        ref, rblocks = sr.stdit_forward(sd, cfgd, x.cpu().half().float(), t.cpu(), y.cpu().float(), mask, spec,
                                        return_blocks=True)
        eb = _rec(parity, "full_size/stdit_block_%s_t%d" % (plan, tv), blocks[0].cpu().float().reshape(1, 16384, 1152),
                  rblocks[0], tokens=16384, C=1152)
        eo = _rec(parity, "full_size/stdit_depth1_model_%s_t%d" % (plan, tv), out, ref)
        # north_star's 1e-3 HOLDS on the full-size block: recorded 5.7e-4 (W8A8) / 5.8e-4 (W4A8), depth-1 model 8.0e-4
        # (profiles/r02_parity.json); asserted at 1.25 x the recorded values
        assert eb["vs_ref_fp32"] < 7.3e-4, eb
        assert eo["vs_ref_fp32"] < 1.0e-3, eo
    assert qnn.check_status() == 0


@pytest.mark.parametrize("w_bits", [4, 8])
def test_full_size_pixart_block_n4096_lp300_matches_oracle(dev, ops, parity, w_bits):
    """ONE PixArt-XL/2 block at PixArt-Sigma 1024^2 size (BASELINE config 5): 4096 image tokens, prompts of up to 300
    tokens, B = 2 (uncond | cond batched as the t2i loop does: per-token scales shared over the batch), 4-bit weights,
    dynamic 8-bit activations - fused HIP route vs the CPU oracle.  Exercises the 4096-token flash attention and the
    varlen cross attention beyond the 128-key register kernel at model level."""
    import viditq_amd  # noqa
    from viditq_amd import synth, t2i
    from viditq_amd.qdiff.models import QuantModel
    from oracle import pixart_ref as pr
    from oracle import stdit_ref as sr
    torch.manual_seed(21)
    m = t2i.PixArtMS(input_size=128, depth=1, hidden_size=1152, num_heads=16, model_max_length=300, caption_channels=64,
                     pe_interpolation=2.0, dtype=torch.float16)
    synth.redraw_zero_init(m, 22)
    m = m.half().to(dev).eval()
    wq, aq = _cfgs(w_bits, T=1, S=4096, mixed_precision=[4, 6, 8])
    qnn = QuantModel(m, wq, aq, model_type="pixart")
    qnn.set_module_name_for_quantizer(qnn.model)
    qnn.fp_layer_list = list(PIX_FP)
    for _, layer in qnn.quant_layers():
        layer.weight_quantizer(layer.weight.detach())
    qnn.set_quant_init_done("weight")
    qnn.set_quant_init_done("activation")
    qnn.set_quant_state(True, True)
    assert all(b.fused_ok() for b in qnn.model.blocks)
    gx = torch.Generator().manual_seed(23)
    x = torch.randn(2, 4, 128, 128, generator=gx).to(dev)
    y = (torch.randn(2, 1, 300, 64, generator=gx) * 0.3).half().to(dev)
    mask = torch.zeros(2, 300, dtype=torch.int64)
    mask[0, :300] = 1
    mask[1, :143] = 1
    t = torch.tensor([500, 500], device=dev)
    from helpers import spy_fused
    with spy_fused(t2i.pixart.PixArtMSBlock) as blocks:
        out = qnn(x, t, y, mask=mask.to(dev)).cpu().float()
    assert len(blocks) == 1
    sd = _sd_of(m)
    pe = qnn.model._pos_embed(dev, torch.float16).cpu().float()
    spec = sr.QSpec(w_bits=w_bits, fp_layers=pr.T2I_FP_LAYERS)
    ref, rblocks = pr.pixart_forward(sd, dict(H=16, depth=1, patch=2, out_ch=8), x.cpu().half().float(), t.cpu(),
                                     y.cpu().float(), mask, spec, pe, return_blocks=True)
    # the BLOCK alone (what north_star's 1e-3 is about) and the depth-1 model behind it, whose output also carries the
    # quantized final_layer.linear of the t2i FP list; yardstick at this width: the reference's own fp16 mode is 0.9e-3
    # (W8A8) / 1.6e-3 (W4A8) from its fp32 result after one block and 5.4e-3 / 1.2e-2 at the model output
    # (xl_width/pixart_* in tests/golden/xl_width_ref.npz)
    eb = _rec(parity, "full_size/pixart_block_w%da8_n4096_lp300_b2" % w_bits, blocks[0].cpu().float().reshape(2, 4096, 1152),
              rblocks[0], tokens=4096, prompt_tokens=[300, 143])
    e = _rec(parity, "full_size/pixart_depth1_model_w%da8_n4096_lp300_b2" % w_bits, out, ref, tokens=4096,
             prompt_tokens=[300, 143])
    assert eb["vs_ref_fp32"] < 6.5e-4, eb        # recorded 4.96e-4 (W4A8) / 4.98e-4 (W8A8): north_star's 1e-3 holds on the block
    assert e["vs_ref_fp32"] < 4.5e-3, e          # recorded 3.5e-3 (W4A8): 4-bit weights + a quantized final layer
    assert qnn.check_status() == 0


# ----------------------------------------------------------------------------- eps-fill at model level
def _eps_fill_case(dev, depth=2):
    import viditq_amd  # noqa
    from viditq_amd import synth
    from viditq_amd.config import loads_yaml
    m = synth.build_stdit(dev, depth=depth, hidden_size=64, num_heads=4, input_size=(4, 8, 8), model_max_length=12,
                          caption_channels=32, seed=5)
    with torch.no_grad():
        m.y_embedder.y_proj.fc1.bias.zero_()
        m.y_embedder.y_proj.fc2.bias.zero_()
    qnn = synth.quantize_model(m, loads_yaml(synth.W8A8_DYNAMIC))
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 4, 4, 8, 8, generator=g).to(dev)
    y = (torch.randn(1, 1, 12, 32, generator=g) * 0.3).half().to(dev)
    mask = torch.ones(1, 12, dtype=torch.int64, device=dev)
    return m, qnn, x, y, mask


def test_kv_linear_global_eps_fill_is_reproduced_exactly(dev, ops, parity):
    """Prompt embeddings are NOT normalised: a prompt token whose embedded row is (near-)constant makes the reference
    set EVERY token's quant step of cross_attn.kv_linear to 1e-6 (base_quantizer.py:219-223) - zero points of ~1e6,
    saturated codes.  The prompt K/V are therefore computed on the integer route AND on the exact fill route
    (vq_fakequant_act + fp16 GEMM) and selected on the device by the quantizer's status bit: with an all-zero prompt
    token the model output equals the ORACLE's (which performs the fill), without it nothing changes bit for bit."""
    import viditq_amd.t2v.stdit as st
    from oracle import stdit_ref as sr
    m, qnn, x, y, mask = _eps_fill_case(dev)
    t = torch.tensor([500], device=dev)
    cfgd = dict(T=4, S=16, H=4, depth=2, patch=(1, 2, 2), in_ch=4, out_ch=8, input_size=(4, 8, 8))
    sd = _sd_of(m)
    base = qnn(x, t, y, mask=mask).clone()
    st.EXACT_KV_EPS_FILL = False
    try:
        assert torch.equal(qnn(x, t, y, mask=mask), base)          # no degenerate token: the select keeps the integer result
    finally:
        st.EXACT_KV_EPS_FILL = True
    assert qnn.check_status() == 0
    y[0, 0, 3] = 0                                  # an all-zero prompt token -> embedded row 0 -> step 0 < 1e-6
    out = qnn(x, t, y, mask=mask).cpu()
    assert torch.isfinite(out).all()
    ref = sr.stdit_forward(sd, cfgd, x.cpu().half().float(), t.cpu(), y.cpu().float(), mask.cpu(), sr.QSpec(w_bits=8))
    e = _rec(parity, "eps_fill/tiny_stdit_zero_prompt_token_exact_route", out, ref)
    assert e["vs_ref_fp32"] < 2.5e-3, e             # the tiny models' usual fp16-storage distance, not a saturated layer
    assert qnn.check_status() == 0                  # reproduced, not flagged
    # and the distance the integer route alone would be at (what the flag used to warn about)
    st.EXACT_KV_EPS_FILL = False
    try:
        flagged = qnn(x, t, y, mask=mask).cpu()
    finally:
        st.EXACT_KV_EPS_FILL = True
    parity["eps_fill/tiny_stdit_zero_prompt_token_integer_route_only"] = {"vs_ref_fp32": rel_l2(flagged, ref)}


def test_kv_linear_eps_fill_is_flagged_when_the_exact_route_is_off(dev, ops):
    """With VQ_EXACT_KV_EPS_FILL=0 the integer route keeps per-token grids and MUST say so: status bit +
    warning, or an exception on request."""
    import warnings
    import viditq_amd.t2v.stdit as st
    m, qnn, x, y, mask = _eps_fill_case(dev, depth=1)
    st.EXACT_KV_EPS_FILL = False
    try:
        qnn(x, torch.tensor([500], device=dev), y, mask=mask)
        assert qnn.check_status() == 0
        y[0, 0, 3] = 0
        out = qnn(x, torch.tensor([500], device=dev), y, mask=mask)
        assert torch.isfinite(out).all()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert qnn.check_status() & 1
        assert any("1e-6" in str(i.message) for i in w)
        with pytest.raises(RuntimeError):
            qnn.check_status(raise_on_eps=True)
    finally:
        st.EXACT_KV_EPS_FILL = True


# ----------------------------------------------------------------------------- prompt sharding: 1 rank == N ranks
def _tiny_sharded_job(dev, rank, world, n_prompts=3, steps=2):
    import viditq_amd  # noqa
    from viditq_amd import shard, synth
    from viditq_amd.config import loads_yaml
    from viditq_amd.t2v import IDDPM
    m = synth.build_stdit(dev, depth=2, hidden_size=64, num_heads=4, input_size=(4, 8, 8), model_max_length=12,
                          caption_channels=32, seed=0)
    cfg = loads_yaml(synth.W8A8_DYNAMIC)
    cfg.quant.activation.quantizer["n_spatial_token"], cfg.quant.activation.quantizer["n_temporal_token"] = 16, 4
    qnn = shard.quantize_and_distribute(m, cfg, rank, world)
    embeds, _ = synth.synthetic_prompts(n_prompts, dev, model_max_length=12, caption_channels=32)
    sch = IDDPM(num_sampling_steps=steps, cfg_scale=4.0)
    return shard.sample_sharded(qnn, sch, embeds, n_prompts, rank, world, z_size=(4, 4, 8, 8), seed=42)


def test_sample_sharded_world1_equals_per_prompt_loops(dev, ops):
    """shard.sample_sharded at world 1 = one DDIM loop per prompt with the per-prompt noise generator (seed + index):
    the property that makes an N-rank run reproduce the 1-rank latents."""
    import viditq_amd  # noqa
    from viditq_amd import shard, synth
    from viditq_amd.config import loads_yaml
    from viditq_amd.t2v import IDDPM
    full = _tiny_sharded_job(dev, 0, 1)
    assert full.shape == (3, 4, 4, 8, 8) and torch.isfinite(full).all()
    m = synth.build_stdit(dev, depth=2, hidden_size=64, num_heads=4, input_size=(4, 8, 8), model_max_length=12,
                          caption_channels=32, seed=0)
    cfg = loads_yaml(synth.W8A8_DYNAMIC)
    cfg.quant.activation.quantizer["n_spatial_token"], cfg.quant.activation.quantizer["n_temporal_token"] = 16, 4
    qnn = synth.quantize_model(m, cfg)
    embeds, _ = synth.synthetic_prompts(3, dev, model_max_length=12, caption_channels=32)
    sch = IDDPM(num_sampling_steps=2, cfg_scale=4.0)
    for i in (2, 0):                                               # any order, any subset: same latents
        z = synth.synthetic_latent(i, z_size=(4, 4, 8, 8), seed=42, device=dev)
        y = embeds["y"][i:i + 1].permute(1, 0, 2, 3, 4).reshape(2, 1, 12, 32)
        out = sch.ddim_sample_loop(qnn, z, dict(y=y, mask=embeds["mask"][i:i + 1]))
        assert torch.equal(out[0], full[i])


def test_quantize_and_distribute_over_a_one_rank_rccl_group(dev, ops):
    """The multi-GPU set-up path on ONE device: a one-rank NCCL (= RCCL) process group, rank 0 packs straight into the
    broadcast arena, the broadcast and install run on device buffers, gather_latents goes through all_gather - and the
    model computes exactly what the world-1 shortcut computes."""
    import socket
    import torch.distributed as dist
    import viditq_amd  # noqa
    from viditq_amd import shard, synth
    from viditq_amd.config import loads_yaml
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        outs = []
        for force in (True, False):
            m = synth.build_stdit(dev, depth=2, hidden_size=64, num_heads=4, input_size=(4, 8, 8), model_max_length=12,
                                  caption_channels=32, seed=0)
            cfg = loads_yaml(synth.W4A8_TIMESTEP_AWARE)      # smooth-quant statistics travel in the broadcast as well
            cfg.quant.activation.quantizer["n_spatial_token"], cfg.quant.activation.quantizer["n_temporal_token"] = 16, 4
            qnn = shard.quantize_and_distribute(m, cfg, 0, 1, force_collective=force)
            if force:
                assert getattr(qnn, "_packed_arena", None) is not None
            embeds, _ = synth.synthetic_prompts(1, dev, model_max_length=12, caption_channels=32)
            z = synth.synthetic_latent(0, z_size=(4, 4, 8, 8), seed=42, device=dev).half()
            y = embeds["y"][0:1].permute(1, 0, 2, 3, 4).reshape(2, 1, 12, 32)
            t = torch.full((1,), 721, device=dev, dtype=torch.long)
            with torch.no_grad():
                outs.append(qnn(z, t, y[:1], mask=embeds["mask"][0:1], timestep_id=721).float())
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
        x = torch.randn(2, 4, 4, 8, 8, device=dev)
        torch.cuda.synchronize()
        # world-1 group, forced through the collective: identity
        pad = [torch.empty_like(x)]
        dist.all_gather(pad, x)
        assert torch.equal(pad[0], x)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _sharded_worker(rank, world, port, ret):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        with torch.no_grad():
            full = _tiny_sharded_job(dev, rank, world)             # RCCL broadcast of the arena + all_gather of latents
            if rank == 0:
                ref = _tiny_sharded_job(dev, 0, 1)
                ret["equal"] = bool(torch.equal(full, ref))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU tier)")
def test_sample_sharded_world2_equals_world1_bit_for_bit():
    """Two ranks over RCCL: rank 0 packs, ONE broadcast, prompts 0,2 on rank 0 and 1 on rank 1, all_gather - and the
    gathered latents equal the single-rank run bit for bit (reference: single device only, quant_txt2video.py:71-72)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_sharded_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("equal") is True
